// Host check of the bucket ownership of the single-pass shard flow (csrc/kc_shard.hpp): for every number of buckets
// and shards the ranges [shard_first_bucket(d), shard_first_bucket(d + 1)) tile [0, P1) in order, and shard_of_bucket
// names the range a bucket lies in; the header sizes are consistent.  Compiled by hipcc, runs on the CPU (no kernel is
// launched).
#include <cstdio>

#include "../../mhm2_kmer_analysis_v2_amd/csrc/kc_shard.hpp"

int main() {
  unsigned long long bad = 0, checked = 0;
  for (uint32_t P1 = 1; P1 <= 1024; P1 += (P1 < 40 ? 1 : 37)) {
    for (uint32_t n = 1; n <= 64 && n <= P1; n++) {
      if (kc::shard_first_bucket(0, P1, n) != 0 || kc::shard_first_bucket(n, P1, n) != P1) bad++;
      for (uint32_t d = 0; d < n; d++) {
        const uint32_t lo = kc::shard_first_bucket(d, P1, n), hi = kc::shard_first_bucket(d + 1, P1, n);
        if (hi < lo) bad++;
        for (uint32_t b = lo; b < hi; b++, checked++)
          if (kc::shard_of_bucket(b, P1, n) != d) bad++;
        if (kc::shard_header_words(hi - lo) != 4 + (hi - lo + 1) / 2) bad++;
      }
    }
  }
  std::printf("checked=%llu bad=%llu\n", checked, bad);
  return bad ? 1 : 0;
}
