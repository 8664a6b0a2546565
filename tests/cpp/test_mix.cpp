// The invertible mix behind the compact records (kc_feistel_fwd / kc_feistel_inv, kc_common.hpp) on the host:
// a permutation of the 4^k k-mers for every k it is used with, its inverse undoes it, and the bits the bucketed path
// reads as bucket, region and probe start are spread evenly.
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../mhm2_kmer_analysis_v2_amd/csrc/kc_common.hpp"

int main() {
  int bad = 0;
  // exhaustive: small k
  for (int k = 3; k <= 11; k++) {
    const uint64_t n = 1ULL << (2 * k);
    std::vector<uint8_t> seen(n, 0);
    for (uint64_t v = 0; v < n; v++) {
      const uint64_t m = kc_feistel_fwd(v, k);
      if (m >= n || seen[m]++) bad++;
      if (kc_feistel_inv(m, k) != v) bad++;
    }
  }
  // sampled: every k up to the largest one that uses the mix
  uint64_t x = 0x9E3779B97F4A7C15ULL;
  for (int k = 12; k <= KC_COMPACT_MAX_K; k++) {
    const uint64_t mask = (1ULL << (2 * k)) - 1;
    for (int i = 0; i < 200000; i++) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      const uint64_t v = x & mask, m = kc_feistel_fwd(v, k);
      if (m > mask || kc_feistel_inv(m, k) != v) bad++;
    }
  }
  // consecutive k-mers (the most regular input) spread over 1024 buckets: no bucket beyond 1.3x the mean
  {
    const int k = 21, P = 1024;
    const uint64_t n = 1 << 22;
    std::vector<uint32_t> hist(P, 0), hist2(P, 0), slots(2048, 0);
    for (uint64_t v = 0; v < n; v++) {
      const uint64_t m = kc_feistel_fwd(v, k);
      hist[m >> (2 * k - 10)]++;
      hist2[(m >> (2 * k - 20)) & 1023]++;
      slots[m & 2047]++;
    }
    for (int b = 0; b < P; b++)
      if (hist[b] > 1.3 * n / P || hist2[b] > 1.3 * n / P) bad++;
    for (int s = 0; s < 2048; s++)
      if (slots[s] > 1.3 * n / 2048) bad++;
  }
  std::printf("bad=%d\n", bad);
  return bad != 0;
}
