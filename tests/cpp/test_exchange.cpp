// Drives kcount_mi355::ShardExchange (csrc/kc_exchange.hpp) the way MHM2's count_kmers + KmerDHT would: blocks of reads
// in, one RCCL communicator, finalize, "<kmer> <count> <L> <R>" lines (kmer_dht.cpp:284 format) out, sorted, each
// behind the word KMER.
//   test_exchange K [RANK NRANKS IDFILE]     reads: one case-masked read per line on stdin (lower case = low quality)
// With one rank (the default) the communicator has a single member: counts are still all-gathered through RCCL, the
// rank's own share goes the direct way.  With NRANKS > 1 start one process per GPU; rank 0 writes the ncclUniqueId to
// IDFILE, the others wait for it; every rank takes the lines whose number is RANK modulo NRANKS and prints its shard.
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "../../mhm2_kmer_analysis_v2_amd/csrc/kc_exchange.hpp"

using namespace kcount_mi355;

#define CHECK(x)                                                                   \
  do {                                                                             \
    if (!(x)) {                                                                    \
      std::fprintf(stderr, "test_exchange: %s failed (line %d)\n", #x, __LINE__); \
      return 3;                                                                    \
    }                                                                              \
  } while (0)

int main(int argc, char **argv) {
  const int k = argc > 1 ? atoi(argv[1]) : 21;
  const int rank = argc > 4 ? atoi(argv[2]) : 0, nranks = argc > 4 ? atoi(argv[3]) : 1;
  const char *idfile = argc > 4 ? argv[4] : nullptr;
  std::vector<std::string> reads;
  std::string line;
  for (size_t i = 0; std::getline(std::cin, line); i++)
    if ((int)(i % nranks) == rank) reads.push_back(line);
  CHECK(hipSetDevice(rank) == hipSuccess);
  ncclUniqueId id;
  if (rank == 0) {
    CHECK(ncclGetUniqueId(&id) == ncclSuccess);
    if (idfile) {
      std::ofstream f(std::string(idfile) + ".tmp", std::ios::binary);
      f.write((const char *)&id, sizeof(id));
      f.close();
      std::rename((std::string(idfile) + ".tmp").c_str(), idfile);
    }
  } else {
    for (int t = 0; t < 600; t++) {
      std::ifstream f(idfile, std::ios::binary);
      if (f.read((char *)&id, sizeof(id))) break;
      usleep(100000);
    }
  }
  ncclComm_t comm;
  CHECK(ncclCommInitRank(&comm, nranks, id, rank) == ncclSuccess);

  kc_config cfg{};
  cfg.kmer_len = k;
  cfg.qual_offset = 33;
  cfg.dmin_thres = 2;
  cfg.device = rank;
  cfg.rank_me = rank;
  cfg.rank_n = nranks;
  cfg.max_kmers_buffered = 1 << 22;
  // KC_EXCHANGE_FLOW=records: the hash-ownership flow; wire-units: the same with the library's own wire record (units of
  // four six-byte records in pieces, kc_wire_unit); default: the single-pass flow (a shard owns level-1 buckets)
  const char *fenv = getenv("KC_EXCHANGE_FLOW");
  const bool wire_units = fenv && !strcmp(fenv, "wire-units");
  const bool records = wire_units || (fenv && !strcmp(fenv, "records"));
  if (wire_units) cfg.flags |= KC_FLAG_WIRE_UNITS;
  int st = 0;
  kc_ctx *ctx = kc_create(&cfg, &st);
  CHECK(ctx != nullptr);
  int unit_words = kc_record_longs(k), unit_records = 1, pieces = 1;
  if (wire_units) {
    kc_tuning t{};  // the benchmark's fan-outs: level 1 writes six-byte records at k = 21
    t.p1 = 1024;
    t.p2 = 1024;
    CHECK(kc_set_tuning(ctx, &t) == KC_OK);
    CHECK(kc_wire_unit(ctx, &unit_words, &unit_records, &pieces) == KC_OK);
    if (k == 21) CHECK(unit_words == 3 && unit_records == 4 && pieces >= 2);
    else CHECK(unit_words == kc_record_longs(k) && unit_records == 1 && pieces == 1);
  }
  const int nl = kc_num_longs(k);
  // blocks of up to 97 reads: several exchanges, the last ones possibly empty on some ranks
  const size_t per_block = 97;
  size_t nblocks = (reads.size() + per_block - 1) / per_block;
  {  // every rank makes the same number of collective calls
    unsigned long long mine = nblocks, *d = nullptr, all = 0;
    CHECK(hipMalloc((void **)&d, 8) == hipSuccess);
    CHECK(hipMemcpy(d, &mine, 8, hipMemcpyHostToDevice) == hipSuccess);
    CHECK(ncclAllReduce(d, d, 1, ncclUint64, ncclMax, comm, nullptr) == ncclSuccess);
    CHECK(hipDeviceSynchronize() == hipSuccess);
    CHECK(hipMemcpy(&all, d, 8, hipMemcpyDeviceToHost) == hipSuccess);
    nblocks = all;
    (void)hipFree(d);
  }
  const int rl = kc_record_longs(k);
  ShardExchange ex(ctx, comm, rank, nranks, records ? unit_words : rl, records ? per_block * 400 + 64 : (per_block * 400 + 64) * rl + 2048, nullptr,
                   records ? ShardExchange::RECORDS : ShardExchange::BUCKETS, pieces);
  if (ex.init() != KC_OK) {
    std::fprintf(stderr, "init: %s\n", ex.last_error());
    return 4;
  }
  uint64_t expect = 0;
  for (size_t b = 0; b < nblocks; b++) {
    std::string bases, quals;
    std::vector<uint64_t> offs(1, 0);
    for (size_t i = b * per_block; i < std::min(reads.size(), (b + 1) * per_block); i++) {
      for (char c : reads[i]) {
        const bool low = c >= 'a' && c <= 'z';
        bases.push_back(low ? (char)(c - 32) : c);
        quals.push_back(low ? '#' : 'I');
      }
      offs.push_back(bases.size());
      if ((int)reads[i].size() >= k + 2) expect += reads[i].size() - k - 1;
    }
    const uint64_t nr = offs.size() - 1;
    if (ex.add_block((const uint8_t *)bases.data(), (const uint8_t *)quals.data(), offs.data(), nr, 0) != KC_OK) {
      std::fprintf(stderr, "add_block: %s\n", ex.last_error());
      return 5;
    }
  }
  if (ex.finish() != KC_OK) {
    std::fprintf(stderr, "finish: %s\n", ex.last_error());
    return 6;
  }
  if (records && unit_records == 1) CHECK(ex.records_sent() == expect);
  if (records && unit_records > 1) CHECK(ex.records_sent() * unit_records >= expect && ex.records_sent() * unit_records <= expect + 4096);
  kc_result r;
  CHECK(kc_finalize(ctx, &r) == KC_OK);
  if (nranks == 1) {
    kc_stats stats;
    CHECK(kc_get_stats(ctx, &stats) == KC_OK);
    CHECK(stats.kmers_inserted == expect);
  }
  std::vector<uint64_t> keys(r.n * nl);
  std::vector<uint16_t> counts(r.n);
  std::vector<uint8_t> left(r.n), right(r.n);
  CHECK(kc_copy_results(ctx, keys.data(), counts.data(), left.data(), right.data()) == KC_OK);
  std::vector<std::string> out;
  for (uint64_t i = 0; i < r.n; i++) {
    std::string s;
    for (int j = 0; j < k; j++) s.push_back("ACGT"[(keys[i * nl + j / 32] >> (2 * (31 - (j % 32)))) & 3]);
    out.push_back(s + " " + std::to_string(counts[i]) + " " + (char)left[i] + " " + (char)right[i]);
  }
  std::sort(out.begin(), out.end());
  for (auto &s : out) std::cout << "KMER " << s << "\n";  // the prefix tells result lines from RCCL's own banner on stdout
  kc_destroy(ctx);
  ncclCommDestroy(comm);
  return 0;
}
