// The member surface src/kcount/kcount_gpu.cpp uses of the two device drivers, written inside `namespace kcount_gpu` with
// the aliases INTEGRATION.md prescribes -- every driver expression of kcount_gpu.cpp:120-156 (process_block), :197
// (get_elapsed_times), :218-219 (the state's member built from `{}`), :253 (init), :290-345 (insert_supermer,
// flush_inserts, get_stats, get_capacity, pass_type, get_qf_load_factor), :376-400 (the contig-pass branch), :412-471
// (done_all_inserts, get_final_capacity, the two iteration loops) and :492 (get_elapsed_time), with the reference's own
// argument types.  Compiled
//   * against the adapter (default): hipcc -std=c++17 test_surface.cpp -lkcount_mi355   (CPU suite: compile + link;
//     GPU suite: run -- reads on stdin, one case-masked read per line, "KMER count L R" lines on stdout);
//   * against the reference's own headers, unmodified, syntax only:
//     g++ -std=c++17 -fsyntax-only -DSURFACE_REFERENCE -I/root/reference/src -I/root/reference/src/kcount test_surface.cpp
// so the same text is valid with either: what compiles here is what the reference's host file needs.
#include <algorithm>
#include <cstdint>
#include <iostream>
#include <string>
#include <tuple>
#include <vector>

#ifdef SURFACE_REFERENCE
#include "kcount-gpu/parse_and_pack.hpp"
#include "kcount-gpu/gpu_hash_table.hpp"
#else
// ---- the edit INTEGRATION.md prescribes for src/kcount/kcount_gpu.cpp, verbatim ----
#include "../../mhm2_kmer_analysis_v2_amd/csrc/kcount_driver.hpp"
namespace kcount_gpu {
using ParseAndPackGPUDriver = kcount_mi355::ParseAndPackDriver;
template <int MAX_K> using HashTableGPUDriver = kcount_mi355::HashTableDriver<MAX_K>;
using kcount_mi355::CountExts; using kcount_mi355::KmerArray; using kcount_mi355::InsertStats;
using kcount_mi355::SupermerInfo;
using kcount_mi355::PASS_TYPE; using kcount_mi355::READ_KMERS_PASS; using kcount_mi355::CTG_KMERS_PASS;
using kcount_mi355::count_t; using kcount_mi355::ext_count_t;
}
// ---- end of the edit ----
#endif

using namespace std;
using namespace kcount_gpu;  // kcount_gpu.cpp:72

using kmer_count_t = uint16_t;  // kmer_dht.hpp:54

// kcount_gpu.cpp:74-91
struct SeqBlockInserterState {
  ParseAndPackGPUDriver *pnp_gpu_driver;
  int64_t num_kmers = 0;
  string seq_block;
  vector<kmer_count_t> depth_block;
};

// kcount_gpu.cpp:214-220
template <int MAX_K>
struct HashTableInserterState {
  HashTableGPUDriver<MAX_K> ht_gpu_driver;

  HashTableInserterState()
      : ht_gpu_driver({}) {}
};

struct Entry {
  string kmer;
  unsigned count;
  char left, right;
};

template <int MAX_K>
static int surface(int rank_me, int rank_n, int kmer_len, int qual_offset, int minimizer_len, const string &block, vector<Entry> &result) {
  // ---- SeqBlockInserter ctor, kcount_gpu.cpp:93-100 ----
  double init_time;
  auto *state = new SeqBlockInserterState();
  state->pnp_gpu_driver = new ParseAndPackGPUDriver(rank_me, rank_n, qual_offset, kmer_len, (MAX_K + 31) / 32, minimizer_len, init_time);
  // ---- HashTableInserter::init, kcount_gpu.cpp:224-267 ----
  vector<HashTableInserterState<MAX_K> *> targets;
  bool use_qf = false;
  for (int r = 0; r < rank_n; r++) {
    auto *hstate = new HashTableInserterState<MAX_K>();
    size_t max_elems = 100000, max_ctg_elems = 0, num_errors = 10000;
    auto gpu_avail_mem_per_rank = (8e9 - 3000000 * 14) * 0.9;  // a double, as at :240
    string driver_msgs, driver_warnings;
    hstate->ht_gpu_driver.init(r, rank_n, kmer_len, max_elems, max_ctg_elems, num_errors, gpu_avail_mem_per_rank, driver_msgs, driver_warnings,
                               use_qf);
    if (!driver_warnings.empty()) cerr << driver_warnings;
    targets.push_back(hstate);
  }
  // ---- process_block, kcount_gpu.cpp:110-165 ----
  state->seq_block = block;
  unsigned int num_valid_kmers = 0;
  bool from_ctgs = !state->depth_block.empty();
  bool success = state->pnp_gpu_driver->process_seq_block(state->seq_block, num_valid_kmers);
  if (!success) return 2;
  state->pnp_gpu_driver->pack_seq_block(state->seq_block);
  int num_targets = (int)state->pnp_gpu_driver->supermers.size();
  for (int i = 0; i < num_targets; i++) {
    auto target = state->pnp_gpu_driver->supermers[i].target;
    auto offset = state->pnp_gpu_driver->supermers[i].offset;
    auto len = state->pnp_gpu_driver->supermers[i].len;
    string seq;
    int packed_len = len / 2;
    if (offset % 2 || len % 2) packed_len++;
    seq = state->pnp_gpu_driver->packed_seqs.substr(offset / 2, packed_len);
    if (offset % 2) seq[0] &= 15;
    if ((offset + len) % 2) seq[seq.length() - 1] &= 240;
    kmer_count_t count = (from_ctgs ? state->depth_block[offset + 1] : (kmer_count_t)1);
    // KmerDHT::add_supermer -> (RPC) -> HashTableInserter::insert_supermer, kcount_gpu.cpp:289-293
    targets[target]->ht_gpu_driver.insert_supermer(seq, count);
    state->num_kmers += (2 * seq.length() - kmer_len);
  }
  // ---- done_processing, kcount_gpu.cpp:197 ----
  auto [gpu_time_tot, gpu_time_kernel] = state->pnp_gpu_driver->get_elapsed_times();
  if (!(gpu_time_tot >= gpu_time_kernel)) return 6;
  for (auto *hstate : targets) {
    // ---- flush_inserts, kcount_gpu.cpp:295-363 ----
    hstate->ht_gpu_driver.flush_inserts();
    int ncalls = hstate->ht_gpu_driver.get_num_gpu_calls();
    (void)ncalls;
    auto insert_stats = hstate->ht_gpu_driver.get_stats();
    uint64_t dropped = (uint64_t)insert_stats.dropped, attempted = (uint64_t)insert_stats.attempted, inserts = (uint64_t)insert_stats.new_inserts;
    uint64_t capacity = hstate->ht_gpu_driver.get_capacity();
    if (hstate->ht_gpu_driver.pass_type == kcount_gpu::READ_KMERS_PASS)
      cerr << "GPU hash table stats for read kmers pass: attempted " << attempted << " dropped " << dropped << " capacity " << capacity << "\n";
    else
      cerr << "GPU hash table stats for ctg kmers pass\n";
    if (use_qf && hstate->ht_gpu_driver.pass_type == kcount_gpu::READ_KMERS_PASS) {
      uint64_t uq = (uint64_t)insert_stats.num_unique_qf, dq = (uint64_t)insert_stats.dropped_qf;
      double qf_load = hstate->ht_gpu_driver.get_qf_load_factor();
      cerr << uq << dq << qf_load;
    }
    double load = (double)(insert_stats.new_inserts) / capacity;
    (void)load;
    (void)inserts;
    // ---- insert_into_local_hashtable, kcount_gpu.cpp:371-506 ----
    if (hstate->ht_gpu_driver.pass_type == CTG_KMERS_PASS) {
      uint64_t attempted_inserts = 0, dropped_inserts = 0, new_inserts = 0;
      hstate->ht_gpu_driver.done_ctg_kmer_inserts(attempted_inserts, dropped_inserts, new_inserts);
      auto all_capacity = (uint64_t)hstate->ht_gpu_driver.get_capacity();
      (void)all_capacity;
    }
    uint64_t num_dropped = 0, num_entries = 0, num_purged = 0;
    hstate->ht_gpu_driver.done_all_inserts(num_dropped, num_entries, num_purged);
    if (num_dropped) return 3;
    auto all_capacity = (uint64_t)hstate->ht_gpu_driver.get_final_capacity();
    (void)all_capacity;
    int64_t max_kmer_count = 0;
    hstate->ht_gpu_driver.begin_iterate();
    while (true) {
      auto [kmer_array, count_exts] = hstate->ht_gpu_driver.get_next_entry();
      if (!kmer_array) break;
      if (count_exts->count > max_kmer_count) max_kmer_count = count_exts->count;
    }
    uint64_t invalid = 0;
    hstate->ht_gpu_driver.begin_iterate();
    while (true) {
      auto [kmer_array, count_exts] = hstate->ht_gpu_driver.get_next_entry();
      if (!kmer_array) break;
      // empty slot
      if (!count_exts->count) continue;
      if ((char)count_exts->left == 'X' || (char)count_exts->right == 'X' || (char)count_exts->left == 'F' || (char)count_exts->right == 'F') {
        invalid++;
        continue;
      }
      if ((count_exts->count < 2)) {
        invalid++;
        continue;
      }
      Entry e;
      const uint64_t *longs = reinterpret_cast<const uint64_t *>(kmer_array->longs);  // Kmer<MAX_K> kmer(longs), :461
      for (int i = 0; i < kmer_len; i++) e.kmer.push_back("ACGT"[(longs[i / 32] >> (2 * (31 - (i % 32)))) & 3]);
      e.count = static_cast<kmer_count_t>(min(count_exts->count, static_cast<count_t>(UINT16_MAX)));
      e.left = (char)count_exts->left;
      e.right = (char)count_exts->right;
      result.push_back(e);
    }
    if (invalid) return 8;  // entries come voted and purged
    double gpu_insert_time = 0, gpu_kernel_time = 0;
    hstate->ht_gpu_driver.get_elapsed_time(gpu_insert_time, gpu_kernel_time);
    delete hstate;
  }
  delete state->pnp_gpu_driver;
  delete state;
  return 0;
}

// the types of the header that host code may name (gpu_hash_table.hpp:54-75,109-115; parse_and_pack.hpp:50-54)
static_assert(sizeof(CountExts) == 8 && sizeof(SupermerInfo) == 12, "layouts the host relies on");
static_assert(sizeof(KmerArray<64>) == 16 && KmerArray<96>::N_LONGS == 3, "KmerArray is its words");
static_assert(READ_KMERS_PASS == 0 && CTG_KMERS_PASS == 1, "PASS_TYPE");
static_assert(sizeof(count_t) == 4 && sizeof(ext_count_t) == 2, "counter types");

template int surface<32>(int, int, int, int, int, const string &, vector<Entry> &);
template int surface<64>(int, int, int, int, int, const string &, vector<Entry> &);
template int surface<96>(int, int, int, int, int, const string &, vector<Entry> &);

#ifndef SURFACE_REFERENCE
int main(int argc, char **argv) {
  const int k = argc > 1 ? atoi(argv[1]) : 21;
  string block, line;
  while (getline(cin, line)) block += line + "_";  // one case-masked read per line (kcount_gpu.cpp:167-180)
  int m = k * 2 / 3 + 1;  // kmer_dht.cpp:117-119
  m = m < 15 ? 15 : m > 27 ? 27 : m;
  if (m > k) m = k;
  vector<Entry> result;
  const int rc = k < 32 ? surface<32>(0, 3, k, 33, m, block, result) : k < 64 ? surface<64>(0, 3, k, 33, m, block, result)
                                                                             : surface<96>(0, 3, k, 33, m, block, result);
  if (rc) return rc;
  vector<string> out;
  for (auto &e : result) out.push_back(e.kmer + " " + to_string(e.count) + " " + e.left + " " + e.right);
  sort(out.begin(), out.end());
  for (auto &l : out) cout << l << "\n";
  return 0;
}
#endif
