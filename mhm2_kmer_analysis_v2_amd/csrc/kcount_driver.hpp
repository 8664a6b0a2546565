// kcount_driver.hpp -- C++ host side over the C ABI (include/kcount_mi355.h): the two device drivers
// the reference's GPU backend is written against, re-created on top of libkcount_mi355 so that
// src/kcount/kcount_gpu.cpp needs only the include and namespace swapped.
//
//   reference class (src/kcount/kcount-gpu/)                      here
//   kcount_gpu::ParseAndPackGPUDriver  parse_and_pack.hpp:57-87    kcount_mi355::ParseAndPackDriver
//   kcount_gpu::HashTableGPUDriver<K>  gpu_hash_table.hpp:117-183  kcount_mi355::HashTableDriver<K>
//
// Same method names, argument meaning and error behaviour (no exceptions: the reference calls these
// from inside UPC++ progress; a failed device call prints and aborts like gpu_common.cpp:59-65 does).
// Differences, all deliberate:
//   * semantics are the CPU backend's (N counts as G inside a k-mer, counters saturate at 65535);
//   * the sender ships k-mer records binned by owner shard instead of 4-bit supermers, so
//     ParseAndPackDriver exposes records()/counts() where the reference exposes supermers/packed_seqs;
//     HashTableDriver still accepts the reference's 4-bit packed supermers (insert_supermer) for
//     mixed runs, and ASCII supermers as the CPU backend receives them (insert_supermer_ascii);
//   * the table never drops: num_dropped is always 0.
// Header-only; link with -lkcount_mi355 and the HIP runtime.
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/kcount_mi355.h"

namespace kcount_mi355 {

using count_t = uint32_t;

[[noreturn]] inline void die(const char *what, int status) {
  std::fprintf(stderr, "kcount_mi355: %s failed: %s (%d) %s\n", what, kc_error_string(status), status, kc_last_error());
  std::abort();
}
inline void check(int status, const char *what) {
  if (status != KC_OK) die(what, status);
}

// gpu_hash_table.hpp:64-67,69-75
struct CountExts {
  count_t count;
  int8_t left, right;
};

template <int MAX_K>
struct KmerArray {
  static const int N_LONGS = (MAX_K + 31) / 32;
  uint64_t longs[N_LONGS];
};

// gpu_hash_table.hpp:109-115
struct InsertStats {
  uint64_t dropped = 0;
  uint64_t dropped_qf = 0;
  uint64_t attempted = 0;
  uint64_t new_inserts = 0;
  uint64_t num_unique_qf = 0;
};

// ---- sender side ---------------------------------------------------------------------------------
class ParseAndPackDriver {
  kc_ctx *ctx = nullptr;
  int rank_n;
  int kmer_len;
  int num_longs;
  uint64_t *d_records = nullptr;  // rank_n segments of seg_capacity records
  uint64_t seg_capacity = 0;
  std::vector<uint64_t> h_counts;
  double t_func = 0;

 public:
  // max sequence block the reference accepts: KCOUNT_SEQ_BLOCK_SIZE (CMakeDefinitions.txt:63); here only a default
  static constexpr size_t SEQ_BLOCK_SIZE = 3000000;

  // parse_and_pack.hpp:81-82.  minimizer_len is accepted and ignored: ownership is a hash of the k-mer (F3 in SURVEY.md).
  ParseAndPackDriver(int upcxx_rank_me, int upcxx_rank_n, int qual_offset, int kmer_len, int num_kmer_longs, int /*minimizer_len*/,
                     double &init_time, int device = 0)
      : rank_n(upcxx_rank_n), kmer_len(kmer_len), num_longs(num_kmer_longs), h_counts(upcxx_rank_n, 0) {
    kc_config cfg{};
    cfg.kmer_len = kmer_len;
    cfg.qual_offset = qual_offset;
    cfg.dmin_thres = 2;
    cfg.device = device;
    cfg.rank_me = upcxx_rank_me;
    cfg.rank_n = upcxx_rank_n;
    cfg.max_kmers_buffered = 1 << 16;  // this context only extracts
    int st = 0;
    ctx = kc_create(&cfg, &st);
    if (!ctx) die("kc_create", st);
    if (num_kmer_longs != kc_num_longs(kmer_len)) die("num_kmer_longs mismatch", KC_ERR_INVALID_ARG);
    init_time = 0;
  }
  ~ParseAndPackDriver() {
    if (d_records) (void)hipFree(d_records);
    kc_destroy(ctx);
  }
  ParseAndPackDriver(const ParseAndPackDriver &) = delete;
  ParseAndPackDriver &operator=(const ParseAndPackDriver &) = delete;

  // parse_and_pack.cpp:281-319.  seqs: case-masked reads joined by '_'.  false if empty or shorter than k.
  // After it returns, counts()[t] records for target t start at records() + t * segment_capacity() * num_longs.
  bool process_seq_block(const std::string &seqs, unsigned int &num_valid_kmers) {
    num_valid_kmers = 0;
    if (seqs.empty() || (int)seqs.length() < kmer_len) return false;
    // worst case every k-mer goes to one target
    if (seqs.length() > seg_capacity) {
      if (d_records) (void)hipFree(d_records);
      seg_capacity = seqs.length();
      if (hipMalloc((void **)&d_records, seg_capacity * rank_n * num_longs * 8) != hipSuccess) die("hipMalloc(records)", KC_ERR_OUT_OF_MEMORY);
    }
    check(kc_extract_partition_seq_block(ctx, seqs.data(), seqs.size(), 0, d_records, seg_capacity, h_counts.data()),
          "kc_extract_partition_seq_block");
    uint64_t tot = 0;
    for (auto c : h_counts) tot += c;
    num_valid_kmers = (unsigned)tot;
    return true;
  }
  // the reference copies the block a second time to 4-bit pack it (parse_and_pack.cpp:321-336): nothing to do here
  void pack_seq_block(const std::string &) {}
  std::tuple<double, double> get_elapsed_times() { return {t_func, t_func}; }

  const uint64_t *records() const { return d_records; }  // device pointer
  uint64_t segment_capacity() const { return seg_capacity; }
  const std::vector<uint64_t> &counts() const { return h_counts; }

};

// ---- receiver side -------------------------------------------------------------------------------
template <int MAX_K>
class HashTableDriver {
  static const int N_LONGS = (MAX_K + 31) / 32;
  kc_ctx *ctx = nullptr;
  int kmer_len = 0;
  std::string elem_buff;  // ASCII supermers joined by '_' until a block is full (gpu_hash_table.cpp:681-695)
  std::vector<KmerArray<MAX_K>> output_keys;
  std::vector<CountExts> output_vals;
  size_t output_index = 0;
  InsertStats stats;
  int num_gpu_calls = 0;
  uint64_t final_capacity = 0;

  static char unpack_nibble(uint8_t v) {  // codes of parse_and_pack.cpp:196-213
    static const char to_base[10] = {'_', 'a', 'c', 'g', 't', 'A', 'C', 'G', 'T', 'N'};
    return v <= 9 ? to_base[v] : '_';
  }
  void flush_block() {
    if (elem_buff.empty()) return;
    check(kc_submit_seq_block(ctx, elem_buff.data(), elem_buff.size(), 0), "kc_submit_seq_block");
    num_gpu_calls++;
    elem_buff.clear();
  }

 public:
  static constexpr size_t HASHTABLE_BLOCK_SIZE = 1u << 24;  // role of KCOUNT_GPU_HASHTABLE_BLOCK_SIZE, much larger

  HashTableDriver() = default;
  ~HashTableDriver() { kc_destroy(ctx); }
  HashTableDriver(const HashTableDriver &) = delete;
  HashTableDriver &operator=(const HashTableDriver &) = delete;

  // gpu_hash_table.hpp:155-156.  gpu_avail_mem and use_qf are accepted and ignored (the filter is forced off in the
  // reference too, kcount_gpu.cpp:227-232); sizing problems come back through warnings like the reference's do.
  void init(int upcxx_rank_me, int upcxx_rank_n, int kmer_len_, size_t max_elems, size_t /*max_ctg_elems*/, size_t num_errors,
            size_t /*gpu_avail_mem*/, std::string &msgs, std::string &warnings, bool /*use_qf*/, int qual_offset = 33, int device = 0,
            int dmin_thres = 2) {
    kmer_len = kmer_len_;
    if (kc_num_longs(kmer_len) != N_LONGS) die("MAX_K does not match kmer_len", KC_ERR_INVALID_ARG);
    kc_config cfg{};
    cfg.kmer_len = kmer_len;
    cfg.qual_offset = qual_offset;
    cfg.dmin_thres = dmin_thres;
    cfg.device = device;
    cfg.rank_me = upcxx_rank_me;
    cfg.rank_n = upcxx_rank_n;
    cfg.max_elems = max_elems + num_errors;
    // the reference's max_elems is its k-mer estimate divided by the assumed depth of 4 (kmer_dht.cpp:126-127)
    cfg.max_kmers_buffered = (uint64_t)max_elems * 4 + (1u << 20);
    int st = 0;
    ctx = kc_create(&cfg, &st);
    if (!ctx) die("kc_create", st);
    msgs += "kcount_mi355: k=" + std::to_string(kmer_len) + " words=" + std::to_string(N_LONGS) + "\n";
    (void)warnings;
  }
  void init_ctg_kmers(uint64_t, size_t) {}  // contig pass: dead in the proxy (F8 in SURVEY.md)

  // gpu_hash_table.cpp:681-695: one 4-bit packed supermer as kcount_gpu.cpp:153-161 cuts it (odd nibbles masked to 0)
  void insert_supermer(const std::string &packed, count_t /*supermer_count*/) {
    std::string s;
    s.reserve(packed.size() * 2);
    for (unsigned char b : packed) {
      s.push_back(unpack_nibble(b >> 4));
      s.push_back(unpack_nibble(b & 15));
    }
    insert_supermer_ascii(s);
  }
  // the CPU backend's wire format (kcount_cpu.cpp:477-493): ASCII, case = quality.  '_' bytes only separate.
  void insert_supermer_ascii(const std::string &seq) {
    if (elem_buff.size() + seq.size() + 1 >= HASHTABLE_BLOCK_SIZE) flush_block();
    elem_buff += seq;
    elem_buff += '_';
    stats.attempted++;
  }
  // records that another shard's ParseAndPackDriver binned for this shard (device pointer)
  void insert_records(const uint64_t *d_records, uint64_t n) {
    check(kc_insert_records(ctx, d_records, n), "kc_insert_records");
    num_gpu_calls++;
  }
  void flush_inserts() {
    flush_block();
    check(kc_flush(ctx), "kc_flush");
  }
  void done_ctg_kmer_inserts(uint64_t &attempted, uint64_t &dropped, uint64_t &new_inserts) { attempted = dropped = new_inserts = 0; }

  // gpu_hash_table.cpp:736-784: purge + compact + copy back
  void done_all_inserts(uint64_t &num_dropped, uint64_t &num_unique, uint64_t &num_purged) {
    flush_block();
    kc_result r;
    check(kc_finalize(ctx, &r), "kc_finalize");
    kc_stats st;
    check(kc_get_stats(ctx, &st), "kc_get_stats");
    num_dropped = 0;
    num_unique = st.num_unique;
    num_purged = st.num_purged;
    stats.new_inserts = st.num_unique;
    final_capacity = r.n;
    std::vector<uint64_t> keys(r.n * N_LONGS);
    std::vector<uint16_t> counts(r.n);
    std::vector<uint8_t> left(r.n), right(r.n);
    check(kc_copy_results(ctx, keys.data(), counts.data(), left.data(), right.data()), "kc_copy_results");
    output_keys.resize(r.n);
    output_vals.resize(r.n);
    for (uint64_t i = 0; i < r.n; i++) {
      for (int j = 0; j < N_LONGS; j++) output_keys[i].longs[j] = keys[i * N_LONGS + j];
      output_vals[i] = {counts[i], (int8_t)left[i], (int8_t)right[i]};
    }
  }
  void begin_iterate() { output_index = 0; }
  // {nullptr, nullptr} at the end; unlike the reference no empty slots are returned (count is never 0)
  std::pair<KmerArray<MAX_K> *, CountExts *> get_next_entry() {
    if (output_index >= output_keys.size()) return {nullptr, nullptr};
    output_index++;
    return {&output_keys[output_index - 1], &output_vals[output_index - 1]};
  }
  void get_elapsed_time(double &insert_time, double &kernel_time) { insert_time = kernel_time = 0; }
  int64_t get_capacity() {
    kc_stats st;
    check(kc_get_stats(ctx, &st), "kc_get_stats");
    return (int64_t)st.capacity;
  }
  int64_t get_final_capacity() { return (int64_t)final_capacity; }
  InsertStats &get_stats() { return stats; }
  int get_num_gpu_calls() { return num_gpu_calls; }
  double get_qf_load_factor() { return 0; }
  kc_ctx *handle() { return ctx; }
};

}  // namespace kcount_mi355
