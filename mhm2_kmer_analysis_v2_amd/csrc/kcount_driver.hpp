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
//   * semantics are the CPU backend's (N counts as G inside a k-mer, counters saturate at 65535), also for the supermers:
//     they are SeqBlockInserter::process_seq's (kcount_cpu.cpp:73-103), cut and 4-bit packed on the device;
//   * ParseAndPackDriver has two modes.  Constructed with the reference's own arguments it is wire compatible: it fills
//     `supermers` and `packed_seqs` exactly as kcount_gpu.cpp:135-164 reads them, targets being the reference's
//     KmerDHT::get_kmer_target_rank.  With records_mode = true it bins k-mer records by owner shard instead
//     (records()/counts()), the native flow that csrc/kc_exchange.hpp ships over RCCL;
//   * HashTableDriver takes the reference's 4-bit packed supermers (insert_supermer; unpacked on the device), ASCII
//     supermers as the CPU backend receives them (insert_supermer_ascii) and records (insert_records);
//   * the table never drops: num_dropped is always 0.
// Header-only; link with -lkcount_mi355 and the HIP runtime.
#pragma once
#include <hip/hip_runtime_api.h>

#include <array>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <tuple>
#include <utility>
#include <vector>

#include "../../include/kcount_mi355.h"

namespace kcount_mi355 {

// gpu_hash_table.hpp:54-57
enum PASS_TYPE { READ_KMERS_PASS = 0, CTG_KMERS_PASS = 1 };

using count_t = uint32_t;
using ext_count_t = uint16_t;

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
static_assert(sizeof(CountExts) == sizeof(kc_count_exts), "CountExts must have the layout of kc_count_exts");

template <int MAX_K>
struct KmerArray {
  static const int N_LONGS = (MAX_K + 31) / 32;
  uint64_t longs[N_LONGS];

  void set(const uint64_t *x) {  // gpu_hash_table.hpp:74
    for (int i = 0; i < N_LONGS; i++) longs[i] = x[i];
  }
};

// gpu_hash_table.hpp:109-115
struct InsertStats {
  uint64_t dropped = 0;
  uint64_t dropped_qf = 0;
  uint64_t attempted = 0;
  uint64_t new_inserts = 0;
  uint64_t num_unique_qf = 0;
};

// parse_and_pack.hpp:50-54 (= kc_supermer of the C ABI)
struct SupermerInfo {
  int target;
  int offset;
  uint16_t len;
};
static_assert(sizeof(SupermerInfo) == sizeof(kc_supermer), "SupermerInfo must have the layout of kc_supermer");

struct StopWatch {  // role of the reference's GPUTimer / chrono timers (gpu_common.hpp:83-105)
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  double stop() const { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); }
};

// ---- sender side ---------------------------------------------------------------------------------
class ParseAndPackDriver {
  kc_ctx *ctx = nullptr;
  int rank_n;
  int kmer_len;
  int rec_longs;   // of a record on the shard wire (kc_record_longs)
  bool records_mode;
  uint64_t *d_records = nullptr;  // rank_n segments of seg_capacity records
  uint64_t seg_capacity = 0;
  std::vector<uint64_t> h_counts;
  std::vector<uint8_t> packed_block;  // the whole block, 4-bit packed, as process_seq_block left it
  double t_func = 0, t_kernel = 0;

 public:
  // max sequence block the reference accepts: KCOUNT_SEQ_BLOCK_SIZE (CMakeDefinitions.txt:63)
  static constexpr size_t SEQ_BLOCK_SIZE = 3000000;

  // PUBLIC result members read by the host (parse_and_pack.hpp:78-79, src/kcount/kcount_gpu.cpp:135-164)
  std::vector<SupermerInfo> supermers;
  std::string packed_seqs;

  // parse_and_pack.hpp:81-82.  minimizer_len must be the reference's own choice for this k (kmer_dht.cpp:117-119): the
  // targets are KmerDHT::get_kmer_target_rank's.  records_mode: bin k-mer records by owner shard instead of building
  // supermers (the native flow; ownership is then a hash of the k-mer, F3 in SURVEY.md).
  ParseAndPackDriver(int upcxx_rank_me, int upcxx_rank_n, int qual_offset, int kmer_len, int num_kmer_longs, int minimizer_len,
                     double &init_time, bool records_mode = false, int device = 0)
      : rank_n(upcxx_rank_n), kmer_len(kmer_len), rec_longs(kc_record_longs(kmer_len)),
        records_mode(records_mode), h_counts(upcxx_rank_n, 0) {
    StopWatch sw;
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
    int m = kmer_len * 2 / 3 + 1;  // kmer_dht.cpp:117-119
    m = m < 15 ? 15 : m > 27 ? 27 : m;
    if (!records_mode && minimizer_len != (m < kmer_len ? m : kmer_len)) die("minimizer_len is not the reference's choice for this k", KC_ERR_INVALID_ARG);
    init_time = sw.stop();
  }
  ~ParseAndPackDriver() {
    if (d_records) (void)hipFree(d_records);
    kc_destroy(ctx);
  }
  ParseAndPackDriver(const ParseAndPackDriver &) = delete;
  ParseAndPackDriver &operator=(const ParseAndPackDriver &) = delete;

  // parse_and_pack.cpp:281-319.  seqs: case-masked reads joined by '_'.  false if empty, shorter than k or too long.
  // Wire mode: fills `supermers`.  Records mode: counts()[t] records for target t start at
  // records() + t * segment_capacity() * record_longs().
  bool process_seq_block(const std::string &seqs, unsigned int &num_valid_kmers) {
    StopWatch sw;
    num_valid_kmers = 0;
    if (seqs.empty() || (int)seqs.length() < kmer_len || seqs.length() >= (1ull << 31)) return false;
    if (!records_mode) {
      supermers.resize(seqs.length());  // one per k-mer at most
      packed_block.resize((seqs.length() + 1) / 2);
      uint32_t n = 0;
      StopWatch kw;
      check(kc_build_supermers(ctx, seqs.data(), seqs.size(), 0, reinterpret_cast<kc_supermer *>(supermers.data()),
                               (uint32_t)supermers.size(), &n, &num_valid_kmers, packed_block.data()),
            "kc_build_supermers");
      t_kernel += kw.stop();
      supermers.resize(n);
      t_func += sw.stop();
      return true;
    }
    // worst case every k-mer goes to one target
    if (seqs.length() > seg_capacity) {
      if (d_records) (void)hipFree(d_records);
      seg_capacity = seqs.length();
      if (hipMalloc((void **)&d_records, seg_capacity * rank_n * rec_longs * 8) != hipSuccess) die("hipMalloc(records)", KC_ERR_OUT_OF_MEMORY);
    }
    StopWatch kw;
    check(kc_extract_partition_seq_block(ctx, seqs.data(), seqs.size(), 0, d_records, seg_capacity, h_counts.data()),
          "kc_extract_partition_seq_block");
    t_kernel += kw.stop();
    uint64_t tot = 0;
    for (auto c : h_counts) tot += c;
    num_valid_kmers = (unsigned)tot;
    t_func += sw.stop();
    return true;
  }
  // parse_and_pack.cpp:321-336: the packed block (made on the device together with the supermers) becomes packed_seqs
  void pack_seq_block(const std::string &seqs) {
    StopWatch sw;
    if (!records_mode) packed_seqs.assign(reinterpret_cast<const char *>(packed_block.data()), (seqs.length() + 1) / 2);
    t_func += sw.stop();
  }
  // parse_and_pack.cpp:338: {time in the driver's functions, time in device calls}
  std::tuple<double, double> get_elapsed_times() { return {t_func, t_kernel}; }

  const uint64_t *records() const { return d_records; }  // device pointer
  uint64_t segment_capacity() const { return seg_capacity; }
  int record_longs() const { return rec_longs; }
  const std::vector<uint64_t> &counts() const { return h_counts; }
};

// ---- receiver side -------------------------------------------------------------------------------
template <int MAX_K>
class HashTableDriver {
  static const int N_LONGS = (MAX_K + 31) / 32;
  kc_ctx *ctx = nullptr;
  int kmer_len = 0;
  std::string elem_buff;    // ASCII supermers joined by '_' until a block is full (gpu_hash_table.cpp:681-695)
  std::string packed_buff;  // 4-bit packed supermers joined by the byte '_', as the reference's elem_buff_host.seqs
  std::vector<KmerArray<MAX_K>> output_keys;
  std::vector<CountExts> output_vals;
  size_t output_index = 0;
  InsertStats stats;
  int num_gpu_calls = 0;
  uint64_t final_capacity = 0;
  double t_insert = 0;

  std::string ctg_seqs;               // contig pass: the supermers as characters, '_'-joined ...
  std::vector<uint16_t> ctg_depths;   // ... and the depth of every character's contig
  InsertStats ctg_stats;

  void flush_ctg_block() {
    if (ctg_seqs.empty()) return;
    StopWatch sw;
    check(kc_submit_ctg_block(ctx, ctg_seqs.data(), ctg_depths.data(), ctg_seqs.size(), 0), "kc_submit_ctg_block");
    num_gpu_calls++;
    ctg_seqs.clear();
    ctg_depths.clear();
    t_insert += sw.stop();
  }

  void flush_block() {
    StopWatch sw;
    if (!elem_buff.empty()) {
      check(kc_submit_seq_block(ctx, elem_buff.data(), elem_buff.size(), 0), "kc_submit_seq_block");
      num_gpu_calls++;
      elem_buff.clear();
    }
    if (!packed_buff.empty()) {
      check(kc_submit_packed_supermers(ctx, reinterpret_cast<const uint8_t *>(packed_buff.data()), packed_buff.size(), 0),
            "kc_submit_packed_supermers");
      num_gpu_calls++;
      packed_buff.clear();
    }
    t_insert += sw.stop();
  }

 public:
  static constexpr size_t HASHTABLE_BLOCK_SIZE = 1u << 24;  // role of KCOUNT_GPU_HASHTABLE_BLOCK_SIZE, much larger

  // gpu_hash_table.hpp:150: read by the host (kcount_gpu.cpp:315,330,376).  The contig pass is dead in the proxy
  // (SURVEY.md F8): nothing here ever sets CTG_KMERS_PASS.
  PASS_TYPE pass_type = READ_KMERS_PASS;

  HashTableDriver() = default;
  ~HashTableDriver() { kc_destroy(ctx); }
  HashTableDriver(const HashTableDriver &) = delete;
  HashTableDriver &operator=(const HashTableDriver &) = delete;
  // the reference's host constructs its member from a temporary (kcount_gpu.cpp:218-219, `ht_gpu_driver({})`): the
  // context moves with the driver
  HashTableDriver(HashTableDriver &&o) noexcept
      : ctx(o.ctx), kmer_len(o.kmer_len), elem_buff(std::move(o.elem_buff)), packed_buff(std::move(o.packed_buff)),
        output_keys(std::move(o.output_keys)), output_vals(std::move(o.output_vals)), output_index(o.output_index), stats(o.stats),
        num_gpu_calls(o.num_gpu_calls), final_capacity(o.final_capacity), t_insert(o.t_insert), ctg_seqs(std::move(o.ctg_seqs)),
        ctg_depths(std::move(o.ctg_depths)), ctg_stats(o.ctg_stats), pass_type(o.pass_type) {
    o.ctx = nullptr;
  }

  // gpu_hash_table.hpp:155-156.  use_qf is accepted and ignored (the filter is forced off in the reference too,
  // kcount_gpu.cpp:227-232).  Sizing problems come back through `warnings` like the reference's do
  // (gpu_hash_table.cpp:575-578): when the k-mer buffer for max_elems would not fit gpu_avail_mem it is made smaller,
  // and what does not fit it later goes through the global table (slower, nothing is dropped).
  void init(int upcxx_rank_me, int upcxx_rank_n, int kmer_len_, size_t max_elems, size_t /*max_ctg_elems*/, size_t num_errors,
            size_t gpu_avail_mem, std::string &msgs, std::string &warnings, bool /*use_qf*/, int qual_offset = 33, int device = 0,
            int dmin_thres = 2) {
    kmer_len = kmer_len_;
    if (kc_num_longs(kmer_len) != N_LONGS) die("MAX_K does not match kmer_len", KC_ERR_INVALID_ARG);
    kc_config cfg{};
    cfg.kmer_len = kmer_len;
    cfg.qual_offset = qual_offset;
    cfg.dmin_thres = dmin_thres;
    cfg.device = device;
    // the context inserts whatever it is handed, like the reference's receiver: the sender has already decided that
    // this rank is the target, by whichever partition function
    (void)upcxx_rank_me;
    (void)upcxx_rank_n;
    cfg.rank_me = 0;
    cfg.rank_n = 1;
    cfg.max_elems = max_elems + num_errors;
    cfg.flags = KC_FLAG_TIME_KERNELS;  // get_elapsed_time reports device time
    // the reference's max_elems is its k-mer estimate divided by the assumed depth of 4 (kmer_dht.cpp:126-127)
    uint64_t buffered = (uint64_t)max_elems * 4 + (1u << 20);
    // bytes per buffered occurrence: a level-1 and a level-2 record, the overflow lists, chain tables
    const double per_kmer = N_LONGS * 8 * (N_LONGS == 1 ? 2.1 : 2.4) + 2.0;
    const double need = (double)buffered * per_kmer;
    if (gpu_avail_mem && need > 0.8 * (double)gpu_avail_mem) {
      const double ratio = 0.8 * (double)gpu_avail_mem / need;
      char line[320];
      std::snprintf(line, sizeof(line),
                    "Insufficient memory for a k-mer buffer of %llu occurrences (%.1f GB of %.1f GB available); reducing it to %.3f of "
                    "that: the rest goes through the slower global table (nothing is dropped)\n",
                    (unsigned long long)buffered, need / 1e9, (double)gpu_avail_mem / 1e9, ratio);
      warnings += line;
      buffered = (uint64_t)((double)buffered * ratio);
      if (buffered < (1u << 16)) buffered = 1u << 16;
    }
    cfg.max_kmers_buffered = buffered;
    int st = 0;
    ctx = kc_create(&cfg, &st);
    if (!ctx) die("kc_create", st);
    msgs += "kcount_mi355: k=" + std::to_string(kmer_len) + " words=" + std::to_string(N_LONGS) + ", k-mer buffer for " +
            std::to_string(buffered) + " occurrences (" + std::to_string((uint64_t)((double)buffered * per_kmer / 1e6)) + " MB)\n";
  }
  // gpu_hash_table.hpp:158: from here on insert_supermer carries contig k-mers with the contig's depth as their count
  // (insert_supermer_from_ctg, kcount_cpu.cpp:357-407).  Dead in the proxy, complete here: csrc/kc_ctg.hpp.
  void init_ctg_kmers(uint64_t max_elems, size_t /*gpu_avail_mem*/) {
    flush_block();
    pass_type = CTG_KMERS_PASS;
    check(kc_begin_ctg_kmers(ctx, max_elems), "kc_begin_ctg_kmers");
  }

  // gpu_hash_table.cpp:681-695: one 4-bit packed supermer as kcount_gpu.cpp:153-161 cuts it (odd nibbles masked to 0);
  // the bytes are joined by '_' exactly like the reference's elem_buff_host and unpacked on the device
  // supermer_count: every k-mer of the supermer is seen that many times (insert_supermer_from_read adds `count` to the
  // k-mer and to its extensions, kcount_cpu.cpp:349-353).  Reads always come with 1 (kcount.cpp:87 passes depth 0,
  // kcount_gpu.cpp:160 turns it into 1); a larger count is submitted as that many copies, which saturate exactly like
  // the reference's min(count + n, 65535) chain does.
  void insert_supermer(const std::string &packed, count_t supermer_count) {
    if (pass_type == CTG_KMERS_PASS) {
      // back to characters (the nibble codes of parse_and_pack.cpp:196-213), every one with the supermer's count as its
      // depth: the block kc_submit_ctg_block takes
      static const char to_base[16] = {'_', 'a', 'c', 'g', 't', 'A', 'C', 'G', 'T', 'N', '_', '_', '_', '_', '_', '_'};
      if (ctg_seqs.size() + 2 * packed.size() + 1 >= HASHTABLE_BLOCK_SIZE) flush_ctg_block();
      for (unsigned char b : packed) {
        ctg_seqs += to_base[b >> 4];
        ctg_seqs += to_base[b & 15];
      }
      ctg_seqs += '_';
      ctg_depths.resize(ctg_seqs.size(), (uint16_t)(supermer_count > 65535 ? 65535 : supermer_count));
      ctg_stats.attempted++;
      return;
    }
    const count_t copies = supermer_count ? supermer_count : 1;
    for (count_t c = 0; c < copies; c++) {
      if (packed_buff.size() + packed.size() + 1 >= HASHTABLE_BLOCK_SIZE) flush_block();
      packed_buff += packed;
      packed_buff += '_';
    }
    stats.attempted++;
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
    StopWatch sw;
    check(kc_insert_records(ctx, d_records, n), "kc_insert_records");
    num_gpu_calls++;
    t_insert += sw.stop();
  }
  void flush_inserts() {
    flush_block();
    flush_ctg_block();
    check(kc_flush(ctx), "kc_flush");
  }
  // gpu_hash_table.cpp:697-734: the contig k-mers are in (they join the results in done_all_inserts)
  void done_ctg_kmer_inserts(uint64_t &attempted, uint64_t &dropped, uint64_t &new_inserts) {
    flush_ctg_block();
    attempted = ctg_stats.attempted;
    dropped = 0;  // (the contig table never drops: a block that does not fit ends in KC_ERR_CAPACITY)
    uint64_t distinct = 0;
    check(kc_ctg_stats(ctx, &distinct, nullptr), "kc_ctg_stats");
    new_inserts = ctg_stats.new_inserts = distinct;  // the distinct contig k-mers this rank's table took
  }

  // gpu_hash_table.cpp:736-784: purge + compact + copy back
  void done_all_inserts(uint64_t &num_dropped, uint64_t &num_unique, uint64_t &num_purged) {
    flush_block();
    flush_ctg_block();
    kc_result r;
    check(kc_finalize(ctx, &r), "kc_finalize");
    kc_stats st;
    check(kc_get_stats(ctx, &st), "kc_get_stats");
    num_dropped = 0;
    num_unique = st.num_unique;
    num_purged = st.num_purged;
    stats.new_inserts = st.num_unique;
    final_capacity = r.n;
    // the two arrays get_next_entry points into are filled by one device-to-host copy each: a KmerArray is its N_LONGS
    // words, a CountExts is packed on the device (kc_copy_results_entries)
    static_assert(sizeof(KmerArray<MAX_K>) == N_LONGS * 8, "KmerArray is its words");
    output_keys.resize(r.n);
    output_vals.resize(r.n);
    check(kc_copy_results_entries(ctx, reinterpret_cast<uint64_t *>(output_keys.data()), reinterpret_cast<kc_count_exts *>(output_vals.data())),
          "kc_copy_results_entries");
  }
  void begin_iterate() { output_index = 0; }
  // {nullptr, nullptr} at the end; unlike the reference no empty slots are returned (count is never 0)
  std::pair<KmerArray<MAX_K> *, CountExts *> get_next_entry() {
    if (output_index >= output_keys.size()) return {nullptr, nullptr};
    output_index++;
    return {&output_keys[output_index - 1], &output_vals[output_index - 1]};
  }
  // gpu_hash_table.hpp:172: {wall time in the insert path, device time of the kernels it launched}
  void get_elapsed_time(double &insert_time, double &kernel_time) {
    insert_time = t_insert;
    kernel_time = 0;
    kc_kernel_time kt[16];
    int n = 0;
    if (kc_get_kernel_times(ctx, kt, 16, &n) == KC_OK)
      for (int i = 0; i < n && i < 16; i++) kernel_time += kt[i].total_ms * 1e-3;
  }
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
