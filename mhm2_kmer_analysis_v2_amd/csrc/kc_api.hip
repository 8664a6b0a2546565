// kc_api.hip -- host side of libkcount_mi355.so: the C ABI of include/kcount_mi355.h.
//
// Plays the role of the reference's two device drivers, ParseAndPackGPUDriver
// (src/kcount/kcount-gpu/parse_and_pack.cpp:239-338) and HashTableGPUDriver
// (src/kcount/kcount-gpu/gpu_hash_table.cpp:519-859), with the CPU backend's semantics.
// No exceptions and no abort(): every failure is a status code (the reference aborts inside
// UPC++ progress, src/gpu-utils/gpu_common.cpp:59-65).
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <new>
#include <thread>
#include <vector>

#include "../../include/kcount_mi355.h"
#include "kc_bucketed.hpp"
#include "kc_shard.hpp"
#include "kc_wire6.hpp"
#include "kc_supermer.hpp"
#include "kc_ctg.hpp"

using namespace kc;

static thread_local char g_last_error[512] = "";

static int hip_fail(hipError_t e, const char *what, int line) {
  snprintf(g_last_error, sizeof(g_last_error), "%s failed at kc_api.hip:%d: %s", what, line, hipGetErrorString(e));
  return e == hipErrorOutOfMemory ? KC_ERR_OUT_OF_MEMORY : KC_ERR_HIP;
}

#define HIPCHK(call)                                          \
  do {                                                        \
    hipError_t e_ = (call);                                   \
    if (e_ != hipSuccess) return hip_fail(e_, #call, __LINE__); \
  } while (0)

enum { KT_EXTRACT_INSERT = 0, KT_EXTRACT_BIN, KT_INSERT_RECORDS, KT_FINALIZE, KT_TILE_FIRST, KT_REHASH, KT_L1_READS, KT_L1_RECORDS,
       KT_L2_SPLIT, KT_COUNT_REGIONS, KT_FALLBACK, KT_SHARD_PACK, KT_L1_READS_UQ, KT_L1_READS16, KT_L2_REC6, KT_BIN16, KT_L1_WIRE6, KT_COUNT };
static const char *const kt_names[KT_COUNT] = {"kc_extract_kernel<insert>", "kc_bin_reads_kernel", "kc_insert_records_kernel",
                                               "kc_finalize_kernel", "kc_tile_first_kernel", "kc_rehash_kernel",
                                               "kc_l1_reads_kernel", "kc_l1_records_kernel", "kc_l2_split_kernel",
                                               "kc_count_kernel", "kc_flagged_to_table_kernel", "kc_shard_pack_kernel",
                                               "kc_l1_reads_kernel<byte-loaded qualities>", "kc_l1_reads16_kernel", "kc_l2_rec6_kernel", "kc_bin16_kernel", "kc_l1_wire6_kernel"};
struct kt_pending {
  hipEvent_t start, stop;
  int kind;
};

struct kc_ctx {
  kc_config cfg;
  std::vector<kt_pending> kt_pend;
  uint64_t kt_launches[KT_COUNT];
  double kt_ms[KT_COUNT];
  int k, nl;   // nl: words of a k-mer inside the library (kc_record_longs)
  int nl_ext;  // the reference's width of the same k-mer (kc_num_longs): results, dumps, lookups
  hipStream_t own_stream, stream;
  // table arena: keys then vals
  uint8_t *arena;
  size_t arena_bytes;
  uint64_t capacity;
  Table table;
  uint64_t *d_ctrs;
  uint64_t *h_ctrs;  // pinned mirror
  uint64_t *d_tile_first;
  uint64_t *d_out_plan;  // hole-closing plan + per-workgroup tails of the block-wise result output
  size_t tile_first_cap;
  // staging for host-resident input (single slot: the '_'-joined blocks of kc_submit_seq_block)
  uint8_t *d_stage_bases, *d_stage_quals;
  uint64_t *d_stage_offsets;
  size_t stage_bytes, stage_reads;
  // host-resident reads: two device slots and two pinned host slots, copies on a stream of their own (host_pipe_*)
  struct {
    uint8_t *d_bases[2], *d_quals[2], *h_bases[2], *h_quals[2];
    uint64_t *d_offs[2], *h_offs[2];
    hipEvent_t copied[2], consumed[2], half[2];
    uint32_t lead_b[2], lead_q[2];  // bytes in front of a slot's first read (the copy started on a page boundary of the source)
    bool used[2];
    hipStream_t copy_stream, copy_stream2;  // a block's bytes cross PCIe as two concurrent copies (two DMA engines)
    size_t cap_bytes, cap_reads;
    bool ready;
  } hp;
  // results
  uint64_t *d_out_keys;
  uint64_t *d_out_keys_ext;  // the results' keys at the reference's width, where that differs (made by kc_finalize)
  uint16_t *d_out_counts;
  uint8_t *d_out_left, *d_out_right;
  uint64_t out_cap, out_n;
  bool finalized;
  uint32_t *d_index;  // lookup index over the results (built on first kc_lookup)
  uint64_t index_cap;
  // the contig pass (kc_ctg.hpp): a table of the contigs' k-mers, merged into the results by kc_finalize
  Table ctg_table;
  uint64_t ctg_cap;
  uint64_t *d_ctg_status;  // [0] a character outside the alphabet was seen, [1] entries
  uint64_t ctg_attempted, ctg_new;
  double arena_probe_tbps;  // rate of level 1's write pattern on the arena pick_fast_arena chose (0: no probe ran)
  kc_synth_table *d_synth;
  // scratch of the reference-wire entry points (kc_build_supermers, kc_submit_packed_supermers)
  uint8_t *d_sm_bytes;    // block / unpacked block
  uint8_t *d_sm_packed;   // packed block / packed supermers
  int32_t *d_sm_targets;
  SupermerInfo *d_sm_out;
  uint32_t *d_sm_ctr;     // n_out, n_kmers, too_long, + a u64 "bad character" flag behind them
  size_t sm_bytes_cap, sm_packed_cap, sm_targets_cap, sm_out_cap;
  // host-side stats
  uint64_t num_reads, num_bases, num_gpu_calls;
  uint64_t purged, sum_counts, unique_at_finalize;
  // bucketed path (kc_bucketed.hpp)
  kc_tuning tuning;
  bool bk_ready;      // buffers allocated for the current geometry
  bool bk_level2;     // regions built for the buffered records
  bool bk_flagged;    // flagged regions and overflow records already moved to the global table
  bool table_mode;    // everything goes through the global table from now on
  bool started;       // something was submitted since create/reset
  Geom gm;
  BucketBufs bb;
  // arrays of an earlier geometry of this context (another k of a sweep), kept so that a later one can take them over
  // instead of allocating again: [i] pairs with the i-th pointer of BucketBufs (bk_slots)
  struct { void *p; size_t bytes; } bk_pool[13];
  size_t bk_held[13];    // bytes behind the pointers bb holds now
  // what the host knows without asking the device (a question is a device-to-host copy and a wait for the stream: the
  // host pipe asked twice per block, and its copies stood still meanwhile)
  uint64_t expect_host;  // == the device's CTR_EXPECT whenever expect_host_ok
  bool expect_host_ok;
  uint64_t ovf1_ub;      // upper bound of the records in the level-1 overflow list: positions launched since it was last read
  bool inc_on;           // level 2 has begun in instalments (kc_l2_split_kernel<..., INC>): bb.done1 / used2 / cnt2 carry its state
  uint64_t *d_cb, *h_cb;
  bool bk_spilled;       // earlier buffer-fulls of this pass were counted and merged into the global table (bk_spill_pass)
  bool wire6;            // KC_FLAG_WIRE_UNITS and a geometry of six-byte records: kc_extract_partition / kc_insert_records speak kc_wire6.hpp's units
  uint64_t *d_w6cur;     // ... the sender's cursors, one per (destination, piece); h_w6cur: their host copy
  uint64_t h_w6cur[WIRE6_MAX_SHARDS << WIRE6_MAX_LG_PIECES];
  bool l1_dropped;       // earlier buffer-fulls of this pass went through level 2 and left level 1 (bk_light_spill): level 2 holds them
  uint64_t l2_held;      // ... that many records (an upper bound)
  uint8_t *d_l2snap;     // level 2's state before an instalment that cannot be run again as a whole pass (l2_snapshot)
  size_t l2snap_bytes;
  bool l2snap_fresh;     // the snapshot is of a level 2 that had not begun (inc_on was false)
  uint64_t l2_per_bucket;  // records every bucket's part of the level-2 arena has room for while level 2 runs in instalments
  uint64_t expect_base;  // CTR_EXPECT when the buffer was last emptied: what is buffered now is CTR_EXPECT - expect_base
  uint64_t expect_prev;  // CTR_EXPECT after the previous block
  uint64_t bk_capacity;  // records the level-1 segments are sized for
  uint64_t bk_buffered;  // upper bound of records buffered so far (positions submitted)
  uint32_t bk_rot;       // first writer of the next level-1 launch
  size_t bk_bytes;
  int num_cus;
  // shard flow (kc_shard.hpp): ownership by level-1 bucket
  struct ShardExtent {
    uint64_t *p;
    uint64_t cap, used;  // words
  };
  struct {
    bool flow;         // this pass runs the shard flow (kc_shard_extract / kc_shard_commit were used)
    bool extracting;   // inside kc_shard_extract: level 1 keeps every k-mer, whoever owns it
    uint64_t *d_plan;  // off[PMAX], totals[SHARD_MAX], flags[SHARD_MAX], loose[SHARD_MAX], wtotals[SHARD_MAX]
    uint64_t *h_plan;  // pinned mirror of totals, flags, loose, wtotals (+ SHARD_MAX header words)
    uint32_t F;        // flat sources (received segments) of this pass
    uint32_t nbo;      // buckets this shard owns (row length of d_cnt / d_at)
    uint32_t *d_cnt;   // [FLAT_MAX][nbo]
    uint64_t *d_at;    // [FLAT_MAX][nbo]
    uint64_t sent, received;  // records, this pass
  } sh;
  std::vector<ShardExtent> sh_extents;  // where received segments live until the regions are built (kc_shard_reserve)
};

// ---- kernel timing (HIP events on the launch stream) --------------------------------------------
static void bk_free(kc_ctx *c, bool keep);
static void host_pipe_free(kc_ctx *c);
static void shard_free(kc_ctx *c);

struct KernelTimer {
  kc_ctx *c;
  kt_pending p;
  bool on;
  KernelTimer(kc_ctx *ctx, int kind);
  ~KernelTimer();
};

// ---- small helpers -----------------------------------------------------------------------------
KernelTimer::KernelTimer(kc_ctx *ctx, int kind) : c(ctx), on(false) {
  ctx->num_gpu_calls++;
  if (!(ctx->cfg.flags & KC_FLAG_TIME_KERNELS)) return;
  p.kind = kind;
  if (hipEventCreate(&p.start) != hipSuccess) return;
  if (hipEventCreate(&p.stop) != hipSuccess) {
    (void)hipEventDestroy(p.start);
    return;
  }
  (void)hipEventRecord(p.start, ctx->stream);
  on = true;
}

KernelTimer::~KernelTimer() {
  if (!on) return;
  (void)hipEventRecord(p.stop, c->stream);
  c->kt_pend.push_back(p);
}

static void drain_kernel_times(kc_ctx *c) {  // stream must be idle
  for (auto &p : c->kt_pend) {
    float ms = 0;
    if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
      c->kt_ms[p.kind] += ms;
      c->kt_launches[p.kind]++;
    }
    (void)hipEventDestroy(p.start);
    (void)hipEventDestroy(p.stop);
  }
  c->kt_pend.clear();
}

static uint64_t next_pow2(uint64_t v) {
  uint64_t p = 1;
  while (p < v) p <<= 1;
  return p;
}

static size_t table_bytes_for(uint64_t capacity, int nl) { return (size_t)capacity * ((size_t)nl * 8 + 36); }

static void carve_table(kc_ctx *c) {
  c->table.keys = (uint64_t *)c->arena;
  c->table.vals = (uint32_t *)(c->arena + (size_t)c->capacity * c->nl * 8);
  c->table.mask = c->capacity - 1;
}

static int clear_table(kc_ctx *c) {
  HIPCHK(hipMemsetAsync(c->table.keys, 0xFF, (size_t)c->capacity * c->nl * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->table.vals, 0, (size_t)c->capacity * 36, c->stream));
  return KC_OK;
}

static int sync_ctrs(kc_ctx *c) {
  HIPCHK(hipMemcpyAsync(c->h_ctrs, c->d_ctrs, CTR_COUNT * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}

template <int NL>
static int grow_table_nl(kc_ctx *c, uint64_t new_capacity) {
  uint8_t *na = nullptr;
  size_t nb = table_bytes_for(new_capacity, NL);
  HIPCHK(hipMalloc((void **)&na, nb));
  Table nt;
  nt.keys = (uint64_t *)na;
  nt.vals = (uint32_t *)(na + (size_t)new_capacity * NL * 8);
  nt.mask = new_capacity - 1;
  HIPCHK(hipMemsetAsync(nt.keys, 0xFF, (size_t)new_capacity * NL * 8, c->stream));
  HIPCHK(hipMemsetAsync(nt.vals, 0, (size_t)new_capacity * 36, c->stream));
  const uint64_t nblk = (c->capacity + 255) / 256;
  {
    KernelTimer kt(c, KT_REHASH);
    hipLaunchKernelGGL(kc_rehash_kernel<NL>, dim3((unsigned)nblk), dim3(256), 0, c->stream, c->table, c->capacity, nt);
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  HIPCHK(hipFree(c->arena));
  c->arena = na;
  c->arena_bytes = nb;
  c->capacity = new_capacity;
  c->table = nt;
  return KC_OK;
}

static int grow_table(kc_ctx *c, uint64_t new_capacity) {
  switch (c->nl) {
    case 1: return grow_table_nl<1>(c, new_capacity);
    case 2: return grow_table_nl<2>(c, new_capacity);
    case 3: return grow_table_nl<3>(c, new_capacity);
    default: return grow_table_nl<4>(c, new_capacity);
  }
}

// make sure `incoming` more distinct k-mers keep the load below 0.9 (syncs the stream)
static int ensure_room(kc_ctx *c, uint64_t incoming) {
  int rc = sync_ctrs(c);
  if (rc) return rc;
  if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
  uint64_t need = c->h_ctrs[CTR_ENTRIES] + incoming;
  uint64_t cap = c->capacity;
  while ((double)need > 0.9 * (double)cap) cap <<= 1;
  if (cap != c->capacity) return grow_table(c, cap);
  return KC_OK;
}

static int ensure_tile_first(kc_ctx *c, size_t ntiles) {
  if (ntiles <= c->tile_first_cap) return KC_OK;
  if (c->d_tile_first) HIPCHK(hipFree(c->d_tile_first));
  c->d_tile_first = nullptr;
  c->tile_first_cap = 0;
  HIPCHK(hipMalloc((void **)&c->d_tile_first, ntiles * 8));
  c->tile_first_cap = ntiles;
  return KC_OK;
}

// ---- library -----------------------------------------------------------------------------------
extern "C" int kc_abi_version(void) { return KC_ABI_VERSION; }

extern "C" const char *kc_error_string(int s) {
  switch (s) {
    case KC_OK: return "ok";
    case KC_ERR_INVALID_ARG: return "invalid argument";
    case KC_ERR_UNSUPPORTED_K: return "unsupported k-mer length";
    case KC_ERR_NO_DEVICE: return "no usable HIP device";
    case KC_ERR_HIP: return "HIP call failed";
    case KC_ERR_OUT_OF_MEMORY: return "out of device memory";
    case KC_ERR_CAPACITY: return "a buffer is too small (kc_last_error says which)";
    case KC_ERR_BAD_BASE: return "read holds a byte outside ACGTN";
    case KC_ERR_STATE: return "call not allowed in this state";
    default: return "unknown status";
  }
}

extern "C" const char *kc_last_error(void) { return g_last_error; }

extern "C" int kc_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int kc_num_longs(int k) { return k / 32 + 1; }

// Words of a k-mer inside the library and of a record on the shard wire: the two extension codes ride in the six spare
// low bits of the last word, and when k % 32 is 30 or 31 that word has none to spare, so those k get one word more
// (key word zero, extension codes only).  Results, dumps and lookups keep the reference's width (kc_num_longs).
extern "C" int kc_record_longs(int k) { return k / 32 + 1 + ((k % 32) >= 30 ? 1 : 0); }

static int check_k(int k) {
  if (k < 3 || k > 127) return KC_ERR_UNSUPPORTED_K;
  if (kc_record_longs(k) > KC_MAX_LONGS) return KC_ERR_UNSUPPORTED_K;  // k = 126, 127: a fifth word would be needed
  return KC_OK;
}

extern "C" int kc_owner(const uint64_t *w, int k, int rank_n) {
  if (!w || check_k(k) || rank_n < 1) return KC_ERR_INVALID_ARG;
  uint64_t a[KC_MAX_LONGS] = {0, 0, 0, 0}, h;
  for (int j = 0; j < kc_num_longs(k); j++) a[j] = w[j];  // the caller's k-mer has the reference's width
  switch (kc_record_longs(k)) {
    case 1: { uint64_t x[1] = {a[0]}; h = kc_hash<1>(x); break; }
    case 2: { uint64_t x[2] = {a[0], a[1]}; h = kc_hash<2>(x); break; }
    case 3: { uint64_t x[3] = {a[0], a[1], a[2]}; h = kc_hash<3>(x); break; }
    default: h = kc_hash<4>(a); break;
  }
  return (int)kc_owner_of_hash(h, (uint32_t)rank_n);
}

extern "C" int kc_owner_reference(const uint64_t *w, int k, int rank_n) {
  if (!w || check_k(k) || rank_n < 1) return KC_ERR_INVALID_ARG;
  switch (kc_num_longs(k)) {
    case 1: { uint64_t a[1] = {w[0]}, r[1]; kc_revcomp<1>(a, k, r); return (int)::kc_reference_owner<1>(a, r, k, (uint32_t)rank_n); }
    case 2: { uint64_t a[2] = {w[0], w[1]}, r[2]; kc_revcomp<2>(a, k, r); return (int)::kc_reference_owner<2>(a, r, k, (uint32_t)rank_n); }
    case 3: { uint64_t a[3] = {w[0], w[1], w[2]}, r[3]; kc_revcomp<3>(a, k, r); return (int)::kc_reference_owner<3>(a, r, k, (uint32_t)rank_n); }
    default: { uint64_t a[4] = {w[0], w[1], w[2], w[3]}, r[4]; kc_revcomp<4>(a, k, r); return (int)::kc_reference_owner<4>(a, r, k, (uint32_t)rank_n); }
  }
}

// keys between the library's width and the reference's (they differ when k % 32 is 30 or 31): the words beyond the
// narrower width are zero
__global__ void kc_rewidth_keys_kernel(const uint64_t *src, uint64_t *dst, uint64_t n, int nl_src, int nl_dst) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int j = 0; j < nl_dst; j++) dst[i * nl_dst + j] = j < nl_src ? src[i * nl_src + j] : 0ULL;
}

// ---- context -----------------------------------------------------------------------------------
static int create_impl(kc_ctx *c) {
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
  c->stream = c->own_stream;
  hipDeviceProp_t prop;
  HIPCHK(hipGetDeviceProperties(&prop, c->cfg.device));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  // the global table starts small: on the bucketed path it only ever holds the regions that did not fit
  c->capacity = 1ULL << 16;
  c->arena_bytes = table_bytes_for(c->capacity, c->nl);
  HIPCHK(hipMalloc((void **)&c->arena, c->arena_bytes));
  carve_table(c);
  HIPCHK(hipMalloc((void **)&c->d_ctrs, CTR_COUNT * 8));
  HIPCHK(hipHostMalloc((void **)&c->h_ctrs, CTR_COUNT * 8, hipHostMallocDefault));
  HIPCHK(hipMemsetAsync(c->d_ctrs, 0, CTR_COUNT * 8, c->stream));
  HIPCHK(hipMalloc((void **)&c->d_cb, CB_COUNT * 8));
  HIPCHK(hipHostMalloc((void **)&c->h_cb, CB_COUNT * 8, hipHostMallocDefault));
  HIPCHK(hipMemsetAsync(c->d_cb, 0, CB_COUNT * 8, c->stream));
  memset(c->h_cb, 0, CB_COUNT * 8);
  int rc = clear_table(c);
  if (rc) return rc;
  HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}

extern "C" kc_ctx *kc_create(const kc_config *cfg, int *status) {
  int st = KC_OK;
  kc_ctx *c = nullptr;
  if (!cfg || cfg->rank_n < 1 || cfg->rank_n > 64 || cfg->rank_me < 0 || cfg->rank_me >= cfg->rank_n) st = KC_ERR_INVALID_ARG;
  if (!st) st = check_k(cfg->kmer_len);
  if (!st && (cfg->device < 0 || cfg->device >= kc_device_count())) {
    st = KC_ERR_NO_DEVICE;
    snprintf(g_last_error, sizeof(g_last_error), "device %d requested, %d visible", cfg->device, kc_device_count());
  }
  if (!st) {
    c = new (std::nothrow) kc_ctx();
    if (!c) st = KC_ERR_OUT_OF_MEMORY;
  }
  if (!st) {
    c->cfg = *cfg;
    if (c->cfg.dmin_thres <= 0) c->cfg.dmin_thres = 2;
    c->k = cfg->kmer_len;
    c->nl = kc_record_longs(c->k);
    c->nl_ext = kc_num_longs(c->k);
    st = create_impl(c);
    if (st) {
      kc_destroy(c);
      c = nullptr;
    }
  }
  if (status) *status = st;
  return c;
}

static void free_index(kc_ctx *c) {
  if (c->d_index) (void)hipFree(c->d_index);
  c->d_index = nullptr;
  c->index_cap = 0;
}

static void free_ctg(kc_ctx *c) {
  if (c->ctg_table.keys) (void)hipFree(c->ctg_table.keys);
  if (c->ctg_table.vals) (void)hipFree(c->ctg_table.vals);
  if (c->d_ctg_status) (void)hipFree(c->d_ctg_status);
  memset(&c->ctg_table, 0, sizeof(c->ctg_table));
  c->d_ctg_status = nullptr;
  c->ctg_cap = c->ctg_attempted = c->ctg_new = 0;
}

static void free_results(kc_ctx *c) {
  free_index(c);
  if (c->d_out_keys) (void)hipFree(c->d_out_keys);
  if (c->d_out_keys_ext) (void)hipFree(c->d_out_keys_ext);
  c->d_out_keys_ext = nullptr;
  if (c->d_out_counts) (void)hipFree(c->d_out_counts);
  if (c->d_out_left) (void)hipFree(c->d_out_left);
  if (c->d_out_right) (void)hipFree(c->d_out_right);
  c->d_out_keys = nullptr;
  c->d_out_counts = nullptr;
  c->d_out_left = c->d_out_right = nullptr;
  c->out_cap = c->out_n = 0;
}

extern "C" void kc_destroy(kc_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  drain_kernel_times(c);
  free_results(c);
  if (c->arena) (void)hipFree(c->arena);
  if (c->d_ctrs) (void)hipFree(c->d_ctrs);
  if (c->d_w6cur) (void)hipFree(c->d_w6cur);
  if (c->d_l2snap) (void)hipFree(c->d_l2snap);
  if (c->h_ctrs) (void)hipHostFree(c->h_ctrs);
  if (c->d_tile_first) (void)hipFree(c->d_tile_first);
  if (c->d_out_plan) (void)hipFree(c->d_out_plan);
  if (c->d_stage_bases) (void)hipFree(c->d_stage_bases);
  if (c->d_stage_quals) (void)hipFree(c->d_stage_quals);
  if (c->d_stage_offsets) (void)hipFree(c->d_stage_offsets);
  if (c->d_synth) (void)hipFree(c->d_synth);
  if (c->d_sm_bytes) (void)hipFree(c->d_sm_bytes);
  if (c->d_sm_packed) (void)hipFree(c->d_sm_packed);
  if (c->d_sm_targets) (void)hipFree(c->d_sm_targets);
  if (c->d_sm_out) (void)hipFree(c->d_sm_out);
  if (c->d_sm_ctr) (void)hipFree(c->d_sm_ctr);
  host_pipe_free(c);
  free_ctg(c);
  bk_free(c, false);
  shard_free(c);
  if (c->d_cb) (void)hipFree(c->d_cb);
  if (c->h_cb) (void)hipHostFree(c->h_cb);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

extern "C" int kc_set_stream(kc_ctx *c, void *s) {
  if (!c) return KC_ERR_INVALID_ARG;
  HIPCHK(hipStreamSynchronize(c->stream));
  c->stream = s ? (hipStream_t)s : c->own_stream;
  return KC_OK;
}

// a new pass: no flat sources, the extents empty (kept, unless the geometry may change)
static void shard_reset(kc_ctx *c, bool geometry_changes) {
  c->sh.flow = c->sh.extracting = false;
  c->sh.F = 0;
  c->sh.sent = c->sh.received = 0;
  for (auto &e : c->sh_extents) e.used = 0;
  if (geometry_changes) {
    if (c->sh.d_cnt) (void)hipFree(c->sh.d_cnt);
    if (c->sh.d_at) (void)hipFree(c->sh.d_at);
    c->sh.d_cnt = nullptr;
    c->sh.d_at = nullptr;
    c->sh.nbo = 0;
  }
}

static void shard_free(kc_ctx *c) {
  shard_reset(c, true);
  for (auto &e : c->sh_extents)
    if (e.p) (void)hipFree(e.p);
  c->sh_extents.clear();
  if (c->sh.d_plan) (void)hipFree(c->sh.d_plan);
  if (c->sh.h_plan) (void)hipHostFree(c->sh.h_plan);
  c->sh.d_plan = c->sh.h_plan = nullptr;
}

extern "C" int kc_reset(kc_ctx *c, int new_k) {
  if (!c) return KC_ERR_INVALID_ARG;
  if (new_k == 0) new_k = c->k;
  int st = check_k(new_k);
  if (st) return st;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  const int old_nl = c->nl, old_k = c->k;
  if (kc_record_longs(new_k) != old_nl) free_results(c);  // else the result arrays are reused by the next finalize
  free_ctg(c);  // a contig pass belongs to one pass over the reads
  c->out_n = 0;
  c->k = new_k;
  c->cfg.kmer_len = new_k;
  c->nl = kc_record_longs(new_k);
  c->nl_ext = kc_num_longs(new_k);
  if (c->d_out_keys_ext) (void)hipFree(c->d_out_keys_ext);
  c->d_out_keys_ext = nullptr;
  // re-carve the same arena for the new word count: largest power of two that fits
  uint64_t cap = 1;
  while (table_bytes_for(cap << 1, c->nl) <= c->arena_bytes) cap <<= 1;
  c->capacity = cap;
  carve_table(c);
  HIPCHK(hipMemsetAsync(c->d_ctrs, 0, CTR_COUNT * 8, c->stream));
  st = clear_table(c);
  if (st) return st;
  c->finalized = false;
  c->num_reads = c->num_bases = 0;
  c->purged = c->sum_counts = c->unique_at_finalize = 0;
  // bucketed path: keep the arrays when the record width is unchanged, else choose the geometry again (so does a
  // new k among the one-word ones: whether the records are compact, and how, depends on k)
  if (c->bk_ready) {
    if (c->nl != old_nl || (c->nl == 1 && new_k != old_k)) {
      bk_free(c, true);  // the next geometry takes the arrays over where they are large enough
    } else {
      const size_t R = (size_t)c->gm.P1 * c->gm.P2;
      HIPCHK(hipMemsetAsync(c->bb.cnt1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
      HIPCHK(hipMemsetAsync(c->bb.used1, 0, (size_t)c->gm.G * 2 * 4, c->stream));
      HIPCHK(hipMemsetAsync(c->bb.cnt2, 0, R * 4, c->stream));
      HIPCHK(hipMemsetAsync(c->bb.flag, 0, R * 4, c->stream));
    }
  }
  c->inc_on = false;
  c->l2_per_bucket = 0;
  HIPCHK(hipMemsetAsync(c->d_cb, 0, CB_COUNT * 8, c->stream));
  c->bk_level2 = c->bk_flagged = c->table_mode = c->started = false;
  c->bk_spilled = false;
  c->l1_dropped = false;
  c->l2_held = 0;
  c->expect_base = c->expect_prev = 0;
  c->expect_host = 0;
  c->expect_host_ok = true;
  c->ovf1_ub = 0;
  shard_reset(c, c->nl != old_nl || (c->nl == 1 && new_k != old_k));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}

// ---- bucketed path: geometry, buffers, launches (kernels in kc_bucketed.hpp) ------------------------
static uint32_t count_smax(int nl) {
  switch (nl) {
    case 1: return CountLDS<1>::SMAX;
    case 2: return CountLDS<2>::SMAX;
    case 3: return CountLDS<3>::SMAX;
    default: return CountLDS<4>::SMAX;
  }
}

// the pointers of BucketBufs, in the order of kc_ctx::bk_pool
static void bk_slots(BucketBufs &b, void ***out) {
  void **ptrs[13] = {(void **)&b.rec1, (void **)&b.chain1, (void **)&b.cnt1, (void **)&b.used1, (void **)&b.rec2, (void **)&b.chain2,
                     (void **)&b.cnt2, (void **)&b.base2, (void **)&b.flag, (void **)&b.ovf1, (void **)&b.ovf2, (void **)&b.done1,
                     (void **)&b.used2};
  for (int i = 0; i < 13; i++) out[i] = ptrs[i];
}

// Give up the current geometry.  keep: its arrays stay with the context (kc_reset to another k, kc_set_tuning) and the
// next geometry takes over every one that is large enough -- BASELINE config 5's sweep k = 21, 33, 55, 77 keeps its
// arenas resident in HBM: they are allocated once per record width at most, and a width seen before allocates nothing.
static void bk_free(kc_ctx *c, bool keep = false) {
  void **slot[13];
  bk_slots(c->bb, slot);
  for (int i = 0; i < 13; i++) {
    void *p = *slot[i];
    if (p && keep && c->bk_held[i] > c->bk_pool[i].bytes) {  // the larger of the two stays
      if (c->bk_pool[i].p) (void)hipFree(c->bk_pool[i].p);
      c->bk_pool[i].p = p;
      c->bk_pool[i].bytes = c->bk_held[i];
    } else if (p) {
      (void)hipFree(p);
    }
    if (!keep && c->bk_pool[i].p) {
      (void)hipFree(c->bk_pool[i].p);
      c->bk_pool[i].p = nullptr;
      c->bk_pool[i].bytes = 0;
    }
    c->bk_held[i] = 0;
  }
  memset(&c->bb, 0, sizeof(c->bb));
  c->bk_ready = false;
  c->wire6 = false;
  c->bk_bytes = 0;
}

// `bytes` of device memory for the i-th array of the bucket buffers: what the context kept from an earlier geometry
// when that is large enough (*reused), else a fresh allocation
static int bk_take(kc_ctx *c, int i, void **out, size_t bytes, bool *reused = nullptr) {
  if (reused) *reused = false;
  if (c->bk_pool[i].p && c->bk_pool[i].bytes >= bytes) {
    *out = c->bk_pool[i].p;
    c->bk_held[i] = c->bk_pool[i].bytes;
    c->bk_pool[i].p = nullptr;
    c->bk_pool[i].bytes = 0;
    if (reused) *reused = true;
    return KC_OK;
  }
  if (c->bk_pool[i].p) {  // too small: make room for its successor
    (void)hipFree(c->bk_pool[i].p);
    c->bk_pool[i].p = nullptr;
    c->bk_pool[i].bytes = 0;
  }
  HIPCHK(hipMalloc(out, bytes));
  c->bk_held[i] = bytes;
  return KC_OK;
}

static uint32_t ilog2(uint64_t v) {
  uint32_t l = 0;
  while ((1ULL << l) < v) l++;
  return l;
}

// The same split kernel on the same virtual addresses runs at different speeds on different ALLOCATIONS of its arena:
// level 1 in 30-32 ms on one, in 36 ms or in 40 ms on another (measured,
// scripts/l1_mode_probe.py and scripts/probe_runs.sh: fixed for the life of the allocation; how the driver backed the
// memory is the suspect).  One millisecond of the kernels' write pattern alone -- every workgroup appending 64-byte runs
// round-robin to a window of 1024 open chunks that jumps through its whole part (kc_arena_probe_kernel) -- tells the
// allocations apart: 4.3 GB at >= 4.8 TB/s on a fast arena, 3.6-4.3 TB/s on a slow one.  Freeing a slow arena and asking again tends to return the
// same memory: every candidate is first written from end to end, freed and asked for again (scrub_allocation, which
// brings most slow ones back fast), and one that is still slow is HELD while another is asked for (when the device has
// the room), at most four in all; the fastest is kept.  First processes on a freshly started box draw slow arenas most often.
// KC_ARENA_PROBE=0 switches this off, =1 logs.
// The rate at which an arena takes level 1's write pattern (many interleaved append streams) depends on which physical
// memory the driver backed it with -- uniform over the allocation, not changed by re-mapping it, out of the
// application's reach (profiles/r03_arena_alloc_probe.txt, r03_arena_permute_pieces.txt): one millisecond of the pattern
// tells.  The yardstick is the arena's OWN first GiB: confined to one GiB the same pattern runs at the same rate on every
// allocation, fast or slow (4.9-5.1 TB/s over twenty allocations of ten processes, profiles/r04_arena_selection.txt),
// while spread over the whole arena it runs at 0.97-1.1 of that on most allocations, at 0.85-0.9 on some (level 1
// 27.3-27.7 instead of 26.5-27.1 ms) and at 0.7-0.75 on the slow ones (30 ms).  So an allocation is taken at once when
// its whole-arena rate reaches KC_ARENA_OK of its first-GiB rate -- a ratio, no rate in TB/s is asked of anything --
// and otherwise it is HELD while another is asked for (the same memory comes back otherwise), up to three, while the
// device has room to hold them; the best ratio wins.  The chosen arena's rate is reported (kc_arena_probe_rate;
// KC_ARENA_PROBE=1 logs every candidate), so that a caller can see a slow draw.
constexpr double KC_ARENA_OK = 0.93;
static int pick_fast_arena(kc_ctx *c, uint64_t **arena, size_t bytes, uint32_t G, const char *what) {
  const char *pe = getenv("KC_ARENA_PROBE");
  hipEvent_t e0, e1;
  HIPCHK(hipEventCreate(&e0));
  HIPCHK(hipEventCreate(&e1));
  const uint32_t rounds = 256;
  const size_t wpw = bytes / 8 / G, wpw_gib = std::min<size_t>(wpw, ((size_t)1 << 30) / 8 / G);
  const double probe_bytes = (double)G * rounds * 65536.0;
  constexpr int NCAND = 3;
  uint64_t *cand[NCAND] = {*arena, nullptr, nullptr};
  double rate[NCAND] = {0, 0, 0}, ratio[NCAND] = {0, 0, 0};
  int ncand = 1;
  auto probe = [&](uint64_t *p, size_t words_per_wg) -> double {  // TB/s of the write pattern on p, the better of two runs; < 0: a HIP call failed
    float best = 1e30f;
    for (int rep = 0; rep < 2; rep++) {
      float ms = 0;
      if (hipEventRecord(e0, c->stream) != hipSuccess) return -1.0;
      hipLaunchKernelGGL(kc_arena_probe_kernel, dim3(G), dim3(WGB), 0, c->stream, p, words_per_wg, rounds);
      if (hipEventRecord(e1, c->stream) != hipSuccess || hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms, e0, e1) != hipSuccess)
        return -1.0;
      best = std::min(best, ms);
    }
    return probe_bytes / (best * 1e-3) / 1e12;
  };
  const bool log = pe || getenv("KC_DEBUG_ADDR");
  int rc = KC_OK;
  for (int i = 0; i < NCAND && !rc; i++) {
    rate[i] = probe(cand[i], wpw);
    const double gib = probe(cand[i], wpw_gib);
    if (rate[i] < 0 || gib <= 0) rc = KC_ERR_HIP;
    else ratio[i] = rate[i] / gib;
    if (log) fprintf(stderr, "kc arena probe (%s): allocation %d, %.2f TB/s, %.2f of its first GiB alone, %p\n", what, i, rate[i], ratio[i], (void *)cand[i]);
    if (rc || ratio[i] >= KC_ARENA_OK || i == NCAND - 1) break;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)8 << 30)) break;  // no room to hold another
    if (hipMalloc((void **)&cand[i + 1], bytes) != hipSuccess) {
      (void)hipGetLastError();
      cand[i + 1] = nullptr;
      break;
    }
    ncand = i + 2;
  }
  int best_i = -1;
  for (int i = 0; i < ncand; i++)
    if (cand[i] && (best_i < 0 || ratio[i] > ratio[best_i])) best_i = i;
  for (int i = 0; i < ncand; i++)
    if (i != best_i && cand[i]) (void)hipFree(cand[i]);
  *arena = best_i >= 0 ? cand[best_i] : nullptr;
  c->arena_probe_tbps = best_i >= 0 ? rate[best_i] : 0.0;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (!rc && !*arena) rc = KC_ERR_OUT_OF_MEMORY;
  if (rc) snprintf(g_last_error, sizeof(g_last_error), "choosing the %s arena failed", what);
  return rc;
}

constexpr double KC_MAX_REGION_LOAD = 0.72;  // highest mean load of the region tables bk_init plans with

// Choose the geometry from the configured sizes and allocate the level-1 / level-2 arenas.
static int bk_init(kc_ctx *c) {
  if (c->bk_ready) return KC_OK;
  const kc_tuning &t = c->tuning;
  Geom &g = c->gm;
  memset(&g, 0, sizeof(g));
  const uint64_t bcap = c->cfg.max_kmers_buffered ? c->cfg.max_kmers_buffered : (1ULL << 26);
  double est = c->cfg.max_elems ? (double)c->cfg.max_elems : 0.35 * (double)bcap;  // 1/depth(4) + error share, kmer_dht.cpp:126-131
  // single-pass shard flow: this shard's k-mers all fall into the 1/rank_n of the buckets it owns, so the geometry as a
  // whole is sized as if it held rank_n times as many
  const double flow_n = ((c->cfg.flags & KC_FLAG_SHARD_BUCKETS) && c->cfg.rank_n > 1) ? (double)c->cfg.rank_n : 1.0;
  est *= flow_n;
  const uint32_t smax = count_smax(c->nl);
  // Region tables are meant to run at about 0.4 load: the lanes of a wave probe in lock step, so a wave pays for its
  // longest probe, and the count kernel's time grows steeply with the load (measured: 1.6x from 0.4 to 0.6), more
  // than the per-region costs (barriers, the exposed latency of the first loads) shrink with fewer, fuller regions.
  // Half-size tables let two region workgroups share a CU (one's barriers and scans overlap the other's inserts:
  // measured 10 % faster counting; k=51 at 0.68 load: 28 ms against 36 ms for the full-size tables at 0.34), so prefer
  // them while 2^20 regions of them stay under KC_MAX_REGION_LOAD (a region's distinct k-mers scatter by a few
  // per cent around the mean: 0.72 leaves 15 standard deviations of room at these sizes).
  g.S = t.slots ? std::min(std::max(t.slots, 16u), smax) : smax;
  while (g.S & (g.S - 1)) g.S &= g.S - 1;  // power of two (round down)
  const double target_load = 0.4;
  if (!t.slots && est <= KC_MAX_REGION_LOAD * (double)(1u << 20) * (g.S / 2)) g.S /= 2;
  const uint64_t regions_needed = std::min<uint64_t>((uint64_t)(est / (target_load * g.S)) + 1, 1ULL << 20);
  // the two fan-outs multiply to the number of regions; any value up to 1024 each (the hash fields are mapped by
  // multiply-shift), level 1 the smaller one because it holds fewer records per round
  g.P1 = t.p1 ? t.p1 : (uint32_t)std::max<double>(1.0, floor(sqrt((double)regions_needed)));
  g.P2 = t.p2 ? t.p2 : (uint32_t)((regions_needed + g.P1 - 1) / g.P1);
  // Compact records (Geom::cp): one-word k-mers with k <= 23 whose regions imply at least 2k - 26 bits of the mixed
  // k-mer.  The fan-outs must be powers of two, so an automatic choice is rounded up to the next such pair; fan-outs
  // set by the caller are used as they are and decide by themselves.  tuning.mode 2 keeps the wide records.
  if (c->nl == 1 && c->k <= KC_COMPACT_MAX_K && t.mode != 2) {
    uint32_t la = 0, lb = 0;
    if (t.p1 || t.p2) {
      la = ilog2(g.P1);
      lb = ilog2(g.P2);
    } else {
      const uint32_t lr = std::min<uint32_t>(20, ilog2(regions_needed));  // 2^lr >= regions_needed
      la = lr / 2;
      lb = lr - la;
      if (2u * (uint32_t)c->k + 6u <= 32u + lr && la >= 1) {
        g.P1 = 1u << la;
        g.P2 = 1u << lb;
      }
    }
    if (g.P1 == (1u << la) && g.P2 == (1u << lb) && la >= 1 && lb >= 1 && 2u * (uint32_t)c->k + 6u <= 32u + la + lb &&
        2u * (uint32_t)c->k > la + lb) {
      g.cp = 1;
      g.la = la;
      g.lb = lb;
      g.k2 = 2u * (uint32_t)c->k;
    }
  }
  if (g.P1 < 1 || g.P2 < 1 || g.P1 > PMAX || g.P2 > PMAX) return KC_ERR_INVALID_ARG;
  // six-byte level-1 records: wherever the kernels that write them run -- compact records whose mix fits 32 bits below the
  // level-1 bucket (k <= 21 with 1024 buckets; k = 21, MHM2's first and only one-word k of its default sweep,
  // src/options.hpp:80, has instantiations of its own)
  // (KC_L1_ROUND16=0: the general kernels with their rounds of eight and 8-byte records, for A/B runs)
  {
    static const bool round16 = !(getenv("KC_L1_ROUND16") && getenv("KC_L1_ROUND16")[0] == '0');
    g.rec6 = (c->nl == 1 && g.cp && g.k2 - g.la <= 32 && round16) ? 1u : 0u;
  }
  // the records flow's wire: units of four six-byte records where level 1 writes those (kc_wire6.hpp), k-mer records otherwise
  // (the owner's eight bits of the mix, 13..20, must lie below the region's: at least 21 bits of the mix that no region implies)
  c->wire6 = (c->cfg.flags & KC_FLAG_WIRE_UNITS) && g.rec6 && g.k2 - g.la - g.lb >= 21 && !(c->cfg.flags & KC_FLAG_REFERENCE_OWNER) &&
             c->cfg.rank_n >= 1 && c->cfg.rank_n <= (int)WIRE6_MAX_SHARDS;
  // one writer per CU, but never so many that a writer's share of the buffer is below a few rounds of records
  g.G = t.writers ? std::min<uint32_t>(t.writers, GMAX)
                  : (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)std::min<int>(c->num_cus, GMAX), bcap / (4 * 16384)));
  if (g.G < 1) g.G = 1;
  const uint64_t R = (uint64_t)g.P1 * g.P2;
  // chunk sizes: at most ~1/8 of a destination's mean share, within [16, 512] / [16, 1024] records
  const double mean1 = (double)bcap / ((double)g.G * g.P1), mean2 = flow_n * (double)bcap / (double)R;
  // (level-1 chunks of up to 4096 records, an eighth of a chain's mean length at most: with 512 a bucket of every wave
  // needed a new chunk in every round, and handing those out -- an LDS counter bump the compiler serialises over the
  // lanes, a chain entry to memory -- was a millisecond of level 1: 31.3 -> 30.0 ms, profiles/r03_tune_chunk_sizes.txt;
  // level-2 chunks larger than 1024 make level 2 slower)
  g.log2CH1 = t.chunk1 ? ilog2(t.chunk1) : std::min<uint32_t>(12, std::max<uint32_t>(4, ilog2((uint64_t)(mean1 / 8) + 1)));
  g.log2CH2 = t.chunk2 ? ilog2(t.chunk2) : std::min<uint32_t>(10, std::max<uint32_t>(4, ilog2((uint64_t)(mean2 / 8) + 1)));
  const uint64_t CH1 = 1ULL << g.log2CH1, CH2 = 1ULL << g.log2CH2;
  // chains may grow to several times the mean: k-mer multiplicities are heavy-tailed
  g.L1MAX = t.chain1_max ? t.chain1_max : (uint32_t)(4.0 * mean1 / (double)CH1) + 8;
  g.L2MAX = std::min<uint32_t>(CHAIN_LDS, t.chain2_max ? t.chain2_max : (uint32_t)(4.0 * mean2 / (double)CH2) + 8);
  // compact records: a region's chain has room for as many records as the count kernel takes at all (65535), so that a
  // buffer that is smaller than the input can hand its records on to level 2 and start again (bk_light_spill): the
  // regions then hold the whole input's records, not a buffer's
  if (g.cp && !t.chain2_max) g.L2MAX = std::min<uint32_t>(CHAIN_LDS, std::max<uint32_t>(g.L2MAX, (uint32_t)((KC_COUNT_MAX + CH2) / CH2) + 1));
  g.A1 = t.arena1 ? t.arena1 : (uint32_t)(1.03 * (double)bcap / ((double)g.G * (double)CH1)) + g.P1 + 16;
  const uint64_t a2 = bcap / CH2 + R + g.P1 + 16;
  if (a2 >= (1ULL << 32) || (uint64_t)g.A1 * g.G >= (1ULL << 32) || bcap / g.P1 >= (1ULL << 31)) return KC_ERR_INVALID_ARG;
  g.A2 = (uint32_t)a2;
  BucketBufs &b = c->bb;
  memset(&b, 0, sizeof(b));
  // Overflow lists.  Neither can lose a record: every level-1 launch is bounded by the free room of the first
  // (bk_ovf1_room) -- a quarter (one-word records) or an eighth of the buffer, so that a block of reads goes through in
  // a few launches -- and level 2 is run again with a second list of the size it asked for when that one was too small
  // (bk_build_regions): the second starts at an eighth of the first (1.7 GB instead of 13 at 50 M reads).
  // (at least one super-tile of positions, whatever the buffer: a level-1 launch covers whole tiles)
  b.ovf1_cap = t.ovf_capacity ? t.ovf_capacity : std::max<uint64_t>(bcap / (c->nl == 1 ? 4 : 8) + 4096, 2 * (uint64_t)SUPER_SPAN);
  b.ovf2_cap = t.ovf_capacity ? t.ovf_capacity : std::max<uint64_t>(bcap / (c->nl == 1 ? 32 : 64) + 4096, 2 * (uint64_t)SUPER_SPAN);
  const size_t w = (size_t)c->nl * 8;
  const size_t nseg = (size_t)g.G * g.P1;
  // (+ 64: a pair of six-byte records is loaded and stored as twelve bytes, wherever in the arena it lies)
  const size_t rec1_bytes = (size_t)g.G * g.A1 * CH1 * (g.rec6 ? 6 : w) + 64, rec2_bytes = (size_t)g.A2 * CH2 * (g.cp ? 4 : w);
  bool rec1_reused = false, rec2_reused = false;
  {
    int rc = bk_take(c, 0, (void **)&b.rec1, rec1_bytes, &rec1_reused);
    // The level-1 arena is chosen before anything else is allocated (pick_fast_arena; for every arena of a GiB or more:
    // below that a stage is over before the difference shows; an arena taken over from an earlier geometry has been
    // chosen already).  (Chosen last, a second candidate that won lay behind the level-2 arena in allocation order, and
    // level 2 then took 28 instead of 26 ms: profiles/r04_arena_selection.txt.)
    const char *pe = getenv("KC_ARENA_PROBE");
    if (!rc && !(pe && pe[0] == '0') && rec1_bytes >= ((size_t)1 << 30) && !rec1_reused) {
      rc = pick_fast_arena(c, &b.rec1, rec1_bytes, g.G, "level 1");
    }
    if (!rc) rc = bk_take(c, 1, (void **)&b.chain1, nseg * g.L1MAX * 4);
    if (!rc) rc = bk_take(c, 2, (void **)&b.cnt1, nseg * 4);
    if (!rc) rc = bk_take(c, 3, (void **)&b.used1, (size_t)g.G * 2 * 4);
    if (!rc) rc = bk_take(c, 4, (void **)&b.rec2, rec2_bytes, &rec2_reused);
    // (level 2's arena chosen the same way -- its ratio does not predict level 2's time, and what level 1 gains on an
    // arena with a high ratio level 2, which reads that arena, loses again: profiles/r04_arena_selection.txt, section E)
    if (!rc) rc = bk_take(c, 5, (void **)&b.chain2, (size_t)R * g.L2MAX * 4);
    if (!rc) rc = bk_take(c, 6, (void **)&b.cnt2, (size_t)R * 4);
    if (!rc) rc = bk_take(c, 7, (void **)&b.base2, ((size_t)g.P1 + 1) * 4);
    if (!rc) rc = bk_take(c, 8, (void **)&b.flag, (size_t)R * 4);
    if (!rc) rc = bk_take(c, 9, (void **)&b.ovf1, b.ovf1_cap * w);
    if (!rc) rc = bk_take(c, 10, (void **)&b.ovf2, b.ovf2_cap * w);
    if (!rc) rc = bk_take(c, 11, (void **)&b.done1, nseg * 4);
    if (!rc) rc = bk_take(c, 12, (void **)&b.used2, (size_t)g.P1 * 4);
    if (rc) return rc;
  }
  c->inc_on = false;
  c->l2_per_bucket = 0;
  HIPCHK(hipMemsetAsync(b.cnt1, 0, nseg * 4, c->stream));
  HIPCHK(hipMemsetAsync(b.used1, 0, (size_t)g.G * 2 * 4, c->stream));
  HIPCHK(hipMemsetAsync(b.cnt2, 0, (size_t)R * 4, c->stream));
  HIPCHK(hipMemsetAsync(b.flag, 0, (size_t)R * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->d_cb, 0, CB_COUNT * 8, c->stream));
  if (getenv("KC_DEBUG_ADDR"))
    fprintf(stderr, "kc arenas: rec1 %p (%zu MB) chain1 %p cnt1 %p rec2 %p chain2 %p ovf1 %p ovf2 %p\n", (void *)b.rec1, rec1_bytes >> 20,
            (void *)b.chain1, (void *)b.cnt1, (void *)b.rec2, (void *)b.chain2, (void *)b.ovf1, (void *)b.ovf2);
  c->bk_bytes = rec1_bytes + rec2_bytes + (b.ovf1_cap + b.ovf2_cap) * w + nseg * (g.L1MAX + 1) * 4 + (size_t)R * (g.L2MAX + 2) * 4;
  c->bk_capacity = bcap;
  c->bk_ready = true;
  c->bk_level2 = c->bk_flagged = false;
  c->bk_rot = 0;
  return KC_OK;
}

static int sync_cb(kc_ctx *c) {
  HIPCHK(hipMemcpyAsync(c->h_cb, c->d_cb, CB_COUNT * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}

static int ensure_room(kc_ctx *c, uint64_t incoming);

template <int NL>
static void launch_ovf1_drain(kc_ctx *c, uint64_t n) {
  auto kern = (NL == 1 && c->gm.cp) ? kc_ovf1_drain_kernel<NL, NL == 1> : kc_ovf1_drain_kernel<NL, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 256 * 32)), dim3(TPB), 0, c->stream, c->gm, c->bb, n,
                     c->table, c->d_ctrs);
}

// Records the level-1 overflow list can still take.  A launch over `want` k-mer positions can overflow at most that
// many records: when the list has less room than that and holds something, its records are moved to the global table
// first (syncs the stream).  The caller bounds its launch by the returned room.
static int bk_ovf1_room(kc_ctx *c, uint64_t want, uint64_t *room) {
  // no launch since the list was last read can have filled it: no need to ask (lost records, if any, are noticed when
  // the regions are built)
  if (c->ovf1_ub + want <= c->bb.ovf1_cap) {
    *room = c->bb.ovf1_cap - c->ovf1_ub;
    return KC_OK;
  }
  int rc = sync_cb(c);
  if (rc) return rc;
  if (c->h_cb[CB_FATAL] & ~(c->inc_on ? (uint64_t)FATAL_OVF2 : 0ULL)) {  // (an instalment's full second list is bk_build_regions' business)
    snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost (fatal bits %llu)", (unsigned long long)c->h_cb[CB_FATAL]);
    return KC_ERR_CAPACITY;
  }
  uint64_t used = std::min<uint64_t>(c->h_cb[CB_OVF1], c->bb.ovf1_cap);
  // (inside a kc_shard_extract the list also holds records of other shards, which must not reach this shard's table: the
  // launch is bounded by the room that is left, and the list is emptied when the block is packed)
  if (used && c->bb.ovf1_cap - used < want && !c->sh.extracting) {
    rc = ensure_room(c, used);
    if (rc) return rc;
    {
      KernelTimer kt(c, KT_FALLBACK);
      switch (c->nl) {
        case 1: launch_ovf1_drain<1>(c, used); break;
        case 2: launch_ovf1_drain<2>(c, used); break;
        case 3: launch_ovf1_drain<3>(c, used); break;
        default: launch_ovf1_drain<4>(c, used); break;
      }
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->d_cb + CB_OVF1, 0, 8, c->stream));
    used = 0;
  }
  c->ovf1_ub = used;
  *room = c->bb.ovf1_cap - used;
  return KC_OK;
}

template <typename K>
static int set_dyn_lds(K kernel, size_t bytes) {
  HIPCHK(hipFuncSetAttribute((const void *)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  return KC_OK;
}

// dynamic LDS of the split kernels: the working set, the sorted staging and (level 1 from reads, wide records) one
// uint16 bucket id per staged record; the sender-side binning stages nothing
template <int NL> static size_t lds_l1_reads() {
  return ((sizeof(L1LDS) + 15) & ~size_t(15)) + Rnd<NL>::STAGE_READS + ((size_t)WGB * Rnd<NL>::RPOS_READS + 64) * 2;
}
template <int NL> static size_t lds_bin_reads() { return (sizeof(L1LDS) + 15) & ~size_t(15); }
template <int NL> static size_t lds_l1_records() { return ((sizeof(L1RLDS) + 15) & ~size_t(15)) + Rnd<NL>::STAGE; }
template <int NL> static size_t lds_l2() { return ((sizeof(L2LDS) + 15) & ~size_t(15)) + Rnd<NL>::STAGE; }

// compact records are a property of the geometry (bk_init) and only exist for one-word k-mers
template <int NL> static bool use_cp(const kc_ctx *c) { return NL == 1 && c->gm.cp != 0; }

template <int NL, int FMT>
static int launch_l1_reads_t(kc_ctx *c, const ExtractArgs &a, uint64_t nsuper) {
  const bool sh = c->cfg.rank_n > 1 && !c->sh.extracting;  // the shard flow ships whole buckets instead of testing k-mers
  // shard flow: the chains of the buckets other shards own are emptied after every block and live at the top of the
  // writers' arenas (ChainDest::own_lo); everywhere else every bucket is the context's own
  c->gm.own_lo = 0;
  c->gm.own_hi = PMAX;
  if (c->sh.extracting && c->cfg.rank_n > 1) {
    c->gm.own_lo = shard_first_bucket((uint32_t)c->cfg.rank_me, c->gm.P1, (uint32_t)c->cfg.rank_n);
    c->gm.own_hi = shard_first_bucket((uint32_t)c->cfg.rank_me + 1, c->gm.P1, (uint32_t)c->cfg.rank_n);
  }
  // compact records at k = 21 (MHM2's first and only one-word k of its default sweep, src/options.hpp:80): the
  // instantiation made for that k; any other k takes the general one
  constexpr int K21 = NL == 1 ? 21 : 0;
  const bool k21 = NL == 1 && c->k == 21 && c->gm.k2 - c->gm.la <= 32;  // its registers hold the 32 bits below the bucket
  auto kern = use_cp<NL>(c) ? (k21 ? (sh ? kc_l1_reads_kernel<NL, FMT, NL == 1, true, K21> : kc_l1_reads_kernel<NL, FMT, NL == 1, false, K21>)
                                   : (sh ? kc_l1_reads_kernel<NL, FMT, NL == 1, true, 0> : kc_l1_reads_kernel<NL, FMT, NL == 1, false, 0>))
                            : (sh ? kc_l1_reads_kernel<NL, FMT, false, true, 0> : kc_l1_reads_kernel<NL, FMT, false, false, 0>);
  const unsigned grid = (unsigned)std::min<uint64_t>(c->gm.G, nsuper);
#ifdef KC_ABLATE
  c->gm.abl = getenv("KC_ABL_L1") ? (uint32_t)atoi(getenv("KC_ABL_L1")) : 0u;
#endif
  if constexpr (NL == 1) {
    // ... and its rounds of sixteen k-mers per thread, six-byte records (kc_l1_reads16_kernel; Geom::rec6)
    if (c->gm.rec6) {
      // (k = 21 has an instantiation of its own, with its shift counts and masks as constants; any other k of the short form
      // runs the same kernel with them in registers)
      auto kern16 = c->k == 21 ? (sh ? kc_l1_reads16_kernel<FMT, true, 21> : kc_l1_reads16_kernel<FMT, false, 21>)
                               : (sh ? kc_l1_reads16_kernel<FMT, true, 0> : kc_l1_reads16_kernel<FMT, false, 0>);
      int rc16 = set_dyn_lds(kern16, l1x16_lds_bytes());
      if (rc16) return rc16;
      KernelTimer kt(c, FMT == FMT_READS_UQ ? KT_L1_READS_UQ : KT_L1_READS16);
      hipLaunchKernelGGL(kern16, dim3(grid), dim3(WGB), l1x16_lds_bytes(), c->stream, a, c->gm, c->bb, nsuper, c->bk_rot, c->d_ctrs, c->d_cb);
      c->bk_rot = (uint32_t)((c->bk_rot + nsuper) % c->gm.G);
      return KC_OK;
    }
  }
  int rc = set_dyn_lds(kern, lds_l1_reads<NL>());
  if (rc) return rc;
  KernelTimer kt(c, FMT == FMT_READS_UQ ? KT_L1_READS_UQ : KT_L1_READS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WGB), lds_l1_reads<NL>(), c->stream, a, c->gm, c->bb, nsuper, c->bk_rot, c->d_ctrs, c->d_cb);
  c->bk_rot = (uint32_t)((c->bk_rot + nsuper) % c->gm.G);
  return KC_OK;
}

static int launch_l1_reads(kc_ctx *c, const ExtractArgs &a, uint64_t ntiles, int fmt) {
  if (fmt == FMT_READS) {
    switch (c->nl) {
      case 1: return launch_l1_reads_t<1, FMT_READS>(c, a, ntiles);
      case 2: return launch_l1_reads_t<2, FMT_READS>(c, a, ntiles);
      case 3: return launch_l1_reads_t<3, FMT_READS>(c, a, ntiles);
      default: return launch_l1_reads_t<4, FMT_READS>(c, a, ntiles);
    }
  }
  if (fmt == FMT_READS_UQ) {
    switch (c->nl) {
      case 1: return launch_l1_reads_t<1, FMT_READS_UQ>(c, a, ntiles);
      case 2: return launch_l1_reads_t<2, FMT_READS_UQ>(c, a, ntiles);
      case 3: return launch_l1_reads_t<3, FMT_READS_UQ>(c, a, ntiles);
      default: return launch_l1_reads_t<4, FMT_READS_UQ>(c, a, ntiles);
    }
  }
  if (fmt == FMT_PACKED) {
    switch (c->nl) {
      case 1: return launch_l1_reads_t<1, FMT_PACKED>(c, a, ntiles);
      case 2: return launch_l1_reads_t<2, FMT_PACKED>(c, a, ntiles);
      case 3: return launch_l1_reads_t<3, FMT_PACKED>(c, a, ntiles);
      default: return launch_l1_reads_t<4, FMT_PACKED>(c, a, ntiles);
    }
  }
  switch (c->nl) {
    case 1: return launch_l1_reads_t<1, FMT_SEQBLOCK>(c, a, ntiles);
    case 2: return launch_l1_reads_t<2, FMT_SEQBLOCK>(c, a, ntiles);
    case 3: return launch_l1_reads_t<3, FMT_SEQBLOCK>(c, a, ntiles);
    default: return launch_l1_reads_t<4, FMT_SEQBLOCK>(c, a, ntiles);
  }
}

template <int NL, int FMT>
static int launch_bin_reads_t(kc_ctx *c, const ExtractArgs &a, uint64_t nsuper) {
  auto kern = kc_bin_reads_kernel<NL, FMT>;
  int rc = set_dyn_lds(kern, lds_bin_reads<NL>());
  if (rc) return rc;
  const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->num_cus, nsuper);
  KernelTimer kt(c, KT_EXTRACT_BIN);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WGB), lds_bin_reads<NL>(), c->stream, a, nsuper, c->d_ctrs);
  return KC_OK;
}

// pieces per destination of the wire units (kc_wire6.hpp)
static uint32_t bin_lg_pieces(const kc_ctx *c) {
  if (!c->wire6) return 0;
  uint32_t lg = wire6_lg_pieces((uint32_t)c->cfg.rank_n);
  if (const char *e = getenv("KC_WIRE6_LG_PIECES")) lg = std::min<uint32_t>(lg, (uint32_t)atoi(e));  // (A/B runs)
  return lg;
}
static uint32_t bin_pieces(const kc_ctx *c) { return 1u << bin_lg_pieces(c); }

template <int FMT>
static int launch_bin16_t(kc_ctx *c, const ExtractArgs &a, uint64_t nsuper) {
  auto kern = c->k == 21 ? kc_bin16_kernel<FMT, 21> : kc_bin16_kernel<FMT, 0>;
  int rc = set_dyn_lds(kern, bin16_lds_bytes());
  if (rc) return rc;
  const unsigned grid = (unsigned)std::min<uint64_t>((uint64_t)c->num_cus, nsuper);
  KernelTimer kt(c, KT_BIN16);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WGB), bin16_lds_bytes(), c->stream, a, c->gm, nsuper, c->d_ctrs, c->d_w6cur, bin_lg_pieces(c));
  return KC_OK;
}

static int launch_bin_reads(kc_ctx *c, const ExtractArgs &a, uint64_t ntiles, int fmt) {
  if (c->wire6) {  // six-byte wire records (kc_wire6.hpp)
    if (fmt == FMT_READS) return launch_bin16_t<FMT_READS>(c, a, ntiles);
    if (fmt == FMT_READS_UQ) return launch_bin16_t<FMT_READS_UQ>(c, a, ntiles);
    if (fmt == FMT_PACKED) return launch_bin16_t<FMT_PACKED>(c, a, ntiles);
    return launch_bin16_t<FMT_SEQBLOCK>(c, a, ntiles);
  }
  if (fmt == FMT_READS) {
    switch (c->nl) {
      case 1: return launch_bin_reads_t<1, FMT_READS>(c, a, ntiles);
      case 2: return launch_bin_reads_t<2, FMT_READS>(c, a, ntiles);
      case 3: return launch_bin_reads_t<3, FMT_READS>(c, a, ntiles);
      default: return launch_bin_reads_t<4, FMT_READS>(c, a, ntiles);
    }
  }
  if (fmt == FMT_READS_UQ) {
    switch (c->nl) {
      case 1: return launch_bin_reads_t<1, FMT_READS_UQ>(c, a, ntiles);
      case 2: return launch_bin_reads_t<2, FMT_READS_UQ>(c, a, ntiles);
      case 3: return launch_bin_reads_t<3, FMT_READS_UQ>(c, a, ntiles);
      default: return launch_bin_reads_t<4, FMT_READS_UQ>(c, a, ntiles);
    }
  }
  if (fmt == FMT_PACKED) {
    switch (c->nl) {
      case 1: return launch_bin_reads_t<1, FMT_PACKED>(c, a, ntiles);
      case 2: return launch_bin_reads_t<2, FMT_PACKED>(c, a, ntiles);
      case 3: return launch_bin_reads_t<3, FMT_PACKED>(c, a, ntiles);
      default: return launch_bin_reads_t<4, FMT_PACKED>(c, a, ntiles);
    }
  }
  switch (c->nl) {
    case 1: return launch_bin_reads_t<1, FMT_SEQBLOCK>(c, a, ntiles);
    case 2: return launch_bin_reads_t<2, FMT_SEQBLOCK>(c, a, ntiles);
    case 3: return launch_bin_reads_t<3, FMT_SEQBLOCK>(c, a, ntiles);
    default: return launch_bin_reads_t<4, FMT_SEQBLOCK>(c, a, ntiles);
  }
}

template <int NL>
static int launch_l1_records_t(kc_ctx *c, const uint64_t *recs, uint64_t n) {
  auto kern = use_cp<NL>(c) ? kc_l1_records_kernel<NL, NL == 1> : kc_l1_records_kernel<NL, false>;
  c->gm.own_lo = 0;
  c->gm.own_hi = PMAX;
  const uint64_t per_round = (uint64_t)WGB * Rnd<NL>::RPOS;
  const unsigned grid = (unsigned)std::min<uint64_t>(c->gm.G, (n + per_round - 1) / per_round);
  if constexpr (NL == 1) {
    static_assert(Rnd<1>::RPOS == R16, "both kernels take sixteen records per thread and round");
    if (c->gm.rec6) {  // short form: 6-byte staging, pair stores, six-byte records (kc_l1_records16_kernel)
      int rc16 = set_dyn_lds(kc_l1_records16_kernel, l1r16_lds_bytes());
      if (rc16) return rc16;
      KernelTimer kt(c, KT_L1_RECORDS);
      hipLaunchKernelGGL(kc_l1_records16_kernel, dim3(grid), dim3(WGB), l1r16_lds_bytes(), c->stream, recs, n, c->gm, c->bb, c->bk_rot, c->d_ctrs, c->d_cb);
      c->bk_rot = (uint32_t)((c->bk_rot + (n + per_round - 1) / per_round) % c->gm.G);
      return KC_OK;
    }
  }
  int rc = set_dyn_lds(kern, lds_l1_records<NL>());
  if (rc) return rc;
  KernelTimer kt(c, KT_L1_RECORDS);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(WGB), lds_l1_records<NL>(), c->stream, recs, n, c->gm, c->bb, c->bk_rot, c->d_ctrs, c->d_cb);
  c->bk_rot = (uint32_t)((c->bk_rot + (n + per_round - 1) / per_round) % c->gm.G);
  return KC_OK;
}

// level 1 from wire units (kc_wire6.hpp): n pieces (<= WIRE6_SRC_PIECES, none empty) of slots[j] six-byte slots at base + j * stride
static int launch_l1_wire6(kc_ctx *c, const uint8_t *base, uint64_t stride, uint32_t n, const uint64_t *slots) {
  c->gm.own_lo = 0;
  c->gm.own_hi = PMAX;
  constexpr uint64_t PPR = (uint64_t)WGB * R16 / 2;  // pairs per round
  Wire6Src src;
  memset(&src, 0, sizeof(src));
  src.base = base;
  src.stride = stride;
  src.n = n;
  uint64_t rounds = 0;
  for (uint32_t j = 0; j < n; j++) {
    src.rpre[j] = (uint32_t)rounds;
    src.pairs[j] = (uint32_t)(slots[j] / 2);
    rounds += (slots[j] / 2 + PPR - 1) / PPR;
  }
  src.rpre[n] = (uint32_t)rounds;
  const unsigned grid = (unsigned)std::min<uint64_t>(c->gm.G, rounds);
  int rc = set_dyn_lds(kc_l1_wire6_kernel, l1r16_lds_bytes());
  if (rc) return rc;
  KernelTimer kt(c, KT_L1_WIRE6);
  hipLaunchKernelGGL(kc_l1_wire6_kernel, dim3(grid), dim3(WGB), l1r16_lds_bytes(), c->stream, src, c->gm, c->bb, c->bk_rot, c->d_ctrs, c->d_cb);
  c->bk_rot = (uint32_t)((c->bk_rot + rounds) % c->gm.G);
  return KC_OK;
}

template <int NL>
static int bk_drain_t(kc_ctx *c) {
  {
    KernelTimer kt(c, KT_FALLBACK);
    auto kern = use_cp<NL>(c) ? kc_l1_to_table_kernel<NL, NL == 1> : kc_l1_to_table_kernel<NL, false>;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<size_t>((size_t)c->gm.G * c->gm.P1, 65536)), dim3(TPB), 0, c->stream,
                       c->gm, c->bb, c->table, c->d_ctrs);
  }
  const uint64_t n1 = std::min<uint64_t>(c->h_cb[CB_OVF1], c->bb.ovf1_cap);
  if (n1) {
    KernelTimer kt(c, KT_INSERT_RECORDS);
    unsigned nblk = (unsigned)std::min<uint64_t>((n1 + TPB - 1) / TPB, 256 * 32);
    hipLaunchKernelGGL(kc_insert_records_kernel<NL>, dim3(nblk), dim3(TPB), 0, c->stream, c->bb.ovf1, n1, c->table, c->d_ctrs, 0u);
  }
  return KC_OK;
}

// The buffer is full: move everything buffered into the global table and stay on the table path.
static int bk_spill_pass(kc_ctx *c);
static int bk_drain_to_table(kc_ctx *c) {
  int rc;
  if (c->l1_dropped) {  // level 2 holds records that level 1 no longer has: count them and merge them into the table first
    rc = bk_spill_pass(c);
    if (rc) return rc;
  }
  rc = sync_ctrs(c);
  if (rc) return rc;
  rc = sync_cb(c);
  if (rc) return rc;
  if (c->h_cb[CB_FATAL] & ~(c->inc_on ? (uint64_t)FATAL_OVF2 : 0ULL)) {  // (what instalments of level 2 made is dropped here anyway)
    snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer overflow lists exhausted: raise max_kmers_buffered");
    return KC_ERR_CAPACITY;
  }
  rc = ensure_room(c, c->h_ctrs[CTR_INSERTED]);
  if (rc) return rc;
  switch (c->nl) {
    case 1: rc = bk_drain_t<1>(c); break;
    case 2: rc = bk_drain_t<2>(c); break;
    case 3: rc = bk_drain_t<3>(c); break;
    default: rc = bk_drain_t<4>(c); break;
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemsetAsync(c->bb.cnt1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.used1, 0, (size_t)c->gm.G * 2 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->d_cb, 0, CB_COUNT * 8, c->stream));
  c->inc_on = false;  // (what instalments of level 2 had taken is still in the level-1 chains, which went to the table whole)
  c->l2_per_bucket = 0;
  c->table_mode = true;
  return KC_OK;
}

static bool bk_active(const kc_ctx *c) { return c->tuning.mode != 1 && !c->table_mode; }
static int bk_spill_pass(kc_ctx *c);
static int bk_light_spill(kc_ctx *c, uint64_t buffered);
static int bk_level2_instalment(kc_ctx *c);
// A shard of several that has started the shard flow owns level-1 buckets, not hash values: the entry points that test
// ownership per k-mer (kc_submit_*, kc_insert_records) would put records where its level 2 never looks.
static bool shard_flow_only(const kc_ctx *c) { return c->sh.flow && c->cfg.rank_n > 1; }

// ---- extraction launches -----------------------------------------------------------------------
template <int NL, int FMT>
static void launch_extract_t(kc_ctx *c, const ExtractArgs &a, unsigned ntiles) {
  hipLaunchKernelGGL((kc_extract_kernel<NL, FMT>), dim3(ntiles), dim3(TPB), 0, c->stream, a, c->table, c->d_ctrs);
}

template <int FMT>
static void launch_extract_m(kc_ctx *c, const ExtractArgs &a, unsigned ntiles) {
  switch (c->nl) {
    case 1: launch_extract_t<1, FMT>(c, a, ntiles); break;
    case 2: launch_extract_t<2, FMT>(c, a, ntiles); break;
    case 3: launch_extract_t<3, FMT>(c, a, ntiles); break;
    default: launch_extract_t<4, FMT>(c, a, ntiles); break;
  }
}

// global-table path: extract and insert in one kernel
static void launch_extract(kc_ctx *c, const ExtractArgs &a, unsigned ntiles, int fmt) {
  KernelTimer kt(c, KT_EXTRACT_INSERT);
  if (fmt == FMT_READS) launch_extract_m<FMT_READS>(c, a, ntiles);
  else if (fmt == FMT_READS_UQ) launch_extract_m<FMT_READS_UQ>(c, a, ntiles);
  else if (fmt == FMT_PACKED) launch_extract_m<FMT_PACKED>(c, a, ntiles);
  else launch_extract_m<FMT_SEQBLOCK>(c, a, ntiles);
}

// One block of device-resident input through extraction, in chunks of tiles.  mode MODE_INSERT feeds this
// shard's own k-mers to the bucketed path (or the global table once the context is in table mode);
// MODE_BIN bins by owner shard into the caller's buffer.
// host_occ: the block's k-mer occurrences with two neighbours where the caller has counted them on the host (the host
// pipe: no question to the device per block then); NULL = ask the device
static int run_extract_device(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *d_offsets, uint64_t nreads,
                              uint64_t total, int mode, int fmt, uint64_t *d_records, uint64_t seg_capacity, const uint64_t *host_occ = nullptr) {
  if (total == 0) return KC_OK;
  ExtractArgs a;
  memset(&a, 0, sizeof(a));
  a.align = (uint32_t)((uintptr_t)bases & 15u);
  a.bases = bases - a.align;
  if (fmt == FMT_READS) {
    a.quals = quals - a.align;
    if (((uintptr_t)a.quals & 15u) != 0) fmt = FMT_READS_UQ;  // not co-aligned with the bases: the byte-load instantiation
  }
  if (fmt != FMT_SEQBLOCK) a.offsets = d_offsets;
  a.nreads = nreads;
  a.total = total;
  a.k = c->k;
  a.qual_cut = c->cfg.qual_offset + KC_QUAL_CUTOFF;
  a.rank_me = (uint32_t)c->cfg.rank_me;
  a.rank_n = (uint32_t)c->cfg.rank_n;
  a.reference_owner = (c->cfg.flags & KC_FLAG_REFERENCE_OWNER) ? 1u : 0u;
  a.records = d_records;
  a.seg_capacity = seg_capacity;
  const int64_t end = (int64_t)a.align + (int64_t)total;  // aligned coordinate one past the last real byte
  bool over_capacity = false;
  if (mode == MODE_INSERT) {
    c->started = true;
    if (bk_active(c)) {
      int rc = bk_init(c);
      if (rc) return rc;
      uint64_t total;
      if (host_occ && c->expect_host_ok) {
        // (a bad character in an earlier block is reported by kc_flush / kc_finalize instead of by this call)
        c->expect_host += *host_occ;
        total = c->expect_host;
      } else {
        rc = sync_ctrs(c);
        if (rc) return rc;
        if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
        // the stats kernel of this block has run: CTR_EXPECT counts every occurrence submitted so far, this block included
        total = c->h_ctrs[CTR_EXPECT];
        c->expect_host = total;
        c->expect_host_ok = true;
      }
      over_capacity = total - c->expect_base > c->bk_capacity;
      // The buffer cannot take this block on top of what it holds: count what it holds now, merge the counted k-mers
      // into the global table and go on with an empty buffer on the fast path (the reference streams insert blocks of
      // 1 MB into its one table for as long as reads come, gpu_hash_table.cpp:681-695).  A block that is too large for
      // the whole buffer takes the table path as before.
      if (over_capacity && !c->sh.flow && c->expect_prev > c->expect_base && total - c->expect_prev <= c->bk_capacity) {
        // Compact records first try the light way: what level 1 holds goes through level 2 now (an instalment) and
        // level 1 starts again empty -- its 8-byte records were the larger part of the buffer, the 4-byte records of level
        // 2 stay until the regions are counted, once, at the end.  Nothing is counted twice, nothing merged: a buffer for
        // 30 % of the input costs what one pass costs (scripts/spill_probe.py).  KC_ERR_OUT_OF_MEMORY: no room for level
        // 2 to grow -- then, as for longer k-mers, the counted buffer is merged into the global table.
        rc = bk_light_spill(c, c->expect_prev - c->expect_base);
        if (rc == KC_ERR_OUT_OF_MEMORY || rc == KC_ERR_UNSUPPORTED_K) rc = bk_spill_pass(c);
        if (rc) return rc;
        c->expect_base = c->expect_prev;
        over_capacity = false;
      }
      c->expect_prev = total;
    } else {
      c->expect_host_ok = false;  // the device goes on counting, the host does not follow
    }
  }
  int64_t p0 = 0;  // tiles of every launch start here; the bucketed kernels and the table kernel differ in tile span
  while (p0 < end) {
    const bool bk = (mode == MODE_INSERT) && bk_active(c);
    const bool super = bk || mode == MODE_BIN;
    const uint64_t span = super ? (uint64_t)SUPER_SPAN : (uint64_t)TILE;
    uint64_t chunk_tiles;
    if (super) {
      chunk_tiles = (1ULL << 31) / span;
    } else {
      // a chunk may add at most one new entry per position: keep it within a quarter of the table
      chunk_tiles = std::max<uint64_t>(64, (c->capacity / 4) / span);
      chunk_tiles = std::min<uint64_t>(chunk_tiles, (1ULL << 28) / span);
    }
    uint64_t nt = std::min<uint64_t>(chunk_tiles, ((uint64_t)(end - p0) + span - 1) / span);
    if (bk && over_capacity && c->sh.flow) {
      snprintf(g_last_error, sizeof(g_last_error), "shard flow: more k-mers than max_kmers_buffered (%llu): raise it, or use kc_extract_partition / kc_insert_records",
               (unsigned long long)c->bk_capacity);
      return KC_ERR_CAPACITY;
    }
    if (bk && over_capacity) {
      int rc = bk_drain_to_table(c);  // out of buffer room: this and every later chunk take the table path
      if (rc) return rc;
      continue;
    }
    if (bk) {  // no launch may overflow more records than the overflow list has room for
      uint64_t room = 0;
      int rc = bk_ovf1_room(c, nt * span, &room);
      if (rc) return rc;
      nt = std::max<uint64_t>(1, std::min<uint64_t>(nt, room / span));
      c->ovf1_ub += nt * span;  // what this launch can add to the list at most
      if (room < span) {  // a list smaller than one super-tile (test geometries only): nothing is bounded by it
        snprintf(g_last_error, sizeof(g_last_error), "overflow list smaller than one tile of %llu positions: raise ovf_capacity", (unsigned long long)span);
        return KC_ERR_CAPACITY;
      }
    }
    if (mode == MODE_INSERT && !bk) {
      int rc = ensure_room(c, nt * span);
      if (rc) return rc;
    }
    if (fmt != FMT_SEQBLOCK) {
      int rc = ensure_tile_first(c, nt);
      if (rc) return rc;
      {
        KernelTimer kt(c, KT_TILE_FIRST);
        hipLaunchKernelGGL(kc_tile_first_kernel, dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, c->stream, d_offsets, nreads,
                           a.align, p0, (uint32_t)span, nt, c->d_tile_first);
      }
      a.tile_first = c->d_tile_first;
    }
    a.pos0 = p0;
    if (bk) {
      int rc = launch_l1_reads(c, a, nt, fmt);
      if (rc) return rc;
    } else if (mode == MODE_BIN) {
      int rc = launch_bin_reads(c, a, nt, fmt);
      if (rc) return rc;
    } else {
      launch_extract(c, a, (unsigned)nt, fmt);
    }
    HIPCHK(hipGetLastError());
    p0 += (int64_t)(nt * span);
    if (fmt != FMT_SEQBLOCK && p0 < end) HIPCHK(hipStreamSynchronize(c->stream));  // d_tile_first is reused
  }
  return KC_OK;
}

static int raw_kmer_stats(kc_ctx *c, const uint64_t *d_offsets, uint64_t nreads, int mode) {
  if (!nreads) return KC_OK;
  unsigned nblk = (unsigned)std::min<uint64_t>((nreads + 255) / 256, 4096);
  hipLaunchKernelGGL(kc_read_stats_kernel, dim3(nblk), dim3(256), 0, c->stream, d_offsets, nreads, c->k, c->d_ctrs,
                     mode == MODE_INSERT ? 1u : 0u);
  c->num_gpu_calls++;
  HIPCHK(hipGetLastError());
  return KC_OK;
}

static int ensure_stage(kc_ctx *c, size_t bytes, size_t reads, bool need_quals) {
  if (bytes + 64 > c->stage_bytes) {
    if (c->d_stage_bases) HIPCHK(hipFree(c->d_stage_bases));
    if (c->d_stage_quals) HIPCHK(hipFree(c->d_stage_quals));
    c->d_stage_bases = c->d_stage_quals = nullptr;
    c->stage_bytes = 0;
    HIPCHK(hipMalloc((void **)&c->d_stage_bases, bytes + 64));
    HIPCHK(hipMalloc((void **)&c->d_stage_quals, bytes + 64));
    c->stage_bytes = bytes + 64;
  }
  (void)need_quals;
  if (reads + 1 > c->stage_reads) {
    if (c->d_stage_offsets) HIPCHK(hipFree(c->d_stage_offsets));
    c->d_stage_offsets = nullptr;
    c->stage_reads = 0;
    HIPCHK(hipMalloc((void **)&c->d_stage_offsets, (reads + 1) * 8));
    c->stage_reads = reads + 1;
  }
  return KC_OK;
}

// ---- host-resident reads: overlapped staging ---------------------------------------------------------------------
// The reference overlaps parsing with communication progress on a worker thread (src/kcount/kcount_gpu.cpp:119-133).
// Here the reads cross PCIe in blocks of HOST_BLOCK bytes of bases (+ as many of qualities): block i+1 is copied by a
// stream of its own while block i is extracted, events order the two streams, and the host never waits for a copy it
// has just issued.  Memory the caller has pinned (hipHostMalloc / hipHostRegister) is copied from where it lies;
// pageable memory goes through two pinned slots filled by a few threads (one thread's memcpy is slower than the link).
// KC_HOST_BLOCK (bytes) in the environment overrides the block size: tests use it to push small inputs through many blocks
static size_t host_block_bytes() {
  const char *e = getenv("KC_HOST_BLOCK");
  const unsigned long long x = e ? strtoull(e, nullptr, 10) : 0;
  return x >= 4096 ? (size_t)x : ((size_t)256 << 20);
}

static void host_pipe_free(kc_ctx *c) {
  auto &h = c->hp;
  for (int s = 0; s < 2; s++) {
    if (h.d_bases[s]) (void)hipFree(h.d_bases[s]);
    if (h.d_quals[s]) (void)hipFree(h.d_quals[s]);
    if (h.d_offs[s]) (void)hipFree(h.d_offs[s]);
    if (h.h_bases[s]) (void)hipHostFree(h.h_bases[s]);
    if (h.h_quals[s]) (void)hipHostFree(h.h_quals[s]);
    if (h.h_offs[s]) (void)hipHostFree(h.h_offs[s]);
    if (h.copied[s]) (void)hipEventDestroy(h.copied[s]);
    if (h.consumed[s]) (void)hipEventDestroy(h.consumed[s]);
    if (h.half[s]) (void)hipEventDestroy(h.half[s]);
  }
  if (h.copy_stream) (void)hipStreamDestroy(h.copy_stream);
  if (h.copy_stream2) (void)hipStreamDestroy(h.copy_stream2);
  memset(&h, 0, sizeof(h));
}

static int host_pipe_init(kc_ctx *c, size_t bytes, size_t reads, bool pinned_source) {
  auto &h = c->hp;
  if (h.ready && h.cap_bytes >= bytes && h.cap_reads >= reads && (pinned_source || h.h_bases[0])) return KC_OK;
  host_pipe_free(c);
  HIPCHK(hipStreamCreateWithFlags(&h.copy_stream, hipStreamNonBlocking));
  HIPCHK(hipStreamCreateWithFlags(&h.copy_stream2, hipStreamNonBlocking));
  for (int s = 0; s < 2; s++) {
    HIPCHK(hipEventCreateWithFlags(&h.half[s], hipEventDisableTiming));
    HIPCHK(hipMalloc((void **)&h.d_bases[s], bytes + 4096 + 64));  // (+ a page: copies start page-aligned in the source)
    HIPCHK(hipMalloc((void **)&h.d_quals[s], bytes + 4096 + 64));
    HIPCHK(hipMalloc((void **)&h.d_offs[s], (reads + 1) * 8));
    HIPCHK(hipHostMalloc((void **)&h.h_offs[s], (reads + 1) * 8, hipHostMallocDefault));
    if (!pinned_source) {
      HIPCHK(hipHostMalloc((void **)&h.h_bases[s], bytes, hipHostMallocDefault));
      HIPCHK(hipHostMalloc((void **)&h.h_quals[s], bytes, hipHostMallocDefault));
    }
    HIPCHK(hipEventCreateWithFlags(&h.copied[s], hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&h.consumed[s], hipEventDisableTiming));
  }
  h.cap_bytes = bytes;
  h.cap_reads = reads;
  h.ready = true;
  return KC_OK;
}

static bool is_pinned_host(const void *p) {
  hipPointerAttribute_t at;
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // pageable memory the runtime has never seen: not an error
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

static void parallel_copy(uint8_t *dst, const uint8_t *src, size_t n) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw / 2 : 2), n >> 22}));
  if (nt == 1) {
    memcpy(dst, src, n);
    return;
  }
  std::vector<std::thread> th;
  const size_t per = (n + nt - 1) / nt;
  for (size_t t = 0; t < nt; t++) {
    const size_t o = t * per;
    if (o >= n) break;
    th.emplace_back([=]() { memcpy(dst + o, src + o, std::min(per, n - o)); });
  }
  for (auto &t : th) t.join();
}

// k-mer occurrences with two neighbours of nr reads (what kc_read_stats_kernel adds to CTR_EXPECT), on a few threads
static uint64_t host_occurrences(const uint64_t *offsets, uint64_t nr, int k) {
  const unsigned hw = std::thread::hardware_concurrency();
  const size_t nt = std::max<size_t>(1, std::min<size_t>({(size_t)8, (size_t)(hw ? hw / 2 : 2), (size_t)(nr >> 16)}));
  std::vector<uint64_t> part(nt, 0);
  auto work = [&](size_t t) {
    uint64_t acc = 0;
    for (uint64_t r = nr * t / nt, e = nr * (t + 1) / nt; r < e; r++) {
      const uint64_t len = offsets[r + 1] - offsets[r];
      acc += len > (uint64_t)k + 1 ? len - (uint64_t)k - 1 : 0;
    }
    part[t] = acc;
  };
  if (nt == 1) {
    work(0);
  } else {
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; t++) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  uint64_t s = 0;
  for (uint64_t v : part) s += v;
  return s;
}

// a block's offsets, copied as the caller holds them, made relative to the block's first byte
__global__ void kc_rebase_offsets_kernel(uint64_t *offs, uint64_t n, uint64_t base) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) offs[i] -= base;
}

static int submit_host_reads(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads, int mode,
                             uint64_t *d_records, uint64_t seg_capacity, int fmt) {
  const size_t HOST_BLOCK = host_block_bytes();
  const bool with_quals = fmt == FMT_READS;
  const bool pinned = is_pinned_host(bases) && (!with_quals || is_pinned_host(quals));
  if (getenv("KC_DEBUG_ADDR")) {
    hipPointerAttribute_t at;
    memset(&at, 0, sizeof(at));
    const hipError_t e = hipPointerGetAttributes(&at, bases);
    (void)hipGetLastError();
    fprintf(stderr, "kc host input: %p pinned=%d (hipPointerGetAttributes: %s, type %d, device %d)\n", (const void *)bases, (int)pinned,
            hipGetErrorString(e), (int)at.type, at.device);
  }
  // blocks of whole reads: at most HOST_BLOCK bytes, and a single read that is longer gets a block of its own.  The end
  // of a block is found by bisection (the offsets only grow): walking the reads one by one, as round 2 did, cost the host
  // more per block than the copy took -- with the loop that rebased the offsets it made the pipe CPU-bound (47 GB/s over
  // a link that does 57)
  struct Blk { uint64_t r0, r1; };
  // (the first block is an eighth of the others: nothing runs on the device until it has arrived)
  const uint64_t total_bytes = offsets[nreads] - offsets[0];
  auto next_block = [&](uint64_t r0) {
    const uint64_t limit = offsets[r0] + ((r0 == 0 && total_bytes > 4 * (uint64_t)HOST_BLOCK) ? HOST_BLOCK / 8 : HOST_BLOCK);
    uint64_t r1 = (uint64_t)(std::upper_bound(offsets + r0 + 1, offsets + nreads + 1, limit) - offsets) - 1;  // last r1 with offsets[r1] <= limit
    if (r1 <= r0) r1 = r0 + 1;
    return Blk{r0, r1};
  };
  size_t max_bytes = 0, max_reads = 0;
  for (uint64_t r0 = 0; r0 < nreads;) {
    const Blk b = next_block(r0);
    max_bytes = std::max<size_t>(max_bytes, (size_t)(offsets[b.r1] - offsets[b.r0]));
    max_reads = std::max<size_t>(max_reads, (size_t)(b.r1 - b.r0));
    r0 = b.r1;
  }
  int rc = host_pipe_init(c, max_bytes, max_reads, pinned);
  if (rc) return rc;
  auto &h = c->hp;
  // stage: host side of one block + its copies (asynchronous, on the copy stream)
  auto stage = [&](int s, const Blk &b) -> int {
    if (h.used[s]) HIPCHK(hipEventSynchronize(h.consumed[s]));  // the kernels of two blocks ago have read slot s
    const uint64_t nb = offsets[b.r1] - offsets[b.r0], nr = b.r1 - b.r0;
    // the block's offsets as they are (a plain copy, a few threads); the device makes them relative to the block
    parallel_copy(reinterpret_cast<uint8_t *>(h.h_offs[s]), reinterpret_cast<const uint8_t *>(offsets + b.r0), (nr + 1) * 8);
    const uint8_t *sb = bases + offsets[b.r0], *sq = with_quals ? quals + offsets[b.r0] : nullptr;
    if (!pinned) {
      parallel_copy(h.h_bases[s], sb, nb);
      sb = h.h_bases[s];
      if (with_quals) {
        parallel_copy(h.h_quals[s], sq, nb);
        sq = h.h_quals[s];
      }
    }
    // two copies at a time, on two streams (one copy alone reached 46-50 GB/s of the link's 63): bases and qualities,
    // or the two halves of a block without qualities (the read cache's bytes)
    // A copy whose source starts on a page boundary runs at the link's rate (57 GB/s measured), one that starts at an
    // arbitrary byte -- a block starts where a read starts -- at 47: start every copy at the page boundary below (the
    // bytes in front belong to the caller's same buffer, `lead` of them, skipped on the device)
    auto lead_of = [&](const uint8_t *p, const uint8_t *buffer_start) -> uint32_t {
      const uint32_t l = (uint32_t)((uintptr_t)p & 4095u);
      return (size_t)(p - buffer_start) >= l ? l : 0u;
    };
    h.lead_b[s] = lead_of(sb, pinned ? bases : h.h_bases[s]);
    h.lead_q[s] = with_quals ? lead_of(sq, pinned ? quals : h.h_quals[s]) : 0u;
    if (with_quals) {
      // The two leads come from two buffers of the caller's: when they differ modulo 16 the qualities land up to 15 bytes
      // further into their slot (the source of the copy stays on its page boundary), so that a read's bases and
      // qualities sit at device addresses equal modulo 16 and level 1 keeps its 16-byte loads (FMT_READS; otherwise
      // every block of such an input fell back to the byte loads of FMT_READS_UQ)
      const uint32_t shift = (h.lead_b[s] - h.lead_q[s]) & 15u;
      HIPCHK(hipMemcpyAsync(h.d_bases[s], sb - h.lead_b[s], nb + h.lead_b[s], hipMemcpyHostToDevice, h.copy_stream));
      HIPCHK(hipMemcpyAsync(h.d_quals[s] + shift, sq - h.lead_q[s], nb + h.lead_q[s], hipMemcpyHostToDevice, h.copy_stream2));
      h.lead_q[s] += shift;
    } else {
      const uint64_t tot = nb + h.lead_b[s], h1 = (tot / 2) & ~(uint64_t)4095;
      const uint8_t *src = sb - h.lead_b[s];
      if (h1) HIPCHK(hipMemcpyAsync(h.d_bases[s], src, h1, hipMemcpyHostToDevice, h.copy_stream));
      HIPCHK(hipMemcpyAsync(h.d_bases[s] + h1, src + h1, tot - h1, hipMemcpyHostToDevice, h.copy_stream2));
    }
    HIPCHK(hipEventRecord(h.half[s], h.copy_stream2));
    HIPCHK(hipMemcpyAsync(h.d_offs[s], h.h_offs[s], (nr + 1) * 8, hipMemcpyHostToDevice, h.copy_stream));
    hipLaunchKernelGGL(kc_rebase_offsets_kernel, dim3((unsigned)((nr + 256) / 256)), dim3(256), 0, h.copy_stream, h.d_offs[s], nr + 1, offsets[b.r0]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamWaitEvent(h.copy_stream, h.half[s], 0));
    HIPCHK(hipEventRecord(h.copied[s], h.copy_stream));
    h.used[s] = true;
    return KC_OK;
  };
  const char *ie = getenv("KC_L2_INSTALMENTS");
  const bool instalments = !(ie && ie[0] == '0');
  Blk cur = next_block(0);
  rc = stage(0, cur);
  if (rc) return rc;
  for (int i = 0;; i++) {
    const int s = i & 1;
    const bool more = cur.r1 < nreads;
    Blk nxt{0, 0};
    if (more) {  // the next block's copy is under way before this block's kernels are even launched
      nxt = next_block(cur.r1);
      rc = stage(s ^ 1, nxt);
      if (rc) return rc;
    }
    const uint64_t nb = offsets[cur.r1] - offsets[cur.r0], nr = cur.r1 - cur.r0;
    HIPCHK(hipStreamWaitEvent(c->stream, h.copied[s], 0));
    rc = raw_kmer_stats(c, h.d_offs[s], nr, mode);
    if (rc) return rc;
    const uint64_t occ = host_occurrences(offsets + cur.r0, nr, c->k);
    rc = run_extract_device(c, h.d_bases[s] + h.lead_b[s], h.d_quals[s] + h.lead_q[s], h.d_offs[s], nr, nb, mode, fmt, d_records, seg_capacity,
                            mode == MODE_INSERT ? &occ : nullptr);
    if (rc) return rc;
    HIPCHK(hipEventRecord(h.consumed[s], c->stream));
    c->num_reads += nr;
    c->num_bases += nb;
    if (!more) break;
    // While the rest of the input is still crossing PCIe the device has time on its hands (the copies take four times
    // as long as level 1): level 2 takes what has arrived, in instalments, so that when the last byte is in only the last
    // block's records and the count kernel are left (end to end 186 -> 163 ms per 50 M packed reads).
    if (mode == MODE_INSERT && instalments) {
      rc = bk_level2_instalment(c);
      if (rc) return rc;
    }
    cur = nxt;
  }
  // the caller's arrays are free to change once every copy has left them
  HIPCHK(hipStreamSynchronize(h.copy_stream));
  HIPCHK(hipStreamSynchronize(h.copy_stream2));
  return KC_OK;
}

// reads (either residence) through extract in `mode`
static int submit_reads_impl(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads,
                             int on_device, int mode, uint64_t *d_records, uint64_t seg_capacity, int fmt = FMT_READS) {
  if (!c || (nreads && (!bases || (fmt == FMT_READS && !quals) || !offsets))) return KC_ERR_INVALID_ARG;
  if ((c->finalized || c->bk_level2) && mode == MODE_INSERT) return KC_ERR_STATE;  // extraction alone never touches the table
  if (mode == MODE_INSERT && shard_flow_only(c) && !c->sh.extracting) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!nreads) return KC_OK;
  if (on_device) {
    uint64_t ends[2];
    HIPCHK(hipMemcpyAsync(&ends[0], offsets, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(&ends[1], offsets + nreads, 8, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (ends[0] != 0) return KC_ERR_INVALID_ARG;  // offsets are relative to `bases`
    int rc = raw_kmer_stats(c, offsets, nreads, mode);
    if (rc) return rc;
    rc = run_extract_device(c, bases, quals, offsets, nreads, ends[1], mode, fmt, d_records, seg_capacity);
    if (rc) return rc;
    c->num_reads += nreads;
    c->num_bases += ends[1];
    return KC_OK;
  }
  return submit_host_reads(c, bases, quals, offsets, nreads, mode, d_records, seg_capacity, fmt);
}

extern "C" int kc_submit_reads(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads,
                               int on_device) {
  return submit_reads_impl(c, bases, quals, offsets, nreads, on_device, MODE_INSERT, nullptr, 0);
}

extern "C" int kc_submit_packed_reads(kc_ctx *c, const uint8_t *packed, const uint64_t *offsets, uint64_t nreads, int on_device) {
  return submit_reads_impl(c, packed, nullptr, offsets, nreads, on_device, MODE_INSERT, nullptr, 0, FMT_PACKED);
}

// ---- FASTQ front end (host only) -----------------------------------------------------------------------------------
extern "C" int kc_fastq_to_packed(const char *text, uint64_t len, int qual_offset, uint8_t *packed, uint64_t packed_capacity,
                                  uint64_t *offsets, uint64_t reads_capacity, uint64_t *nreads, uint64_t *nbytes) {
  if ((len && !text) || !nreads || !nbytes) return KC_ERR_INVALID_ARG;
  // base codes of PackedRead (packed_reads.cpp:99-124): 255 = the reference DIEs
  uint8_t code[256];
  memset(code, 255, sizeof(code));
  const char *acgt = "ACGT";
  for (int i = 0; i < 4; i++) code[(uint8_t)acgt[i]] = code[(uint8_t)(acgt[i] | 0x20)] = (uint8_t)i;
  code[(uint8_t)'N'] = code[(uint8_t)'n'] = 4;
  for (const char *p = "URYKMSWBDHV"; *p; p++) code[(uint8_t)*p] = 4;
  uint64_t nr = 0, nb = 0, pos = 0, line_no = 0;
  bool fits = true;
  auto next_line = [&](uint64_t &b, uint64_t &e) -> bool {  // [b, e): the line without its end and trailing white space
    if (pos >= len) return false;
    b = pos;
    while (pos < len && text[pos] != '\n') pos++;
    e = pos;
    if (pos < len) pos++;
    while (e > b && (text[e - 1] == '\r' || text[e - 1] == ' ' || text[e - 1] == '\t')) e--;
    line_no++;
    return true;
  };
  if (offsets && reads_capacity + 1 > 0) offsets[0] = 0;
  for (;;) {
    uint64_t b0, e0, b1, e1, b2, e2, b3, e3;
    if (!next_line(b0, e0)) break;
    if (e0 == b0 && pos >= len) break;  // a final empty line
    if (!next_line(b1, e1) || !next_line(b2, e2) || !next_line(b3, e3)) {
      snprintf(g_last_error, sizeof(g_last_error), "FASTQ ends inside the record that starts at line %llu", (unsigned long long)(line_no - (line_no - 1) % 4));
      return KC_ERR_INVALID_ARG;
    }
    if (e0 == b0 || text[b0] != '@') {
      snprintf(g_last_error, sizeof(g_last_error), "Invalid FASTQ: expected read name (@) at line %llu", (unsigned long long)(line_no - 3));
      return KC_ERR_INVALID_ARG;
    }
    if (e2 == b2 || text[b2] != '+') {
      snprintf(g_last_error, sizeof(g_last_error), "Invalid FASTQ: expected '+' at line %llu", (unsigned long long)(line_no - 1));
      return KC_ERR_INVALID_ARG;
    }
    const uint64_t sl = e1 - b1;
    if (sl != e3 - b3) {
      snprintf(g_last_error, sizeof(g_last_error), "Invalid FASTQ: sequence length %llu != %llu quals length at line %llu",
               (unsigned long long)sl, (unsigned long long)(e3 - b3), (unsigned long long)(line_no - 2));
      return KC_ERR_INVALID_ARG;
    }
    const bool room = fits && packed && offsets && nr < reads_capacity && nb + sl <= packed_capacity;
    for (uint64_t i = 0; i < sl; i++) {
      const uint8_t cb = code[(uint8_t)text[b1 + i]];
      if (cb == 255) {
        snprintf(g_last_error, sizeof(g_last_error), "Illegal char in comp nucleotide (int=%d) at line %llu", (int)(uint8_t)text[b1 + i],
                 (unsigned long long)(line_no - 2));
        return KC_ERR_BAD_BASE;
      }
      if (room) {
        int q = (int)(uint8_t)text[b3 + i] - qual_offset;
        if (q > 31) q = 31;
        packed[nb + i] = (uint8_t)(cb | ((uint8_t)q << 3));  // like the reference's (unsigned char)std::min(q, 31) << 3
      }
    }
    if (!room) fits = false;
    nb += sl;
    nr++;
    if (room) offsets[nr] = nb;
  }
  *nreads = nr;
  *nbytes = nb;
  if (!fits && (nr || nb)) {
    snprintf(g_last_error, sizeof(g_last_error), "%llu reads with %llu bases do not fit the arrays", (unsigned long long)nr, (unsigned long long)nb);
    return KC_ERR_CAPACITY;
  }
  return KC_OK;
}

extern "C" int kc_submit_seq_block(kc_ctx *c, const char *seqs, uint64_t len, int on_device) {
  if (!c || (len && !seqs)) return KC_ERR_INVALID_ARG;
  if (c->finalized || c->bk_level2 || (shard_flow_only(c) && !c->sh.extracting)) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!len) return KC_OK;
  const uint8_t *d = (const uint8_t *)seqs;
  if (!on_device) {
    int rc = ensure_stage(c, (size_t)len, 0, false);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyAsync(c->d_stage_bases, seqs, len, hipMemcpyHostToDevice, c->stream));
    d = c->d_stage_bases;
  }
  unsigned nblk = (unsigned)std::min<uint64_t>(((len + SEQSTAT_SPAN - 1) / SEQSTAT_SPAN + 3) / 4, 4096);  // four waves each
  hipLaunchKernelGGL(kc_seqblock_stats_kernel, dim3(nblk), dim3(256), 0, c->stream, d, len, c->k, c->d_ctrs, 1u);
  c->num_gpu_calls++;
  int rc = run_extract_device(c, d, nullptr, nullptr, 0, len, MODE_INSERT, FMT_SEQBLOCK, nullptr, 0);
  if (rc) return rc;
  c->num_bases += len;
  if (!on_device) HIPCHK(hipStreamSynchronize(c->stream));  // caller's buffer is free to change
  return KC_OK;
}

// ---- the reference's wire format ----------------------------------------------------------------------------------
template <typename T>
static int grow_dev(T **p, size_t *cap, size_t need) {
  if (need <= *cap) return KC_OK;
  if (*p) HIPCHK(hipFree(*p));
  *p = nullptr;
  *cap = 0;
  HIPCHK(hipMalloc((void **)p, need * sizeof(T)));
  *cap = need;
  return KC_OK;
}

extern "C" int kc_build_supermers(kc_ctx *c, const char *seqs, uint64_t len, int on_device, kc_supermer *out, uint32_t capacity,
                                  uint32_t *n_out, uint32_t *num_valid_kmers, uint8_t *packed_out) {
  static_assert(sizeof(kc_supermer) == sizeof(SupermerInfo) && sizeof(kc_supermer) == 12, "layout of kcount_gpu::SupermerInfo");
  if (!c || !n_out || (len && !seqs) || (capacity && !out) || len >= (1ULL << 31)) return KC_ERR_INVALID_ARG;
  *n_out = 0;
  if (num_valid_kmers) *num_valid_kmers = 0;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!len) return KC_OK;
  int rc = grow_dev(&c->d_sm_targets, &c->sm_targets_cap, (size_t)len);
  if (!rc) rc = grow_dev(&c->d_sm_out, &c->sm_out_cap, (size_t)std::max<uint32_t>(capacity, 1));
  if (!rc) rc = grow_dev(&c->d_sm_packed, &c->sm_packed_cap, (size_t)(len + 1) / 2);
  if (!rc && !on_device) rc = grow_dev(&c->d_sm_bytes, &c->sm_bytes_cap, (size_t)len);
  if (rc) return rc;
  if (!c->d_sm_ctr) HIPCHK(hipMalloc((void **)&c->d_sm_ctr, 32));
  HIPCHK(hipMemsetAsync(c->d_sm_ctr, 0, 32, c->stream));
  const uint8_t *d = (const uint8_t *)seqs;
  if (!on_device) {
    HIPCHK(hipMemcpyAsync(c->d_sm_bytes, seqs, len, hipMemcpyHostToDevice, c->stream));
    d = c->d_sm_bytes;
  }
  uint64_t *d_bad = (uint64_t *)(c->d_sm_ctr + 4);
  if (c->k > SM_HALO - 1) return KC_ERR_UNSUPPORTED_K;
  hipLaunchKernelGGL(kc_supermer_targets_kernel, dim3((unsigned)((len + SM_TILE - 1) / SM_TILE)), dim3(SM_WG), 0, c->stream, d, len, c->k,
                     (uint32_t)c->cfg.rank_n, c->d_sm_targets, d_bad);
  hipLaunchKernelGGL(kc_supermer_build_kernel, dim3((unsigned)((len + SB_WG * SB_PER - 1) / (SB_WG * SB_PER))), dim3(SB_WG), 0, c->stream, c->d_sm_targets, len, c->k, c->d_sm_out, capacity,
                     c->d_sm_ctr, c->d_sm_ctr + 1, c->d_sm_ctr + 2);
  hipLaunchKernelGGL(kc_pack_seqs_kernel, dim3((unsigned)(((len + 15) / 16 + 255) / 256)), dim3(256), 0, c->stream, d, len, c->d_sm_packed);
  c->num_gpu_calls += 3;
  HIPCHK(hipGetLastError());
  uint32_t h[6];
  HIPCHK(hipMemcpyAsync(h, c->d_sm_ctr, 24, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  if (h[4] | h[5]) return KC_ERR_BAD_BASE;
  if (h[2]) {
    snprintf(g_last_error, sizeof(g_last_error), "a supermer is longer than 65535 characters");
    return KC_ERR_INVALID_ARG;
  }
  *n_out = h[0];
  if (num_valid_kmers) *num_valid_kmers = h[1];
  if (h[0] > capacity) {
    snprintf(g_last_error, sizeof(g_last_error), "%u supermers, room for %u", h[0], capacity);
    return KC_ERR_CAPACITY;
  }
  if (h[0]) HIPCHK(hipMemcpy(out, c->d_sm_out, (size_t)h[0] * sizeof(kc_supermer), hipMemcpyDeviceToHost));
  if (packed_out) HIPCHK(hipMemcpy(packed_out, c->d_sm_packed, (size_t)(len + 1) / 2, hipMemcpyDeviceToHost));
  return KC_OK;
}

extern "C" int kc_submit_packed_supermers(kc_ctx *c, const uint8_t *packed, uint64_t len, int on_device) {
  if (!c || (len && !packed)) return KC_ERR_INVALID_ARG;
  if (c->finalized || c->bk_level2) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!len) return KC_OK;
  // the unpacked block of the previous call may still be read by its kernels
  HIPCHK(hipStreamSynchronize(c->stream));
  int rc = grow_dev(&c->d_sm_bytes, &c->sm_bytes_cap, (size_t)len * 2);
  if (!rc && !on_device) rc = grow_dev(&c->d_sm_packed, &c->sm_packed_cap, (size_t)len);
  if (rc) return rc;
  if (!c->d_sm_ctr) HIPCHK(hipMalloc((void **)&c->d_sm_ctr, 32));
  HIPCHK(hipMemsetAsync(c->d_sm_ctr, 0, 32, c->stream));
  const uint8_t *d = packed;
  if (!on_device) {
    HIPCHK(hipMemcpyAsync(c->d_sm_packed, packed, len, hipMemcpyHostToDevice, c->stream));
    d = c->d_sm_packed;
  }
  uint64_t *d_bad = (uint64_t *)(c->d_sm_ctr + 4);
  hipLaunchKernelGGL(kc_unpack_supermers_kernel, dim3((unsigned)(((len + 7) / 8 + 255) / 256)), dim3(256), 0, c->stream, d, len, c->d_sm_bytes, d_bad);
  c->num_gpu_calls++;
  HIPCHK(hipGetLastError());
  uint32_t h[2];
  HIPCHK(hipMemcpyAsync(h, d_bad, 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));  // also: the caller's buffer is free to change
  if (h[0] | h[1]) return KC_ERR_BAD_BASE;  // a nibble above 9 (reference: WARN "index out of range for to_base")
  return kc_submit_seq_block(c, (const char *)c->d_sm_bytes, len * 2, 1);
}

// the two ends of a kc_extract_partition call.  With KC_FLAG_WIRE_UNITS the geometry decides what a unit is, so it is
// chosen now (bk_init); the counters then count slots of six bytes, four to a unit.
static int bin_begin(kc_ctx *c) {
  if ((c->cfg.flags & KC_FLAG_WIRE_UNITS) && bk_active(c)) {
    int rc = bk_init(c);
    if (rc) return rc;
  }
  HIPCHK(hipMemsetAsync(c->d_ctrs + CTR_OVERFLOW, 0, (1 + 64) * 8, c->stream));
  if (c->wire6) {
    if (!c->d_w6cur) HIPCHK(hipMalloc((void **)&c->d_w6cur, sizeof(c->h_w6cur)));
    HIPCHK(hipMemsetAsync(c->d_w6cur, 0, sizeof(c->h_w6cur), c->stream));
  }
  return KC_OK;
}
static int bin_end(kc_ctx *c, uint64_t *d_records, uint64_t seg_capacity, uint64_t *h_counts) {
  const uint32_t np = (uint32_t)c->cfg.rank_n * bin_pieces(c);
  if (c->wire6) {  // every piece closed to whole units
    hipLaunchKernelGGL(kc_wire6_seal_kernel, dim3((np + 63) / 64), dim3(64), 0, c->stream, c->d_w6cur, np, reinterpret_cast<uint8_t *>(d_records),
                       seg_capacity * WIRE6_UNIT_RECORDS);
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->h_w6cur, c->d_w6cur, (size_t)np * 8, hipMemcpyDeviceToHost, c->stream));
  }
  int rc = sync_ctrs(c);
  if (rc) return rc;
  for (uint32_t j = 0; j < np; j++) h_counts[j] = c->wire6 ? c->h_w6cur[j] / WIRE6_UNIT_RECORDS : c->h_ctrs[CTR_BIN0 + j];
  if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
  if (c->h_ctrs[CTR_OVERFLOW]) {
    snprintf(g_last_error, sizeof(g_last_error), "a shard segment is too small: raise seg_capacity (%llu %s)", (unsigned long long)seg_capacity,
             c->wire6 ? "units of four records per piece" : "records");
    return KC_ERR_CAPACITY;
  }
  return KC_OK;
}

extern "C" int kc_extract_partition(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads,
                                    int on_device, uint64_t *d_records, uint64_t seg_capacity, uint64_t *h_counts) {
  if (!c || !h_counts || (nreads && !d_records)) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bin_begin(c);
  if (rc) return rc;
  rc = submit_reads_impl(c, bases, quals, offsets, nreads, on_device, MODE_BIN, d_records, seg_capacity);
  if (rc) return rc;
  return bin_end(c, d_records, seg_capacity, h_counts);
}

extern "C" int kc_extract_partition_seq_block(kc_ctx *c, const char *seqs, uint64_t len, int on_device, uint64_t *d_records,
                                              uint64_t seg_capacity, uint64_t *h_counts) {
  if (!c || !h_counts || (len && (!seqs || !d_records))) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bin_begin(c);
  if (rc) return rc;
  const uint8_t *d = (const uint8_t *)seqs;
  if (len && !on_device) {
    rc = ensure_stage(c, (size_t)len, 0, false);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyAsync(c->d_stage_bases, seqs, len, hipMemcpyHostToDevice, c->stream));
    d = c->d_stage_bases;
  }
  rc = run_extract_device(c, d, nullptr, nullptr, 0, len, MODE_BIN, FMT_SEQBLOCK, d_records, seg_capacity);
  if (rc) return rc;
  return bin_end(c, d_records, seg_capacity, h_counts);
}

template <int NL>
static void launch_insert_records(kc_ctx *c, const uint64_t *recs, uint64_t n, uint32_t count_inserted) {
  unsigned nblk = (unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 256 * 32);
  hipLaunchKernelGGL(kc_insert_records_kernel<NL>, dim3(nblk), dim3(TPB), 0, c->stream, recs, n, c->table, c->d_ctrs, count_inserted);
}

// records straight into the global table, growing it as needed
static int table_insert_records(kc_ctx *c, const uint64_t *d_records, uint64_t n, uint32_t count_inserted) {
  uint64_t done = 0;
  while (done < n) {
    uint64_t m = std::min<uint64_t>(n - done, std::max<uint64_t>(c->capacity / 4, 1u << 18));
    int rc = ensure_room(c, m);
    if (rc) return rc;
    const uint64_t *p = d_records + done * c->nl;
    {
      KernelTimer kt(c, KT_INSERT_RECORDS);
      switch (c->nl) {
        case 1: launch_insert_records<1>(c, p, m, count_inserted); break;
        case 2: launch_insert_records<2>(c, p, m, count_inserted); break;
        case 3: launch_insert_records<3>(c, p, m, count_inserted); break;
        default: launch_insert_records<4>(c, p, m, count_inserted); break;
      }
    }
    HIPCHK(hipGetLastError());
    done += m;
  }
  return KC_OK;
}

extern "C" int kc_insert_records(kc_ctx *c, const uint64_t *d_records, uint64_t n) {
  if (!c || (n && !d_records)) return KC_ERR_INVALID_ARG;
  if (c->finalized || c->bk_level2 || shard_flow_only(c)) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!n) return KC_OK;
  c->started = true;
  // wire units (KC_FLAG_WIRE_UNITS with a geometry of six-byte records): n counts units of four slots, markers among them
  if (bk_active(c)) {
    int rc = bk_init(c);
    if (rc) return rc;
  }
  const bool w6 = c->wire6;
  if (w6) n *= WIRE6_UNIT_RECORDS;  // slots from here on
  if (bk_active(c)) {
    int rc = sync_ctrs(c);
    if (rc) return rc;
    const uint64_t buffered = c->h_ctrs[CTR_EXPECT] - c->expect_base;
    bool fits = buffered + n <= c->bk_capacity;
    if (!fits && buffered && n <= c->bk_capacity) {  // hand what is buffered on to level 2 (compact records), or count it and merge it into the table; go on empty
      rc = bk_light_spill(c, buffered);
      if (rc == KC_ERR_OUT_OF_MEMORY || rc == KC_ERR_UNSUPPORTED_K) rc = bk_spill_pass(c);
      if (rc) return rc;
      c->expect_base = c->h_ctrs[CTR_EXPECT];
      fits = true;
    }
    c->expect_prev = c->h_ctrs[CTR_EXPECT] + n;
    c->expect_host_ok = false;  // the records kernel adds its own count on the device
    if (!fits) {
      rc = bk_drain_to_table(c);
      if (rc) return rc;
    } else {
      // pieces no larger than the overflow list's free room (bk_ovf1_room)
      uint64_t done = 0;
      while (done < n) {
        uint64_t room = 0;
        rc = bk_ovf1_room(c, n - done, &room);
        if (rc) return rc;
        uint64_t m = std::min<uint64_t>(n - done, room);
        if (!m) {
          snprintf(g_last_error, sizeof(g_last_error), "overflow list has no room: raise ovf_capacity");
          return KC_ERR_CAPACITY;
        }
        c->ovf1_ub += m;
        if (w6) {
          if (m < n - done) m &= ~(uint64_t)(WIRE6_UNIT_RECORDS - 1);  // pieces of whole units
          if (!m) {
            snprintf(g_last_error, sizeof(g_last_error), "overflow list has no room: raise ovf_capacity");
            return KC_ERR_CAPACITY;
          }
          rc = launch_l1_wire6(c, reinterpret_cast<const uint8_t *>(d_records) + done * 6, 0, 1, &m);
          if (rc) return rc;
          HIPCHK(hipGetLastError());
          done += m;
          continue;
        }
        const uint64_t *p = d_records + done * c->nl;
        switch (c->nl) {
          case 1: rc = launch_l1_records_t<1>(c, p, m); break;
          case 2: rc = launch_l1_records_t<2>(c, p, m); break;
          case 3: rc = launch_l1_records_t<3>(c, p, m); break;
          default: rc = launch_l1_records_t<4>(c, p, m); break;
        }
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        done += m;
      }
      return KC_OK;
    }
  }
  if (w6) {  // the context has left the bucketed path: back to k-mer records for the global table
    uint64_t *tmp = nullptr;
    HIPCHK(hipMalloc((void **)&tmp, (n + 1) * 8));
    int rc = KC_OK;
    unsigned long long h_n = 0;
    if (hipMemsetAsync(tmp + n, 0, 8, c->stream) != hipSuccess) rc = KC_ERR_HIP;
    if (!rc) {
      hipLaunchKernelGGL(kc_wire6_expand_kernel, dim3((unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 256 * 32)), dim3(TPB), 0, c->stream,
                         reinterpret_cast<const uint8_t *>(d_records), n, c->gm, tmp, (unsigned long long *)(tmp + n));
      c->num_gpu_calls++;
      if (hipMemcpyAsync(&h_n, tmp + n, 8, hipMemcpyDeviceToHost, c->stream) != hipSuccess || hipStreamSynchronize(c->stream) != hipSuccess) rc = KC_ERR_HIP;
    }
    if (!rc && h_n) rc = table_insert_records(c, tmp, h_n, 1u);
    if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = KC_ERR_HIP;
    (void)hipFree(tmp);
    return rc;
  }
  return table_insert_records(c, d_records, n, 1u);
}

extern "C" int kc_insert_record_pieces(kc_ctx *c, const uint64_t *d_records, uint64_t piece_stride_units, int npieces, const uint64_t *h_units) {
  if (!c || npieces < 0 || (npieces && (!d_records || !h_units))) return KC_ERR_INVALID_ARG;
  if (c->finalized || c->bk_level2 || shard_flow_only(c)) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (bk_active(c)) {
    int rc = bk_init(c);
    if (rc) return rc;
  }
  const uint64_t unit_words = c->wire6 ? WIRE6_UNIT_WORDS : (uint64_t)c->nl;
  // Wire units on the bucketed path: the pieces that hold anything in launches of up to sixteen -- a round of a workgroup
  // reads out of one piece --, as long as the buffer and the overflow list take them all; anything else piece by piece
  // through kc_insert_records, which knows what to do when the buffer is full
  if (c->wire6 && bk_active(c)) {
    std::vector<uint64_t> slots, first;
    uint64_t total = 0;
    bool small = true;
    for (int j = 0; j < npieces; j++) {
      if (!h_units[j]) continue;
      slots.push_back(h_units[j] * WIRE6_UNIT_RECORDS);
      first.push_back((uint64_t)j);
      total += slots.back();
      small = small && slots.back() < (1ULL << 32);
    }
    if (!total) return KC_OK;
    int rc = sync_ctrs(c);
    if (rc) return rc;
    uint64_t room = 0;
    rc = bk_ovf1_room(c, total, &room);
    if (rc) return rc;
    if (small && c->h_ctrs[CTR_EXPECT] - c->expect_base + total <= c->bk_capacity && room >= total) {
      c->started = true;
      c->expect_prev = c->h_ctrs[CTR_EXPECT] + total;
      c->expect_host_ok = false;
      c->ovf1_ub += total;
      // pieces that lie at multiples of the stride from the first of a group go in one launch
      size_t i = 0;
      while (i < slots.size()) {
        const size_t n = std::min<size_t>(WIRE6_SRC_PIECES, slots.size() - i);
        // (the table holds consecutive indices: pieces skipped because they are empty break a group)
        size_t m = 1;
        while (m < n && first[i + m] == first[i] + m) m++;
        rc = launch_l1_wire6(c, reinterpret_cast<const uint8_t *>(d_records + first[i] * piece_stride_units * unit_words),
                             piece_stride_units * unit_words * 8, (uint32_t)m, &slots[i]);
        if (rc) return rc;
        HIPCHK(hipGetLastError());
        i += m;
      }
      return KC_OK;
    }
  }
  for (int j = 0; j < npieces; j++) {
    if (!h_units[j]) continue;
    int rc = kc_insert_records(c, d_records + (uint64_t)j * piece_stride_units * unit_words, h_units[j]);
    if (rc) return rc;
  }
  return KC_OK;
}

extern "C" int kc_wire_unit(kc_ctx *c, int *unit_words, int *unit_records, int *pieces) {
  if (!c || !unit_words || !unit_records || !pieces) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  if ((c->cfg.flags & KC_FLAG_WIRE_UNITS) && bk_active(c)) {
    int rc = bk_init(c);  // the geometry decides
    if (rc) return rc;
  }
  *unit_words = c->wire6 ? (int)WIRE6_UNIT_WORDS : c->nl;
  *unit_records = c->wire6 ? (int)WIRE6_UNIT_RECORDS : 1;
  *pieces = (int)bin_pieces(c);
  return KC_OK;
}

extern "C" int kc_partition_owner(kc_ctx *c, const uint64_t *kmer, int *owner) {
  if (!c || !kmer || !owner) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  if ((c->cfg.flags & KC_FLAG_WIRE_UNITS) && bk_active(c)) {
    int rc = bk_init(c);
    if (rc) return rc;
  }
  if (c->wire6) {
    const uint64_t mix = kc_feistel_fwd(kmer[0] >> (64u - c->gm.k2), c->k);
    *owner = (int)wire6_owner((uint32_t)mix, (uint32_t)c->cfg.rank_n);
    return KC_OK;
  }
  const int o = (c->cfg.flags & KC_FLAG_REFERENCE_OWNER) ? kc_owner_reference(kmer, c->k, c->cfg.rank_n) : kc_owner(kmer, c->k, c->cfg.rank_n);
  if (o < 0) return o;
  *owner = o;
  return KC_OK;
}

// ---- shard flow: ownership by level-1 bucket (kernels and the wire format in kc_shard.hpp) -------------------
static uint64_t shard_signature(const kc_ctx *c) {
  return (0x4B53ULL << 48) | ((uint64_t)c->gm.P1 << 32) | ((uint64_t)c->k << 16) | ((uint64_t)c->nl << 12) | ((uint64_t)(c->gm.cp ? 1 : 0) << 8) |
         (uint64_t)c->cfg.rank_n;
}

static int shard_init(kc_ctx *c) {
  if (c->sh.d_plan) return KC_OK;
  HIPCHK(hipMalloc((void **)&c->sh.d_plan, ((size_t)PMAX + 4 * SHARD_MAX) * 8));
  HIPCHK(hipHostMalloc((void **)&c->sh.h_plan, (size_t)5 * SHARD_MAX * 8, hipHostMallocDefault));
  return KC_OK;
}

template <int NL>
static void launch_shard_pack(kc_ctx *c, uint64_t *segs, uint64_t seg_words, const uint64_t *off, const uint64_t *flags) {
  const uint32_t Q = 4;  // workgroups per bucket
  KernelTimer kt(c, KT_SHARD_PACK);
  auto kern = (NL == 1 && shard_wire_of(c->gm) == SHARD_WIRE_COMPACT) ? kc_shard_pack_kernel<NL, NL == 1> : kc_shard_pack_kernel<NL, false>;
  hipLaunchKernelGGL(kern, dim3(c->gm.P1 * Q), dim3(WGB), 0, c->stream, c->gm, c->bb, (uint32_t)c->cfg.rank_me, (uint32_t)c->cfg.rank_n, segs,
                     seg_words, off, flags, Q);
}

template <int NL>
static void launch_shard_route(kc_ctx *c, uint64_t n1, uint64_t *segs, uint64_t seg_words, const uint64_t *wtotals, uint64_t *loose, uint64_t *flags) {
  auto kern = use_cp<NL>(c) ? kc_shard_route_ovf1_kernel<NL, NL == 1> : kc_shard_route_ovf1_kernel<NL, false>;
  KernelTimer kt(c, KT_FALLBACK);
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>((n1 + TPB - 1) / TPB, 256 * 32)), dim3(TPB), 0, c->stream, c->gm, c->bb, n1,
                     (uint32_t)c->cfg.rank_me, (uint32_t)c->cfg.rank_n, segs, seg_words, wtotals, loose, flags, c->table, c->d_ctrs);
}

template <int NL>
static void launch_shard_loose(kc_ctx *c, const uint64_t *recs, uint64_t n) {
  auto kern = use_cp<NL>(c) ? kc_shard_loose_kernel<NL, NL == 1> : kc_shard_loose_kernel<NL, false>;
  KernelTimer kt(c, KT_FALLBACK);
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 256 * 32)), dim3(TPB), 0, c->stream, c->gm, c->bb, recs, n,
                     c->table, c->d_ctrs);
}

// checks and set-up shared by the two forms of kc_shard_extract
static int shard_extract_begin(kc_ctx *c, uint64_t *d_segments, uint64_t seg_words, uint64_t *h_words) {
  if (!c || !h_words) return KC_ERR_INVALID_ARG;
  const uint32_t n = (uint32_t)c->cfg.rank_n;
  if (n > 1 && (!d_segments || seg_words < SHARD_HDR + PMAX / 2)) return KC_ERR_INVALID_ARG;
  for (uint32_t d = 0; d < n; d++) h_words[d] = 0;
  if (c->finalized || c->bk_level2) return KC_ERR_STATE;
  if (!bk_active(c)) {
    snprintf(g_last_error, sizeof(g_last_error), "the shard flow needs the bucketed path (this context is on the global table): use kc_extract_partition / kc_insert_records");
    return KC_ERR_STATE;
  }
  if (n > 1 && c->started && !c->sh.flow) {
    snprintf(g_last_error, sizeof(g_last_error), "this pass already took k-mers by hash ownership (kc_submit_* / kc_insert_records): kc_reset first");
    return KC_ERR_STATE;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bk_init(c);
  if (rc) return rc;
  if (n > c->gm.P1) {
    snprintf(g_last_error, sizeof(g_last_error), "more shards (%u) than level-1 buckets (%u)", n, c->gm.P1);
    return KC_ERR_INVALID_ARG;
  }
  return KC_OK;
}

// after level 1 has taken the block: what other shards own leaves the chains for the wire segments
static int shard_extract_pack(kc_ctx *c, bool anything, uint64_t *d_segments, uint64_t seg_words, uint64_t *h_words) {
  const uint32_t n = (uint32_t)c->cfg.rank_n, me = (uint32_t)c->cfg.rank_me;
  int rc;
  if (n == 1 || !anything) return KC_OK;  // a single shard owns every bucket: nothing leaves
  rc = shard_init(c);
  if (rc) return rc;
  uint64_t *off = c->sh.d_plan, *totals = off + PMAX, *flags = totals + SHARD_MAX, *loose = flags + SHARD_MAX, *wtotals = loose + SHARD_MAX;
  HIPCHK(hipMemsetAsync(loose, 0, SHARD_MAX * 8, c->stream));
  hipLaunchKernelGGL(kc_shard_plan_kernel, dim3(1), dim3(WGB), 0, c->stream, c->gm, c->bb, me, n, d_segments, seg_words, shard_signature(c),
                     (uint32_t)c->nl, off, totals, flags, wtotals);
  switch (c->nl) {
    case 1: launch_shard_pack<1>(c, d_segments, seg_words, off, flags); break;
    case 2: launch_shard_pack<2>(c, d_segments, seg_words, off, flags); break;
    case 3: launch_shard_pack<3>(c, d_segments, seg_words, off, flags); break;
    default: launch_shard_pack<4>(c, d_segments, seg_words, off, flags); break;
  }
  c->num_gpu_calls += 2;
  HIPCHK(hipGetLastError());
  // what found no room at level 1: this shard's own to its table, the rest behind the buckets of its destination
  rc = sync_cb(c);
  if (rc) return rc;
  if (c->h_cb[CB_FATAL]) {
    snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost (fatal bits %llu)", (unsigned long long)c->h_cb[CB_FATAL]);
    return KC_ERR_CAPACITY;
  }
  const uint64_t n1 = std::min<uint64_t>(c->h_cb[CB_OVF1], c->bb.ovf1_cap);
  if (n1) {
    rc = ensure_room(c, n1);
    if (rc) return rc;
    switch (c->nl) {
      case 1: launch_shard_route<1>(c, n1, d_segments, seg_words, wtotals, loose, flags); break;
      case 2: launch_shard_route<2>(c, n1, d_segments, seg_words, wtotals, loose, flags); break;
      case 3: launch_shard_route<3>(c, n1, d_segments, seg_words, wtotals, loose, flags); break;
      default: launch_shard_route<4>(c, n1, d_segments, seg_words, wtotals, loose, flags); break;
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->d_cb + CB_OVF1, 0, 8, c->stream));
  }
  {
    const size_t cells = (size_t)c->gm.G * c->gm.P1;
    hipLaunchKernelGGL(kc_shard_release_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, c->stream, c->gm, c->bb, me, n, totals, flags,
                       c->d_ctrs);
    c->num_gpu_calls++;
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(c->sh.h_plan, totals, (size_t)4 * SHARD_MAX * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint64_t *h_totals = c->sh.h_plan, *h_flags = h_totals + SHARD_MAX, *h_loose = h_flags + SHARD_MAX, *h_wtotals = h_loose + SHARD_MAX;
  uint64_t *h_hdr = c->sh.h_plan + 4 * SHARD_MAX;
  bool patched = false;
  for (uint32_t d = 0; d < n; d++) {
    if (d == me) continue;
    if (h_flags[d]) {
      snprintf(g_last_error, sizeof(g_last_error), "the segment of shard %u is too small: %llu records (+ %llu loose) do not fit %llu words", d,
               (unsigned long long)h_totals[d], (unsigned long long)h_loose[d], (unsigned long long)seg_words);
      return KC_ERR_CAPACITY;
    }
    const uint32_t nb = shard_first_bucket(d + 1, c->gm.P1, n) - shard_first_bucket(d, c->gm.P1, n);
    h_words[d] = shard_header_words(nb) + h_wtotals[d] + h_loose[d] * (uint64_t)c->nl;  // (five bytes a record in the compact wire form)
    if (h_loose[d]) {
      h_hdr[d] = (uint64_t)nb | (h_loose[d] << 32);
      HIPCHK(hipMemcpyAsync(d_segments + (size_t)d * seg_words + 1, &h_hdr[d], 8, hipMemcpyHostToDevice, c->stream));
    }
    c->sh.sent += h_totals[d] + h_loose[d];
    patched = patched || h_loose[d] != 0;
  }
  // the contract of kc_shard_extract: when it returns, every byte of the segments is in place (the caller may hand
  // them to another stream or to the network at once) -- the header patches above included
  if (patched) HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}

extern "C" int kc_shard_extract(kc_ctx *c, const uint8_t *bases, const uint8_t *quals, const uint64_t *offsets, uint64_t nreads, int on_device,
                                uint64_t *d_segments, uint64_t seg_words, uint64_t *h_words) {
  int rc = shard_extract_begin(c, d_segments, seg_words, h_words);
  if (rc) return rc;
  c->sh.flow = true;
  c->sh.extracting = true;
  rc = submit_reads_impl(c, bases, quals, offsets, nreads, on_device, MODE_INSERT, nullptr, 0);
  c->sh.extracting = false;
  if (rc) return rc;
  return shard_extract_pack(c, nreads != 0, d_segments, seg_words, h_words);
}

extern "C" int kc_shard_extract_seq_block(kc_ctx *c, const char *seqs, uint64_t len, int on_device, uint64_t *d_segments, uint64_t seg_words,
                                          uint64_t *h_words) {
  int rc = shard_extract_begin(c, d_segments, seg_words, h_words);
  if (rc) return rc;
  c->sh.flow = true;
  c->sh.extracting = true;
  rc = kc_submit_seq_block(c, seqs, len, on_device);
  c->sh.extracting = false;
  if (rc) return rc;
  return shard_extract_pack(c, len != 0, d_segments, seg_words, h_words);
}

extern "C" int kc_shard_reserve(kc_ctx *c, uint64_t nwords, uint64_t **d_dst) {
  if (!c || !d_dst) return KC_ERR_INVALID_ARG;
  *d_dst = nullptr;
  if (!nwords) return KC_OK;
  HIPCHK(hipSetDevice(c->cfg.device));
  const uint64_t need = (nwords + 1) & ~1ULL;  // 16-byte granules
  for (auto &e : c->sh_extents) {
    if (e.cap - e.used >= need) {
      *d_dst = e.p + e.used;
      e.used += need;
      return KC_OK;
    }
  }
  // a new extent (the old ones stay where they are: segments already there are being read): an eighth of the buffer or
  // what this call needs
  const uint64_t bcap = c->cfg.max_kmers_buffered ? c->cfg.max_kmers_buffered : (1ULL << 26);
  kc_ctx::ShardExtent e;
  e.cap = std::max<uint64_t>(need, std::max<uint64_t>(bcap * (uint64_t)c->nl / 8, 1ULL << 17));
  e.used = need;
  e.p = nullptr;
  HIPCHK(hipMalloc((void **)&e.p, (size_t)e.cap * 8));
  c->sh_extents.push_back(e);
  *d_dst = e.p;
  return KC_OK;
}

extern "C" int kc_shard_commit(kc_ctx *c, const uint64_t *d_segment, uint64_t nwords) {
  if (!c || (nwords && !d_segment)) return KC_ERR_INVALID_ARG;
  if (!nwords) return KC_OK;
  if (c->finalized || c->bk_level2) return KC_ERR_STATE;
  const uint32_t n = (uint32_t)c->cfg.rank_n, me = (uint32_t)c->cfg.rank_me;
  if (n < 2 || !bk_active(c)) return KC_ERR_STATE;
  if (c->started && !c->sh.flow) return KC_ERR_STATE;
  bool inside = false;
  for (auto &e : c->sh_extents) inside |= d_segment >= e.p && d_segment + nwords <= e.p + e.used;
  if (!inside) {
    snprintf(g_last_error, sizeof(g_last_error), "kc_shard_commit: the segment does not lie in memory handed out by kc_shard_reserve");
    return KC_ERR_INVALID_ARG;
  }
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bk_init(c);
  if (rc) return rc;
  rc = shard_init(c);
  if (rc) return rc;
  if (nwords < SHARD_HDR) return KC_ERR_INVALID_ARG;
  uint64_t *hdr = c->sh.h_plan;
  HIPCHK(hipMemcpyAsync(hdr, d_segment, SHARD_HDR * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipMemcpyAsync(c->h_ctrs, c->d_ctrs, CTR_COUNT * 8, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(hipStreamSynchronize(c->stream));
  const uint32_t lo = shard_first_bucket(me, c->gm.P1, n), nbo = shard_first_bucket(me + 1, c->gm.P1, n) - lo;
  const uint32_t nb = (uint32_t)hdr[1];
  const uint64_t loose = hdr[1] >> 32, nrec = hdr[2], rec_words = hdr[3] & ((1ULL << 56) - 1);
  const uint32_t wire = (uint32_t)(hdr[3] >> 56);
  if (hdr[0] != shard_signature(c) || nb != nbo || wire != shard_wire_of(c->gm) ||
      shard_header_words(nb) + rec_words + loose * (uint64_t)c->nl != nwords) {
    snprintf(g_last_error, sizeof(g_last_error), "kc_shard_commit: not a segment for this shard (made by a context with another k, geometry or shard count?)");
    return KC_ERR_INVALID_ARG;
  }
  {
    // the per-bucket counts must add up to the records the header announces: level 2 takes every bucket's address from
    // them, and a truncated or corrupt segment would otherwise be read past its end at finalize
    std::vector<uint32_t> counts(nb);
    HIPCHK(hipMemcpy(counts.data(), d_segment + SHARD_HDR, (size_t)nb * 4, hipMemcpyDeviceToHost));
    uint64_t sum = 0, wsum = 0;
    for (uint32_t v : counts) {
      sum += v;
      wsum += shard_bucket_words(v, (uint32_t)c->nl, wire);
    }
    if (sum != nrec || wsum != rec_words) {
      snprintf(g_last_error, sizeof(g_last_error),
               "kc_shard_commit: the segment's bucket counts add up to %llu records in %llu words, its header says %llu in %llu",
               (unsigned long long)sum, (unsigned long long)wsum, (unsigned long long)nrec, (unsigned long long)rec_words);
      return KC_ERR_INVALID_ARG;
    }
  }
  if (c->sh.F >= std::min<uint32_t>(FLAT_MAX, GMAX - c->gm.G)) {
    snprintf(g_last_error, sizeof(g_last_error), "shard flow: %u segments received in one pass is the limit: use larger blocks", c->sh.F);
    return KC_ERR_CAPACITY;
  }
  if (c->h_ctrs[CTR_INSERTED] + nrec + loose > c->bk_capacity) {
    snprintf(g_last_error, sizeof(g_last_error), "shard flow: this shard would hold more k-mers than max_kmers_buffered (%llu)", (unsigned long long)c->bk_capacity);
    return KC_ERR_CAPACITY;
  }
  if (c->sh.nbo != nbo || !c->sh.d_cnt) {  // first segment for this geometry
    if (c->sh.d_cnt) (void)hipFree(c->sh.d_cnt);
    if (c->sh.d_at) (void)hipFree(c->sh.d_at);
    c->sh.d_cnt = nullptr;
    c->sh.d_at = nullptr;
    HIPCHK(hipMalloc((void **)&c->sh.d_cnt, (size_t)FLAT_MAX * nbo * 4));
    HIPCHK(hipMalloc((void **)&c->sh.d_at, (size_t)FLAT_MAX * nbo * 8));
    c->sh.nbo = nbo;
  }
  c->sh.flow = true;
  c->started = true;
  const uint32_t f = c->sh.F++;
  hipLaunchKernelGGL(kc_shard_index_kernel, dim3(1), dim3(WGB), 0, c->stream, d_segment, nb, (uint32_t)c->nl, wire, c->sh.d_cnt + (size_t)f * nbo,
                     c->sh.d_at + (size_t)f * nbo, c->d_ctrs);
  c->num_gpu_calls++;
  HIPCHK(hipGetLastError());
  if (loose) {
    rc = ensure_room(c, loose);
    if (rc) return rc;
    const uint64_t *lp = d_segment + shard_header_words(nb) + rec_words;
    switch (c->nl) {
      case 1: launch_shard_loose<1>(c, lp, loose); break;
      case 2: launch_shard_loose<2>(c, lp, loose); break;
      case 3: launch_shard_loose<3>(c, lp, loose); break;
      default: launch_shard_loose<4>(c, lp, loose); break;
    }
    HIPCHK(hipGetLastError());
  }
  c->sh.received += nrec + loose;
  return KC_OK;
}

extern "C" int kc_shard_capacity(kc_ctx *c, uint64_t *max_distinct) {
  if (!c || !max_distinct) return KC_ERR_INVALID_ARG;
  *max_distinct = 0;
  if (!bk_active(c)) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bk_init(c);
  if (rc) return rc;
  const uint32_t n = (uint32_t)c->cfg.rank_n, me = (uint32_t)c->cfg.rank_me;
  const uint64_t mine = shard_first_bucket(me + 1, c->gm.P1, n) - shard_first_bucket(me, c->gm.P1, n);
  *max_distinct = (uint64_t)(KC_MAX_REGION_LOAD * (double)(mine * c->gm.P2) * (double)c->gm.S);  // the load bk_init accepts
  return KC_OK;
}

extern "C" int kc_shard_owner(kc_ctx *c, const uint64_t *kmer, int *owner) {
  if (!c || !kmer || !owner) return KC_ERR_INVALID_ARG;
  if (!bk_active(c)) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = bk_init(c);  // the geometry decides
  if (rc) return rc;
  uint64_t a[KC_MAX_LONGS] = {0, 0, 0, 0};
  for (int j = 0; j < c->nl_ext; j++) a[j] = kmer[j];
  uint32_t b1;
  if (c->gm.cp) {
    const uint64_t mix = kc_feistel_fwd(a[0] >> (64u - c->gm.k2), c->k);
    b1 = (uint32_t)(mix >> (c->gm.k2 - c->gm.la));
  } else {
    uint64_t h;
    switch (c->nl) {
      case 1: { uint64_t x[1] = {a[0]}; h = kc_hash<1>(x); break; }
      case 2: { uint64_t x[2] = {a[0], a[1]}; h = kc_hash<2>(x); break; }
      case 3: { uint64_t x[3] = {a[0], a[1], a[2]}; h = kc_hash<3>(x); break; }
      default: h = kc_hash<4>(a); break;
    }
    b1 = (uint32_t)((((uint32_t)h & 0xFFFFu) * c->gm.P1) >> 16);
  }
  *owner = (int)shard_of_bucket(b1, c->gm.P1, (uint32_t)c->cfg.rank_n);
  return KC_OK;
}

extern "C" int kc_flush(kc_ctx *c) {
  if (!c) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = sync_ctrs(c);
  if (rc) return rc;
  if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
  return KC_OK;
}

// ---- finalize ----------------------------------------------------------------------------------
template <int NL>
static void launch_finalize(kc_ctx *c) {
  unsigned nblk = (unsigned)std::min<uint64_t>((c->capacity + TPB - 1) / TPB, 256 * 16);
  hipLaunchKernelGGL(kc_finalize_kernel<NL>, dim3(nblk), dim3(TPB), 0, c->stream, c->table, c->capacity, c->cfg.dmin_thres,
                     c->d_out_keys, c->d_out_counts, c->d_out_left, c->d_out_right, c->d_ctrs);
}

static int alloc_results(kc_ctx *c, uint64_t cap) {
  if (!cap) cap = 1;
  if (c->d_out_keys && c->out_cap >= cap) {  // the arrays of an earlier run are big enough: keep them
    c->out_n = 0;
    free_index(c);
    return KC_OK;
  }
  free_results(c);
  HIPCHK(hipMalloc((void **)&c->d_out_keys, cap * c->nl * 8));
  HIPCHK(hipMalloc((void **)&c->d_out_counts, cap * 2));
  HIPCHK(hipMalloc((void **)&c->d_out_left, cap));
  HIPCHK(hipMalloc((void **)&c->d_out_right, cap));
  c->out_cap = cap;
  return KC_OK;
}

// make room for `need` results, keeping the first `keep` already written
static int grow_results(kc_ctx *c, uint64_t need, uint64_t keep) {
  if (need <= c->out_cap) return KC_OK;
  uint64_t *k0 = c->d_out_keys;
  uint16_t *c0 = c->d_out_counts;
  uint8_t *l0 = c->d_out_left, *r0 = c->d_out_right;
  c->d_out_keys = nullptr;
  c->d_out_counts = nullptr;
  c->d_out_left = c->d_out_right = nullptr;
  int rc = alloc_results(c, need);
  if (!rc && keep) {
    HIPCHK(hipMemcpyAsync(c->d_out_keys, k0, keep * c->nl * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_out_counts, c0, keep * 2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_out_left, l0, keep, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_out_right, r0, keep, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
  }
  if (k0) (void)hipFree(k0);
  if (c0) (void)hipFree(c0);
  if (l0) (void)hipFree(l0);
  if (r0) (void)hipFree(r0);
  return rc;
}

// S7/S8 over the global table, appended to the result arrays at ctrs[CTR_OUT]
static int table_finalize_append(kc_ctx *c) {
  int rc = sync_ctrs(c);
  if (rc) return rc;
  const uint64_t entries = c->h_ctrs[CTR_ENTRIES];
  if (!entries) return KC_OK;
  rc = grow_results(c, c->h_ctrs[CTR_OUT] + entries, c->h_ctrs[CTR_OUT]);
  if (rc) return rc;
  {
    KernelTimer kt(c, KT_FINALIZE);
    switch (c->nl) {
      case 1: launch_finalize<1>(c); break;
      case 2: launch_finalize<2>(c); break;
      case 3: launch_finalize<3>(c); break;
      default: launch_finalize<4>(c); break;
    }
  }
  HIPCHK(hipGetLastError());
  return KC_OK;
}

// ---- bucketed path: regions, counting, flagged regions ------------------------------------------
// The kernels of level 2, no host wait.  inc: an instalment (kc_l2_split_kernel<..., INC>) -- the first one fixes the
// buckets' parts of the level-2 arena (the buffer's capacity over the fan-out each) and clears the state the
// instalments carry; inc with c->inc_on already set: the next one.  !inc && c->inc_on never happens (bk_level2_t).
template <int NL>
static int bk_level2_launch(kc_ctx *c, bool inc) {
  // short register form: compact records whose mix fits 32 bits below the level-1 bucket
  const bool cr = use_cp<NL>(c) && c->gm.k2 - c->gm.la <= 32;
  const bool fl = shard_flow_only(c);  // only this shard's buckets, their flat sources behind their chains
  if (inc && fl) return KC_ERR_STATE;
  auto kern = inc ? (use_cp<NL>(c) ? (cr ? kc_l2_split_kernel<NL, NL == 1, NL == 1, false, true> : kc_l2_split_kernel<NL, NL == 1, false, false, true>)
                                   : kc_l2_split_kernel<NL, false, false, false, true>)
              : use_cp<NL>(c) ? (cr ? (fl ? kc_l2_split_kernel<NL, NL == 1, NL == 1, true> : kc_l2_split_kernel<NL, NL == 1, NL == 1, false>)
                                    : (fl ? kc_l2_split_kernel<NL, NL == 1, false, true> : kc_l2_split_kernel<NL, NL == 1, false, false>))
                              : (fl ? kc_l2_split_kernel<NL, false, false, true> : kc_l2_split_kernel<NL, false, false, false>);
  // six-byte level-1 records have a level 2 of their own (kc_l2_rec6_kernel)
  const bool r6 = NL == 1 && c->gm.rec6 != 0;
  auto kern6 = inc ? kc_l2_rec6_kernel<false, true> : (fl ? kc_l2_rec6_kernel<true, false> : kc_l2_rec6_kernel<false, false>);
  int rc = r6 ? set_dyn_lds(kern6, l2r6_lds_bytes()) : set_dyn_lds(kern, lds_l2<NL>());
  if (rc) return rc;
  FlatSrc fs;
  memset(&fs, 0, sizeof(fs));
  fs.b_hi = c->gm.P1;
  if (fl) {
    fs.cnt = c->sh.d_cnt;
    fs.at = c->sh.d_at;
    fs.F = c->sh.F;
    fs.b_lo = shard_first_bucket((uint32_t)c->cfg.rank_me, c->gm.P1, (uint32_t)c->cfg.rank_n);
    fs.b_hi = shard_first_bucket((uint32_t)c->cfg.rank_me + 1, c->gm.P1, (uint32_t)c->cfg.rank_n);
    fs.nbo = fs.b_hi - fs.b_lo;
  }
  if (!inc || !c->inc_on) {
    if (inc && !c->l2_per_bucket) c->l2_per_bucket = (c->bk_capacity + c->gm.P1 - 1) / c->gm.P1;
    const uint64_t per_bucket = inc ? c->l2_per_bucket : 0;
    hipLaunchKernelGGL(kc_bucket_prefix_kernel, dim3(1), dim3(WGB), 0, c->stream, c->gm, c->bb, fs, c->d_cb, per_bucket);
    c->num_gpu_calls++;
    if (inc) {
      HIPCHK(hipMemsetAsync(c->bb.cnt2, 0, (size_t)c->gm.P1 * c->gm.P2 * 4, c->stream));
      HIPCHK(hipMemsetAsync(c->bb.done1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
      HIPCHK(hipMemsetAsync(c->bb.used2, 0, (size_t)c->gm.P1 * 4, c->stream));
      c->inc_on = true;
    }
  }
#ifdef KC_ABLATE
  c->gm.abl = getenv("KC_ABL_L2") ? (uint32_t)atoi(getenv("KC_ABL_L2")) : 0u;
#endif
  if (fs.b_hi > fs.b_lo) {
    KernelTimer kt(c, r6 ? KT_L2_REC6 : KT_L2_SPLIT);
    const dim3 grid(std::min<unsigned>(fs.b_hi - fs.b_lo, (unsigned)c->num_cus));
    if (r6) hipLaunchKernelGGL(kern6, grid, dim3(WGB), l2r6_lds_bytes(), c->stream, c->gm, c->bb, fs, c->d_cb);
    else hipLaunchKernelGGL(kern, grid, dim3(WGB), lds_l2<NL>(), c->stream, c->gm, c->bb, fs, c->d_cb);
  }
  HIPCHK(hipGetLastError());
  return KC_OK;
}

// an instalment of level 2 over what has been buffered since the last one (the host pipe, between two blocks)
static int bk_level2_instalment(kc_ctx *c) {
  if (!bk_active(c) || !c->bk_ready || c->bk_level2 || c->sh.flow) return KC_OK;
  switch (c->nl) {
    case 1: return bk_level2_launch<1>(c, true);
    case 2: return bk_level2_launch<2>(c, true);
    case 3: return bk_level2_launch<3>(c, true);
    default: return bk_level2_launch<4>(c, true);
  }
}

// Room in the level-2 arena for `need` compact records in all while level 2 runs in instalments: every bucket's part is
// fixed up front (kc_bucket_prefix_kernel with per_bucket), so more records than planned mean a larger arena and, once
// instalments have begun, the parts moved into it (kc_l2_grow_kernel).  KC_ERR_OUT_OF_MEMORY when the device has no room.
static int bk_l2_reserve(kc_ctx *c, uint64_t need) {
  const Geom &g = c->gm;
  const uint64_t CH2 = 1ULL << g.log2CH2;
  const uint64_t pb_need = (uint64_t)((double)need * 1.08 / (double)g.P1) + 2 * CH2;  // (buckets differ by a few per cent)
  const uint64_t pb_now = c->inc_on ? c->l2_per_bucket : (c->bk_capacity + g.P1 - 1) / g.P1;
  if (pb_need <= pb_now) {
    if (!c->inc_on) c->l2_per_bucket = pb_now;
    return KC_OK;
  }
  const uint64_t pb = c->inc_on ? std::max<uint64_t>(2 * pb_now, pb_need) : pb_need;
  const uint64_t a2 = (uint64_t)g.P1 * ((pb + CH2 - 1) / CH2 + g.P2) + 16;
  if (a2 >= (1ULL << 32)) return KC_ERR_OUT_OF_MEMORY;
  uint64_t *bigger = nullptr;
  const size_t bytes = (size_t)a2 * CH2 * 4;
  size_t free_b = 0, total_b = 0;
  if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + ((size_t)4 << 30) || hipMalloc((void **)&bigger, bytes) != hipSuccess) {
    (void)hipGetLastError();
    return KC_ERR_OUT_OF_MEMORY;
  }
  if (c->inc_on) {
    uint32_t *nb = nullptr;
    if (hipMalloc((void **)&nb, ((size_t)g.P1 + 1) * 4) != hipSuccess) {
      (void)hipGetLastError();
      (void)hipFree(bigger);
      return KC_ERR_OUT_OF_MEMORY;
    }
    BucketBufs tmp = c->bb;
    tmp.base2 = nb;
    FlatSrc fs;
    memset(&fs, 0, sizeof(fs));
    hipLaunchKernelGGL(kc_bucket_prefix_kernel, dim3(1), dim3(WGB), 0, c->stream, c->gm, tmp, fs, c->d_cb, pb);
    hipLaunchKernelGGL(kc_l2_grow_kernel, dim3(g.P1), dim3(WGB), 0, c->stream, c->gm, reinterpret_cast<const uint32_t *>(c->bb.rec2),
                       reinterpret_cast<uint32_t *>(bigger), c->bb.base2, nb, c->bb.used2, c->bb.chain2, c->bb.cnt2);
    c->num_gpu_calls += 2;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(c->bb.base2, nb, ((size_t)g.P1 + 1) * 4, hipMemcpyDeviceToDevice, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(nb);
    if (e != hipSuccess) {
      (void)hipFree(bigger);
      return hip_fail(e, "bk_l2_reserve", __LINE__);
    }
  }
  HIPCHK(hipFree(c->bb.rec2));
  c->bk_bytes += bytes - c->bk_held[4];
  c->bb.rec2 = bigger;
  c->bk_held[4] = bytes;
  c->gm.A2 = (uint32_t)a2;
  c->l2_per_bucket = pb;
  return KC_OK;
}

// An instalment of level 2 behind which level 1 is emptied (bk_light_spill), and the last pass of a context that has done
// that, cannot be answered by "run the whole pass again" when the region overflow list fills up: level 1 no longer holds
// every record.  So level 2's state -- the regions' lengths, the buckets' arena marks, how far every level-1 chain has been
// read, the list's length -- is saved before such a launch (a few MB), and when the launch reports a full list the state
// is put back, the list grown to what the launch asked for, and the same launch repeated.
static int l2_snapshot(kc_ctx *c) {
  const Geom &g = c->gm;
  const size_t b_cnt2 = (size_t)g.P1 * g.P2 * 4, b_used2 = (size_t)g.P1 * 4, b_done1 = (size_t)g.G * g.P1 * 4, need = b_cnt2 + b_used2 + b_done1 + 8;
  c->l2snap_fresh = !c->inc_on;
  if (c->l2snap_fresh) return KC_OK;  // nothing to save: the launch itself starts level 2 from zero
  if (c->l2snap_bytes < need) {
    if (c->d_l2snap) HIPCHK(hipFree(c->d_l2snap));
    c->d_l2snap = nullptr;
    c->l2snap_bytes = 0;
    HIPCHK(hipMalloc((void **)&c->d_l2snap, need));
    c->l2snap_bytes = need;
  }
  HIPCHK(hipMemcpyAsync(c->d_l2snap, c->bb.cnt2, b_cnt2, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_l2snap + b_cnt2, c->bb.used2, b_used2, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_l2snap + b_cnt2 + b_used2, c->bb.done1, b_done1, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(hipMemcpyAsync(c->d_l2snap + b_cnt2 + b_used2 + b_done1, c->d_cb + CB_OVF2, 8, hipMemcpyDeviceToDevice, c->stream));
  return KC_OK;
}
// back to the snapshot with a list of at least `need` records (its entries from before the launch kept)
static int l2_restore_and_grow(kc_ctx *c, uint64_t need, uint64_t kept) {
  const Geom &g = c->gm;
  const size_t b_cnt2 = (size_t)g.P1 * g.P2 * 4, b_used2 = (size_t)g.P1 * 4, b_done1 = (size_t)g.G * g.P1 * 4;
  if (need > c->bb.ovf2_cap) {
    uint64_t *bigger = nullptr;
    if (hipMalloc((void **)&bigger, need * (size_t)c->nl * 8) != hipSuccess) {
      (void)hipGetLastError();
      snprintf(g_last_error, sizeof(g_last_error), "no memory for a region overflow list of %llu records", (unsigned long long)need);
      return KC_ERR_OUT_OF_MEMORY;
    }
    if (kept) HIPCHK(hipMemcpyAsync(bigger, c->bb.ovf2, kept * (size_t)c->nl * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(c->bb.ovf2));
    c->bk_bytes += (need - c->bb.ovf2_cap) * (size_t)c->nl * 8;
    c->bb.ovf2 = bigger;
    c->bb.ovf2_cap = need;
    c->bk_held[10] = need * (size_t)c->nl * 8;
  }
  if (c->l2snap_fresh) {
    c->inc_on = false;  // the launch starts level 2 from zero again
    HIPCHK(hipMemsetAsync(c->d_cb + CB_OVF2, 0, 2 * 8, c->stream));  // OVF2, FATAL (nothing was on the list before level 2 began)
  } else {
    HIPCHK(hipMemcpyAsync(c->bb.cnt2, c->d_l2snap, b_cnt2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bb.used2, c->d_l2snap + b_cnt2, b_used2, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->bb.done1, c->d_l2snap + b_cnt2 + b_used2, b_done1, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(c->d_cb + CB_OVF2, c->d_l2snap + b_cnt2 + b_used2 + b_done1, 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipMemsetAsync(c->d_cb + CB_FATAL, 0, 8, c->stream));
  }
  return KC_OK;
}

// The buffer is full and more reads are coming (compact records): level 1's records go through level 2 now, level 1
// starts again empty, level 2 keeps what it has until the regions are counted.  buffered: records level 1 holds (an upper
// bound).  KC_ERR_UNSUPPORTED_K: not a geometry this works for (the caller takes the other way).
static int bk_light_spill(kc_ctx *c, uint64_t buffered) {
  if (!c->gm.cp || c->nl != 1 || !bk_active(c) || !c->bk_ready || c->bk_level2 || c->sh.flow) return KC_ERR_UNSUPPORTED_K;
  const char *e = getenv("KC_LIGHT_SPILL");
  if (e && e[0] == '0') return KC_ERR_UNSUPPORTED_K;  // (A/B runs: the counted buffer merged into the global table, as for longer k-mers)
  int rc = sync_cb(c);
  if (rc) return rc;
  if (c->h_cb[CB_FATAL]) {
    snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost (fatal bits %llu)", (unsigned long long)c->h_cb[CB_FATAL]);
    return KC_ERR_CAPACITY;
  }
  rc = bk_l2_reserve(c, c->l2_held + buffered);
  if (rc) return rc;
  // the region overflow list takes what this instalment's heavy regions spill: at least half of it free, or it grows
  const uint64_t used2 = std::min<uint64_t>(c->h_cb[CB_OVF2], c->bb.ovf2_cap);
  if (used2 * 2 > c->bb.ovf2_cap) {
    const uint64_t cap = 2 * c->bb.ovf2_cap;
    uint64_t *bigger = nullptr;
    if (hipMalloc((void **)&bigger, cap * 8) != hipSuccess) {
      (void)hipGetLastError();
      return KC_ERR_OUT_OF_MEMORY;
    }
    HIPCHK(hipMemcpyAsync(bigger, c->bb.ovf2, used2 * 8, hipMemcpyDeviceToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipFree(c->bb.ovf2));
    c->bk_bytes += (cap - c->bb.ovf2_cap) * 8;
    c->bb.ovf2 = bigger;
    c->bb.ovf2_cap = cap;
    c->bk_held[10] = cap * 8;
  }
  const uint64_t listed = std::min<uint64_t>(c->h_cb[CB_OVF2], c->bb.ovf2_cap);  // on the list before this instalment
  rc = l2_snapshot(c);
  if (rc) return rc;
  for (int attempt = 0;; attempt++) {
    rc = bk_level2_launch<1>(c, true);
    if (rc) return rc;
    rc = sync_cb(c);
    if (rc) return rc;
    const uint64_t fatal = c->h_cb[CB_FATAL];
    if (!fatal) break;
    if (fatal != FATAL_OVF2 || attempt > 0) {
      snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost in an instalment of level 2 (fatal bits %llu)", (unsigned long long)fatal);
      return KC_ERR_CAPACITY;
    }
    // the list was too small for what this instalment's heavy regions spill: the counter kept counting past its end
    rc = l2_restore_and_grow(c, c->h_cb[CB_OVF2] + c->h_cb[CB_OVF2] / 16 + 4096, c->l2snap_fresh ? 0 : listed);
    if (rc) return rc;
  }
  // what found no room at level 1 joins the flagged regions' list (as bk_level2_t does at the end of a pass)
  const uint64_t n1 = std::min<uint64_t>(c->h_cb[CB_OVF1], c->bb.ovf1_cap);
  if (n1) {
    hipLaunchKernelGGL((kc_ovf1_to_regions_kernel<1, true>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, c->stream, c->gm, c->bb, n1, c->d_cb);
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemsetAsync(c->d_cb + CB_OVF1, 0, 8, c->stream));
  }
  // level 1 starts again: every chain empty, every writer's arena whole
  HIPCHK(hipMemsetAsync(c->bb.cnt1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.done1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.used1, 0, (size_t)c->gm.G * 2 * 4, c->stream));
  c->ovf1_ub = 0;
  c->l2_held += buffered;
  c->l1_dropped = true;
  return KC_OK;
}

template <int NL>
static int bk_level2_t(kc_ctx *c) {
  int rc = bk_level2_launch<NL>(c, c->inc_on);  // (after instalments: one more, over the rest)
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  rc = sync_cb(c);
  if (rc) return rc;
#ifdef KC_STAMPS
  fprintf(stderr, "l2 kernel cycles (thread 0, summed over workgroups): hist %llu barrierA %llu scan+reserve %llu scatter %llu copyout %llu take-over+loads %llu\n",
          (unsigned long long)c->h_cb[8], (unsigned long long)c->h_cb[9], (unsigned long long)c->h_cb[10],
          (unsigned long long)c->h_cb[11], (unsigned long long)c->h_cb[12], (unsigned long long)c->h_cb[13]);
  HIPCHK(hipMemsetAsync(c->d_cb + 8, 0, 8 * 8, c->stream));
#endif
  const uint64_t n1 = std::min<uint64_t>(c->h_cb[CB_OVF1], c->bb.ovf1_cap);
  if (n1) {
    auto okern = use_cp<NL>(c) ? kc_ovf1_to_regions_kernel<NL, NL == 1> : kc_ovf1_to_regions_kernel<NL, false>;
    hipLaunchKernelGGL(okern, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, c->stream, c->gm, c->bb, n1, c->d_cb);
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
  }
  return KC_OK;
}

// build the regions from everything buffered (once per reset)
static int bk_build_regions(kc_ctx *c) {
  if (c->bk_level2) return KC_OK;
  int rc = sync_cb(c);
  if (rc) return rc;
  // (an instalment of level 2 may have filled the second overflow list already: the loop below deals with that)
  if (c->h_cb[CB_FATAL] & ~(c->inc_on ? (uint64_t)FATAL_OVF2 : 0ULL)) {
    snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost at level 1 (fatal bits %llu)", (unsigned long long)c->h_cb[CB_FATAL]);
    return KC_ERR_CAPACITY;
  }
#ifdef KC_STAMPS
  (void)sync_cb(c);
  fprintf(stderr, "l1 kernel cycles (thread 0, summed over workgroups): extract+hist %llu barrierA %llu scan+reserve %llu scatter %llu copyout %llu stage %llu\n",
          (unsigned long long)c->h_cb[8], (unsigned long long)c->h_cb[9], (unsigned long long)c->h_cb[10],
          (unsigned long long)c->h_cb[11], (unsigned long long)c->h_cb[12], (unsigned long long)c->h_cb[13]);
  (void)sync_ctrs(c);
  fprintf(stderr, "   encode: first barrier %llu groups %llu offsets %llu last barrier %llu\n", (unsigned long long)c->h_ctrs[CTR_BIN0 + 40],
          (unsigned long long)c->h_ctrs[CTR_BIN0 + 41], (unsigned long long)c->h_ctrs[CTR_BIN0 + 42], (unsigned long long)c->h_ctrs[CTR_BIN0 + 43]);
  HIPCHK(hipMemsetAsync(c->d_cb + 8, 0, 8 * 8, c->stream));
#endif
  uint64_t listed = 0;
  if (c->l1_dropped) {  // (the last pass of a context whose level 1 has been emptied behind earlier instalments)
    listed = std::min<uint64_t>(c->h_cb[CB_OVF2], c->bb.ovf2_cap);
    rc = l2_snapshot(c);
    if (rc) return rc;
  }
  for (int attempt = 0;; attempt++) {
    switch (c->nl) {
      case 1: rc = bk_level2_t<1>(c); break;
      case 2: rc = bk_level2_t<2>(c); break;
      case 3: rc = bk_level2_t<3>(c); break;
      default: rc = bk_level2_t<4>(c); break;
    }
    if (rc) return rc;
    rc = sync_cb(c);
    if (rc) return rc;
    const uint64_t fatal = c->h_cb[CB_FATAL];
    if (!fatal) break;
    if (fatal != FATAL_OVF2 || attempt > 0) {
      snprintf(g_last_error, sizeof(g_last_error), "k-mer buffer: records were lost building the regions (fatal bits %llu)", (unsigned long long)fatal);
      return KC_ERR_CAPACITY;
    }
    if (c->l1_dropped) {  // level 1 no longer holds every record: back to where this pass began, with a larger list
      rc = l2_restore_and_grow(c, c->h_cb[CB_OVF2] + c->h_cb[CB_OVF2] / 16 + 4096, c->l2snap_fresh ? 0 : listed);
      if (rc) return rc;
      continue;
    }
    // The second overflow list was too small for the regions that outgrew their chains (heavy hitters).  Level 1 is
    // untouched and the counter kept counting past the end, so it says exactly how much room the same pass needs.
    // (After instalments the pass is run whole: the level-1 chains still hold every record.)
    c->inc_on = false;
  c->l2_per_bucket = 0;
    const uint64_t need = c->h_cb[CB_OVF2] + c->h_cb[CB_OVF2] / 64 + 4096;
    uint64_t *bigger = nullptr;
    HIPCHK(hipMalloc((void **)&bigger, need * (size_t)c->nl * 8));
    HIPCHK(hipFree(c->bb.ovf2));
    c->bk_bytes += (need - c->bb.ovf2_cap) * (size_t)c->nl * 8;
    c->bb.ovf2 = bigger;
    c->bb.ovf2_cap = need;
    c->bk_held[10] = need * (size_t)c->nl * 8;
    HIPCHK(hipMemsetAsync(c->d_cb + CB_OVF2, 0, 2 * 8, c->stream));  // OVF2, FATAL
  }
  c->bk_level2 = true;
  return KC_OK;
}

template <int NL, bool DUMP>
static int bk_count_t(kc_ctx *c, const OutBufs &ob) {
  auto kern = use_cp<NL>(c) ? kc_count_kernel<NL, DUMP, NL == 1> : kc_count_kernel<NL, DUMP, false>;
  const size_t lds = CountLDS<NL>::bytes(c->gm.S, use_cp<NL>(c));
  int rc = set_dyn_lds(kern, lds);
  if (rc) return rc;
  const uint64_t R = (uint64_t)c->gm.P1 * c->gm.P2;
  // as many workgroups per CU as the LDS admits (at most 2: 1024 threads each), so that one region's barriers
  // and scans overlap another's inserts
  const unsigned per_cu = lds * 2 <= 160 * 1024 ? 2u : 1u;
#ifdef KC_ABLATE
  c->gm.abl = getenv("KC_ABL_COUNT") ? (uint32_t)atoi(getenv("KC_ABL_COUNT")) : 0u;
#endif
  KernelTimer kt(c, KT_COUNT_REGIONS);
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>(R, (uint64_t)c->num_cus * per_cu)), dim3(WGB), lds, c->stream, c->gm, c->bb,
                     ob, c->cfg.dmin_thres, c->d_ctrs, c->d_cb);
  return KC_OK;
}

static int bk_count(kc_ctx *c, const OutBufs &ob, bool dump) {
  int rc;
  if (dump) {
    switch (c->nl) {
      case 1: rc = bk_count_t<1, true>(c, ob); break;
      case 2: rc = bk_count_t<2, true>(c, ob); break;
      case 3: rc = bk_count_t<3, true>(c, ob); break;
      default: rc = bk_count_t<4, true>(c, ob); break;
    }
  } else {
    switch (c->nl) {
      case 1: rc = bk_count_t<1, false>(c, ob); break;
      case 2: rc = bk_count_t<2, false>(c, ob); break;
      case 3: rc = bk_count_t<3, false>(c, ob); break;
      default: rc = bk_count_t<4, false>(c, ob); break;
    }
  }
  if (rc) return rc;
  HIPCHK(hipGetLastError());
  return KC_OK;
}

template <int NL>
static void launch_flagged_to_table(kc_ctx *c) {
  const uint64_t R = (uint64_t)c->gm.P1 * c->gm.P2;
  KernelTimer kt(c, KT_FALLBACK);
  auto kern = use_cp<NL>(c) ? kc_flagged_to_table_kernel<NL, NL == 1> : kc_flagged_to_table_kernel<NL, false>;
  hipLaunchKernelGGL(kern, dim3((unsigned)std::min<uint64_t>(R, 65536)), dim3(TPB), 0, c->stream, c->gm, c->bb,
                     c->table, c->d_ctrs);
}

// after a count pass: regions that did not fit, and the overflow records, go to the global table (once)
static int bk_move_flagged(kc_ctx *c) {
  if (c->bk_flagged) return KC_OK;
  const uint64_t R = (uint64_t)c->gm.P1 * c->gm.P2;
  HIPCHK(hipMemsetAsync(c->d_cb + CB_FLAGGED_RECS, 0, 8, c->stream));
  hipLaunchKernelGGL(kc_sum_flagged_kernel, dim3((unsigned)std::min<uint64_t>((R + 255) / 256, 1024)), dim3(256), 0, c->stream, c->gm, c->bb,
                     c->d_cb);
  c->num_gpu_calls++;
  int rc = sync_cb(c);
  if (rc) return rc;
  if (c->h_cb[CB_FATAL]) {
    snprintf(g_last_error, sizeof(g_last_error), "region overflow list exhausted: raise max_kmers_buffered");
    return KC_ERR_CAPACITY;
  }
  const uint64_t nflag = c->h_cb[CB_FLAGGED_RECS], n2 = std::min<uint64_t>(c->h_cb[CB_OVF2], c->bb.ovf2_cap);
  if (nflag) {
    rc = ensure_room(c, nflag + n2);
    if (rc) return rc;
    switch (c->nl) {
      case 1: launch_flagged_to_table<1>(c); break;
      case 2: launch_flagged_to_table<2>(c); break;
      case 3: launch_flagged_to_table<3>(c); break;
      default: launch_flagged_to_table<4>(c); break;
    }
    HIPCHK(hipGetLastError());
  }
  if (n2) {
    rc = table_insert_records(c, c->bb.ovf2, n2, 0u);
    if (rc) return rc;
  }
  c->bk_flagged = true;
  return KC_OK;
}

static int bk_finalize(kc_ctx *c) {
  int rc = bk_build_regions(c);
  if (rc) return rc;
  rc = sync_ctrs(c);
  if (rc) return rc;
  // survivors have count >= 2, so at most half the buffered occurrences; usually far fewer
  uint64_t cap = std::max<uint64_t>(1u << 16, std::min<uint64_t>(c->h_ctrs[CTR_INSERTED] / 2 + 1,
                                                                   c->cfg.max_elems ? c->cfg.max_elems / 2 : c->h_ctrs[CTR_INSERTED] / 8 + 1));
  // Output positions are handed out in blocks per workgroup (OutBufs::block); the unused tails are closed afterwards.
  // The arrays need room for one block per workgroup beyond the entries themselves.
  const uint32_t max_wg = 2u * (uint32_t)c->num_cus;
  // blocks of up to 8192 entries, smaller when few results are expected (all the tails together stay within a
  // quarter of the estimate)
  uint32_t block = 0;
  if (2 * max_wg <= PLAN_RUNS) {  // (a workgroup may leave two holes: its current block's tail and an unused spare)
    block = 64;
    while (block < 8192u && (uint64_t)block * 2 * max_wg * 4 <= cap) block *= 2;
  }
  if (block && !c->d_out_plan) {
    HIPCHK(hipMalloc((void **)&c->d_out_plan, (PLAN_WORDS + 4 * (size_t)max_wg) * 8));
  }
  // a workgroup holds at most two blocks that are not full, each a multiple of `block` that covers a region table
  const uint64_t slack = block ? 2 * (((uint64_t)c->gm.S + block - 1) / block * block) * max_wg : 0;
  cap += slack;
  for (int attempt = 0; attempt < 3; attempt++) {
    rc = alloc_results(c, cap);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->d_ctrs + CTR_OUT, 0, 3 * 8, c->stream));  // OUT, PURGED, SUM_COUNTS
    HIPCHK(hipMemsetAsync(c->d_cb + CB_ENTRIES, 0, 2 * 8, c->stream));  // ENTRIES, OUT_RESERVED
    OutBufs ob;
    ob.keys = c->d_out_keys;
    ob.counts = c->d_out_counts;
    ob.left = c->d_out_left;
    ob.right = c->d_out_right;
    ob.exts = nullptr;
    ob.cap = c->out_cap;
    ob.cursor = c->d_ctrs + CTR_OUT;
    ob.block = block;
    ob.tails = block ? c->d_out_plan + PLAN_WORDS : nullptr;
    if (block) HIPCHK(hipMemsetAsync(ob.tails, 0, 4 * (size_t)max_wg * 8, c->stream));
    rc = bk_count(c, ob, false);
    if (rc) return rc;
    if (block) {
      hipLaunchKernelGGL(kc_out_plan_kernel, dim3(1), dim3(WGB), 0, c->stream, ob.tails, 2 * max_wg, ob.cursor, c->d_cb + CB_OUT_RESERVED,
                         c->d_out_plan);
      const unsigned mgrid = (unsigned)std::min<uint64_t>((slack + 255) / 256, 4096);
      switch (c->nl) {
        case 1: hipLaunchKernelGGL(kc_out_move_kernel<1>, dim3(mgrid), dim3(256), 0, c->stream, c->d_out_plan, ob); break;
        case 2: hipLaunchKernelGGL(kc_out_move_kernel<2>, dim3(mgrid), dim3(256), 0, c->stream, c->d_out_plan, ob); break;
        case 3: hipLaunchKernelGGL(kc_out_move_kernel<3>, dim3(mgrid), dim3(256), 0, c->stream, c->d_out_plan, ob); break;
        default: hipLaunchKernelGGL(kc_out_move_kernel<4>, dim3(mgrid), dim3(256), 0, c->stream, c->d_out_plan, ob); break;
      }
      c->num_gpu_calls += 2;
      HIPCHK(hipGetLastError());
    }
    rc = sync_ctrs(c);
    if (rc) return rc;
#ifdef KC_STAMPS
    (void)sync_cb(c);
    fprintf(stderr, "count kernel cycles (thread 0, summed over workgroups): header %llu insert %llu table pass %llu\n",
            (unsigned long long)c->h_cb[8], (unsigned long long)c->h_cb[9], (unsigned long long)c->h_cb[10]);
#endif
    if (block) {
      rc = sync_cb(c);
      if (rc) return rc;
      // every block ever taken must lie inside the arrays, or entries were dropped: run again with enough room
      if (c->h_cb[CB_OUT_RESERVED] <= c->out_cap) break;
      cap = c->h_ctrs[CTR_OUT] + slack;
      if (attempt == 2) {
        snprintf(g_last_error, sizeof(g_last_error), "result arrays still too small after three passes (%llu entries)", (unsigned long long)cap);
        return KC_ERR_CAPACITY;
      }
      continue;
    }
    if (c->h_ctrs[CTR_OUT] <= c->out_cap) break;
    cap = c->h_ctrs[CTR_OUT];  // the pass only counted past the end: run it again with exactly enough room
    if (attempt == 2) {
      snprintf(g_last_error, sizeof(g_last_error), "result arrays still too small after three passes (%llu entries)", (unsigned long long)cap);
      return KC_ERR_CAPACITY;
    }
  }
  rc = bk_move_flagged(c);
  if (rc) return rc;
  return table_finalize_append(c);
}

static int ctg_merge(kc_ctx *c);

extern "C" int kc_finalize(kc_ctx *c, kc_result *out) {
  if (!c) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  if (!c->finalized) {
    int rc = sync_ctrs(c);
    if (rc) return rc;
    if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
    if (c->bk_ready && !c->table_mode && c->bk_spilled) {  // k-mers may sit both in the table and in the buffer: merge, then the table's finalize
      rc = bk_spill_pass(c);
      if (rc) return rc;
      rc = sync_ctrs(c);
      if (rc) return rc;
    }
    if (c->bk_ready && !c->table_mode && !c->bk_spilled) {
      rc = bk_finalize(c);
      if (rc) return rc;
    } else {
      rc = alloc_results(c, c->h_ctrs[CTR_ENTRIES]);
      if (rc) return rc;
      HIPCHK(hipMemsetAsync(c->d_ctrs + CTR_OUT, 0, 3 * 8, c->stream));  // OUT, PURGED, SUM_COUNTS
      rc = table_finalize_append(c);
      if (rc) return rc;
    }
    rc = sync_ctrs(c);
    if (rc) return rc;
    rc = sync_cb(c);
    if (rc) return rc;
    c->out_n = c->h_ctrs[CTR_OUT];
    c->purged = c->h_ctrs[CTR_PURGED];
    c->sum_counts = c->h_ctrs[CTR_SUM_COUNTS];
    c->unique_at_finalize = c->h_ctrs[CTR_ENTRIES] + c->h_cb[CB_ENTRIES];
    if (c->ctg_table.keys) {  // the contig pass: its k-mers join the reads' results (kc_ctg.hpp)
      rc = ctg_merge(c);
      if (rc) return rc;
    }
    if (c->nl != c->nl_ext) {  // k % 32 in {30, 31}: the caller sees the reference's width
      if (c->d_out_keys_ext) HIPCHK(hipFree(c->d_out_keys_ext));
      c->d_out_keys_ext = nullptr;
      HIPCHK(hipMalloc((void **)&c->d_out_keys_ext, std::max<uint64_t>(c->out_n, 1) * c->nl_ext * 8));
      if (c->out_n) {
        hipLaunchKernelGGL(kc_rewidth_keys_kernel, dim3((unsigned)((c->out_n + 255) / 256)), dim3(256), 0, c->stream, c->d_out_keys,
                           c->d_out_keys_ext, c->out_n, c->nl, c->nl_ext);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(c->stream));
      }
    }
    c->finalized = true;
  }
  if (out) {
    out->n = c->out_n;
    out->num_longs = c->nl_ext;
    out->reserved = 0;
    out->d_keys = c->nl != c->nl_ext ? c->d_out_keys_ext : c->d_out_keys;
    out->d_counts = c->d_out_counts;
    out->d_left = c->d_out_left;
    out->d_right = c->d_out_right;
  }
  return KC_OK;
}

extern "C" int kc_copy_results(kc_ctx *c, uint64_t *keys, uint16_t *counts, uint8_t *left, uint8_t *right) {
  if (!c) return KC_ERR_INVALID_ARG;
  if (!c->finalized) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  const uint64_t n = c->out_n;
  if (!n) return KC_OK;
  if (keys) HIPCHK(hipMemcpy(keys, c->nl != c->nl_ext ? c->d_out_keys_ext : c->d_out_keys, n * c->nl_ext * 8, hipMemcpyDeviceToHost));
  if (counts) HIPCHK(hipMemcpy(counts, c->d_out_counts, n * 2, hipMemcpyDeviceToHost));
  if (left) HIPCHK(hipMemcpy(left, c->d_out_left, n, hipMemcpyDeviceToHost));
  if (right) HIPCHK(hipMemcpy(right, c->d_out_right, n, hipMemcpyDeviceToHost));
  return KC_OK;
}

// count / left / right of every result packed into the 8 bytes of a kc_count_exts (kcount_gpu::CountExts)
__global__ void kc_pack_count_exts_kernel(const uint16_t *counts, const uint8_t *left, const uint8_t *right, uint64_t n, uint64_t *vals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) vals[i] = (uint64_t)counts[i] | ((uint64_t)left[i] << 32) | ((uint64_t)right[i] << 40);
}

extern "C" int kc_copy_results_entries(kc_ctx *c, uint64_t *keys, kc_count_exts *vals) {
  static_assert(sizeof(kc_count_exts) == 8, "kc_count_exts is one 8-byte word");
  if (!c) return KC_ERR_INVALID_ARG;
  if (!c->finalized) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  const uint64_t n = c->out_n;
  if (!n) return KC_OK;
  if (keys) HIPCHK(hipMemcpy(keys, c->nl != c->nl_ext ? c->d_out_keys_ext : c->d_out_keys, n * c->nl_ext * 8, hipMemcpyDeviceToHost));
  if (vals) {
    uint64_t *d_vals = nullptr;
    HIPCHK(hipMalloc((void **)&d_vals, n * 8));
    hipLaunchKernelGGL(kc_pack_count_exts_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, c->d_out_counts, c->d_out_left,
                       c->d_out_right, n, d_vals);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e == hipSuccess) e = hipMemcpy(vals, d_vals, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d_vals);
    if (e != hipSuccess) return hip_fail(e, "kc_copy_results_entries", __LINE__);
  }
  return KC_OK;
}

// entries of the LDS-counted regions, unfiltered: two passes (size, then write)
static int bk_dump(kc_ctx *c, uint64_t **dk, uint16_t **dc, uint16_t **de, uint64_t *n_regions) {
  int rc = bk_build_regions(c);
  if (rc) return rc;
  uint64_t cap = 0;
  *dk = nullptr;
  *dc = nullptr;
  *de = nullptr;
  for (int pass = 0; pass < 2; pass++) {
    HIPCHK(hipMemsetAsync(c->d_cb + CB_DUMP, 0, 8, c->stream));
    OutBufs ob;
    memset(&ob, 0, sizeof(ob));
    ob.keys = *dk;
    ob.counts = *dc;
    ob.exts = *de;
    ob.cap = cap;
    ob.cursor = c->d_cb + CB_DUMP;
    rc = bk_count(c, ob, true);
    if (rc) return rc;
    rc = sync_cb(c);
    if (rc) return rc;
    if (pass == 0) {
      cap = c->h_cb[CB_DUMP];
      if (!cap) break;
      HIPCHK(hipMalloc((void **)dk, cap * c->nl * 8));
      HIPCHK(hipMalloc((void **)dc, cap * 2));
      HIPCHK(hipMalloc((void **)de, cap * 16));
    }
  }
  *n_regions = cap;
  return bk_move_flagged(c);
}

// The buffer is full and more reads are coming: count what it holds (the regions' entries with their raw counters,
// like kc_dump_table), add every counted k-mer to the global table in ONE table operation (not one per occurrence:
// about a seventh of the operations at the benchmark's depth), and start the buffer again empty.  The table is where
// the passes meet; kc_finalize then runs one last pass and the table's own finalize.
template <int NL>
static void launch_merge_entries(kc_ctx *c, const uint64_t *dk, const uint16_t *dc, const uint16_t *de, uint64_t n) {
  KernelTimer kt(c, KT_INSERT_RECORDS);
  hipLaunchKernelGGL(kc_merge_entries_kernel<NL>, dim3((unsigned)std::min<uint64_t>((n + TPB - 1) / TPB, 256 * 32)), dim3(TPB), 0, c->stream, dk,
                     dc, de, n, c->table, c->d_ctrs);
}

static int bk_spill_pass(kc_ctx *c) {
  uint64_t *dk = nullptr;
  uint16_t *dc = nullptr, *de = nullptr;
  uint64_t n = 0;
  int rc = bk_dump(c, &dk, &dc, &de, &n);  // regions built, entries listed, flagged regions and overflow records to the table
  if (!rc && n) {
    rc = ensure_room(c, n);
    if (!rc) {
      switch (c->nl) {
        case 1: launch_merge_entries<1>(c, dk, dc, de, n); break;
        case 2: launch_merge_entries<2>(c, dk, dc, de, n); break;
        case 3: launch_merge_entries<3>(c, dk, dc, de, n); break;
        default: launch_merge_entries<4>(c, dk, dc, de, n); break;
      }
      if (hipGetLastError() != hipSuccess) rc = KC_ERR_HIP;
    }
  }
  if (hipStreamSynchronize(c->stream) != hipSuccess && !rc) rc = KC_ERR_HIP;
  if (dk) (void)hipFree(dk);
  if (dc) (void)hipFree(dc);
  if (de) (void)hipFree(de);
  if (rc) return rc;
  const size_t R = (size_t)c->gm.P1 * c->gm.P2;
  HIPCHK(hipMemsetAsync(c->bb.cnt1, 0, (size_t)c->gm.G * c->gm.P1 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.used1, 0, (size_t)c->gm.G * 2 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.cnt2, 0, R * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->bb.flag, 0, R * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->d_cb, 0, CB_COUNT * 8, c->stream));
  c->bk_level2 = c->bk_flagged = false;
  c->inc_on = false;
  c->l2_per_bucket = 0;
  c->l1_dropped = false;
  c->l2_held = 0;
  c->bk_spilled = true;
  return KC_OK;
}

// n keys from a device array of the library's width to a host array of the reference's width
static int keys_to_host(kc_ctx *c, uint64_t *h_dst, const uint64_t *d_src, uint64_t n) {
  if (!n) return KC_OK;
  if (c->nl == c->nl_ext) {
    HIPCHK(hipMemcpy(h_dst, d_src, n * c->nl * 8, hipMemcpyDeviceToHost));
    return KC_OK;
  }
  uint64_t *tmp = nullptr;
  HIPCHK(hipMalloc((void **)&tmp, n * c->nl_ext * 8));
  hipLaunchKernelGGL(kc_rewidth_keys_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, d_src, tmp, n, c->nl, c->nl_ext);
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess) e = hipMemcpy(h_dst, tmp, n * c->nl_ext * 8, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  if (e != hipSuccess) return hip_fail(e, "keys_to_host", __LINE__);
  return KC_OK;
}

extern "C" int kc_dump_table(kc_ctx *c, uint64_t *keys, uint16_t *counts, uint16_t *exts, uint64_t *n_out) {
  if (!c || !n_out) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = sync_ctrs(c);
  if (rc) return rc;
  if (c->h_ctrs[CTR_BAD_BASE]) return KC_ERR_BAD_BASE;
  uint64_t *rk = nullptr;
  uint16_t *rc16 = nullptr, *re = nullptr;
  uint64_t nreg = 0;
  if (c->bk_ready && !c->table_mode && c->bk_spilled) {  // one k-mer, one entry: what is buffered joins the table first
    rc = bk_spill_pass(c);
    if (rc) return rc;
    rc = sync_ctrs(c);
    if (rc) return rc;
  } else if (c->bk_ready && !c->table_mode) {
    rc = bk_dump(c, &rk, &rc16, &re, &nreg);
    if (rc) return rc;
    rc = sync_ctrs(c);
    if (rc) return rc;
  }
  const uint64_t ntab = c->h_ctrs[CTR_ENTRIES];
  const uint64_t n = nreg + ntab;
  *n_out = n;
  if (!keys || !n) {
    if (rk) (void)hipFree(rk);
    if (rc16) (void)hipFree(rc16);
    if (re) (void)hipFree(re);
    return KC_OK;
  }
  if (nreg) {
    rc = keys_to_host(c, keys, rk, nreg);
    if (rc) return rc;
    HIPCHK(hipMemcpy(counts, rc16, nreg * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(exts, re, nreg * 16, hipMemcpyDeviceToHost));
    (void)hipFree(rk);
    (void)hipFree(rc16);
    (void)hipFree(re);
  }
  if (ntab) {
    uint64_t *dk = nullptr, *dcur = nullptr;
    uint16_t *dc = nullptr, *de = nullptr;
    HIPCHK(hipMalloc((void **)&dk, ntab * c->nl * 8));
    HIPCHK(hipMalloc((void **)&dc, ntab * 2));
    HIPCHK(hipMalloc((void **)&de, ntab * 16));
    HIPCHK(hipMalloc((void **)&dcur, 8));
    HIPCHK(hipMemsetAsync(dcur, 0, 8, c->stream));
    const unsigned nblk = (unsigned)((c->capacity + 255) / 256);
    switch (c->nl) {
      case 1: hipLaunchKernelGGL(kc_dump_kernel<1>, dim3(nblk), dim3(256), 0, c->stream, c->table, c->capacity, dk, dc, de, dcur); break;
      case 2: hipLaunchKernelGGL(kc_dump_kernel<2>, dim3(nblk), dim3(256), 0, c->stream, c->table, c->capacity, dk, dc, de, dcur); break;
      case 3: hipLaunchKernelGGL(kc_dump_kernel<3>, dim3(nblk), dim3(256), 0, c->stream, c->table, c->capacity, dk, dc, de, dcur); break;
      default: hipLaunchKernelGGL(kc_dump_kernel<4>, dim3(nblk), dim3(256), 0, c->stream, c->table, c->capacity, dk, dc, de, dcur); break;
    }
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    rc = keys_to_host(c, keys + nreg * c->nl_ext, dk, ntab);
    if (rc) return rc;
    HIPCHK(hipMemcpy(counts + nreg, dc, ntab * 2, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(exts + nreg * 8, de, ntab * 16, hipMemcpyDeviceToHost));
    (void)hipFree(dk);
    (void)hipFree(dc);
    (void)hipFree(de);
    (void)hipFree(dcur);
  }
  return KC_OK;
}

extern "C" int kc_get_stats(kc_ctx *c, kc_stats *o) {
  if (!c || !o) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  int rc = sync_ctrs(c);
  if (rc) return rc;
  memset(o, 0, sizeof(*o));
  o->num_reads = c->num_reads;
  o->num_bases = c->num_bases;
  o->raw_kmers = c->h_ctrs[CTR_RAW_KMERS];
  o->kmers_inserted = c->h_ctrs[CTR_INSERTED];
  o->num_unique = c->finalized ? c->unique_at_finalize : c->h_ctrs[CTR_ENTRIES];
  o->num_purged = c->purged;
  o->total_kmers = c->out_n;
  o->sum_counts = c->sum_counts;
  o->num_dropped = 0;
  o->capacity = c->capacity;
  o->num_gpu_calls = c->num_gpu_calls;
  o->table_bytes = c->arena_bytes + c->bk_bytes;
  return KC_OK;
}

extern "C" int kc_set_tuning(kc_ctx *c, const kc_tuning *t) {
  if (!c || !t) return KC_ERR_INVALID_ARG;
  if (c->started) return KC_ERR_STATE;
  if (t->mode > 2) return KC_ERR_INVALID_ARG;
  auto pow2_or_zero = [](uint32_t v) { return (v & (v - 1)) == 0; };
  if (!pow2_or_zero(t->chunk1) || !pow2_or_zero(t->chunk2) || t->p1 > PMAX || t->p2 > PMAX || t->writers > GMAX)
    return KC_ERR_INVALID_ARG;
  if (memcmp(&c->tuning, t, sizeof(*t)) == 0) return KC_OK;  // unchanged: keep the arenas
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  c->tuning = *t;
  bk_free(c, true);  // the geometry is chosen again at the first submit
  return KC_OK;
}

template <int NL>
static int ensure_index(kc_ctx *c) {
  if (!c->d_index) {
    if (c->out_n >= 0xFFFFFFFFull) {
      snprintf(g_last_error, sizeof(g_last_error), "the lookup index holds at most 2^32 - 1 results");
      return KC_ERR_CAPACITY;
    }
    uint64_t cap = next_pow2(std::max<uint64_t>(1024, c->out_n * 2));
    HIPCHK(hipMalloc((void **)&c->d_index, cap * 4));
    c->index_cap = cap;
    HIPCHK(hipMemsetAsync(c->d_index, 0, cap * 4, c->stream));
    if (c->out_n) {
      hipLaunchKernelGGL(kc_index_build_kernel<NL>, dim3((unsigned)((c->out_n + 255) / 256)), dim3(256), 0, c->stream, c->d_out_keys,
                         c->out_n, c->d_index, cap - 1);
      c->num_gpu_calls++;
      HIPCHK(hipGetLastError());
    }
  }
  return KC_OK;
}

template <int NL>
static int lookup_t(kc_ctx *c, const uint64_t *dq, uint64_t nq, uint16_t *dc, uint8_t *dl, uint8_t *dr) {
  int rc = ensure_index<NL>(c);
  if (rc) return rc;
  if (nq) {
    hipLaunchKernelGGL(kc_lookup_kernel<NL>, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, c->stream, dq, nq, c->k, c->d_index,
                       c->index_cap - 1, c->d_out_keys, c->d_out_counts, c->d_out_left, c->d_out_right, dc, dl, dr);
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
  }
  return KC_OK;
}

extern "C" int kc_lookup(kc_ctx *c, const uint64_t *queries, uint64_t nq, int on_device, uint16_t *counts, uint8_t *left, uint8_t *right) {
  if (!c || (nq && (!queries || !counts))) return KC_ERR_INVALID_ARG;
  if (!c->finalized) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  const uint64_t *dq = queries;
  uint64_t *sq = nullptr;
  uint16_t *dc = counts;
  uint8_t *dl = left, *dr = right;
  uint64_t *wq = nullptr;  // the queries at the library's width, where that differs from the caller's
  if (!on_device && nq) {
    HIPCHK(hipMalloc((void **)&sq, nq * c->nl_ext * 8 + nq * 4));
    HIPCHK(hipMemcpyAsync(sq, queries, nq * c->nl_ext * 8, hipMemcpyHostToDevice, c->stream));
    dq = sq;
    dc = (uint16_t *)(sq + nq * c->nl_ext);
    dl = (uint8_t *)(dc + nq);
    dr = dl + nq;
  }
  if (c->nl != c->nl_ext && nq) {
    HIPCHK(hipMalloc((void **)&wq, nq * c->nl * 8));
    hipLaunchKernelGGL(kc_rewidth_keys_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, c->stream, dq, wq, nq, c->nl_ext, c->nl);
    HIPCHK(hipGetLastError());
    dq = wq;
  }
  int rc;
  switch (c->nl) {
    case 1: rc = lookup_t<1>(c, dq, nq, dc, dl, dr); break;
    case 2: rc = lookup_t<2>(c, dq, nq, dc, dl, dr); break;
    case 3: rc = lookup_t<3>(c, dq, nq, dc, dl, dr); break;
    default: rc = lookup_t<4>(c, dq, nq, dc, dl, dr); break;
  }
  if (!rc && !on_device && nq) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpy(counts, dc, nq * 2, hipMemcpyDeviceToHost));
    if (left) HIPCHK(hipMemcpy(left, dl, nq, hipMemcpyDeviceToHost));
    if (right) HIPCHK(hipMemcpy(right, dr, nq, hipMemcpyDeviceToHost));
  }
  if (wq) {
    (void)hipStreamSynchronize(c->stream);
    (void)hipFree(wq);
  }
  if (sq) (void)hipFree(sq);
  return rc;
}

// ---- the contig pass (kc_ctg.hpp) ----------------------------------------------------------------------------------
extern "C" int kc_begin_ctg_kmers(kc_ctx *c, uint64_t max_ctg_kmers) {
  if (!c) return KC_ERR_INVALID_ARG;
  if (c->finalized) return KC_ERR_STATE;
  HIPCHK(hipSetDevice(c->cfg.device));
  free_ctg(c);
  const uint64_t cap = next_pow2(std::max<uint64_t>(1024, 2 * max_ctg_kmers));
  HIPCHK(hipMalloc((void **)&c->ctg_table.keys, cap * (size_t)c->nl * 8));
  HIPCHK(hipMalloc((void **)&c->ctg_table.vals, cap * 2 * 4));
  HIPCHK(hipMalloc((void **)&c->d_ctg_status, 4 * 8));
  HIPCHK(hipMemsetAsync(c->ctg_table.keys, 0xFF, cap * (size_t)c->nl * 8, c->stream));
  HIPCHK(hipMemsetAsync(c->ctg_table.vals, 0, cap * 2 * 4, c->stream));
  HIPCHK(hipMemsetAsync(c->d_ctg_status, 0, 4 * 8, c->stream));
  c->ctg_table.mask = cap - 1;
  c->ctg_cap = cap;
  return KC_OK;
}

extern "C" int kc_submit_ctg_block(kc_ctx *c, const char *seqs, const uint16_t *depths, uint64_t len, int on_device) {
  if (!c || (len && (!seqs || !depths))) return KC_ERR_INVALID_ARG;
  if (c->finalized || !c->ctg_table.keys) return KC_ERR_STATE;  // kc_begin_ctg_kmers first
  if (!len) return KC_OK;
  HIPCHK(hipSetDevice(c->cfg.device));
  const uint8_t *d_seqs = reinterpret_cast<const uint8_t *>(seqs);
  const uint16_t *d_depths = depths;
  uint8_t *tmp = nullptr;
  if (!on_device) {
    HIPCHK(hipMalloc((void **)&tmp, len * 3 + 16));
    HIPCHK(hipMemcpyAsync(tmp, seqs, len, hipMemcpyHostToDevice, c->stream));
    uint16_t *td = reinterpret_cast<uint16_t *>(tmp + ((len + 15) & ~(uint64_t)15));
    HIPCHK(hipMemcpyAsync(td, depths, len * 2, hipMemcpyHostToDevice, c->stream));
    d_seqs = tmp;
    d_depths = td;
  }
  // which k-mers this context keeps: everything as one rank; as one of several, what the read path keeps there
  CtgOwn own;
  memset(&own, 0, sizeof(own));
  own.mode = CTG_OWN_ALL;
  own.rank_me = (uint32_t)c->cfg.rank_me;
  own.rank_n = (uint32_t)c->cfg.rank_n;
  int rc = KC_OK;
  if (c->cfg.rank_n > 1) {
    if (c->sh.flow || (c->cfg.flags & KC_FLAG_SHARD_BUCKETS)) {  // the shard flow: the owner of the k-mer's level-1 bucket
      rc = bk_active(c) ? bk_init(c) : KC_ERR_STATE;
      own.mode = CTG_OWN_BUCKET;
      own.gm = c->gm;
      own.own_lo = shard_first_bucket((uint32_t)c->cfg.rank_me, c->gm.P1, (uint32_t)c->cfg.rank_n);
      own.own_hi = shard_first_bucket((uint32_t)c->cfg.rank_me + 1, c->gm.P1, (uint32_t)c->cfg.rank_n);
    } else {
      own.mode = (c->cfg.flags & KC_FLAG_REFERENCE_OWNER) ? CTG_OWN_REFERENCE : CTG_OWN_HASH;
    }
  }
  // A launch may add as many distinct k-mers as it has positions: it gets no more positions than the table has room for
  // below three quarters, so that a probe always ends at an empty slot (an estimate that is too low -- the driver passes
  // the reference's -- must end in KC_ERR_CAPACITY, not in a kernel that never returns).
  hipError_t e = hipSuccess;
  uint64_t st[2] = {0, c->ctg_new};
  for (uint64_t p0 = 0; rc == KC_OK && p0 < len;) {
    const uint64_t limit = c->ctg_cap / 4 * 3;
    if (st[1] >= limit) {
      snprintf(g_last_error, sizeof(g_last_error), "contig pass: %llu distinct k-mers fill the table kc_begin_ctg_kmers made (%llu slots)",
               (unsigned long long)st[1], (unsigned long long)c->ctg_cap);
      rc = KC_ERR_CAPACITY;
      break;
    }
    const uint64_t p1 = std::min<uint64_t>(len, p0 + (limit - st[1]));
    const unsigned nblk = (unsigned)((p1 - p0 + 255) / 256);
    switch (c->nl) {
      case 1: hipLaunchKernelGGL(kc_ctg_insert_kernel<1>, dim3(nblk), dim3(256), 0, c->stream, d_seqs, d_depths, p0, p1, len, c->k, c->ctg_table, c->d_ctg_status, own); break;
      case 2: hipLaunchKernelGGL(kc_ctg_insert_kernel<2>, dim3(nblk), dim3(256), 0, c->stream, d_seqs, d_depths, p0, p1, len, c->k, c->ctg_table, c->d_ctg_status, own); break;
      case 3: hipLaunchKernelGGL(kc_ctg_insert_kernel<3>, dim3(nblk), dim3(256), 0, c->stream, d_seqs, d_depths, p0, p1, len, c->k, c->ctg_table, c->d_ctg_status, own); break;
      default: hipLaunchKernelGGL(kc_ctg_insert_kernel<4>, dim3(nblk), dim3(256), 0, c->stream, d_seqs, d_depths, p0, p1, len, c->k, c->ctg_table, c->d_ctg_status, own); break;
    }
    c->num_gpu_calls++;
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(st, c->d_ctg_status, 16, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) break;
    c->ctg_new = st[1];
    p0 = p1;
  }
  if (tmp) (void)hipFree(tmp);
  if (e != hipSuccess) return hip_fail(e, "kc_submit_ctg_block", __LINE__);
  if (rc) return rc;
  if (st[0]) return KC_ERR_BAD_BASE;
  c->ctg_attempted += len;
  return KC_OK;
}

extern "C" int kc_arena_probe_rate(kc_ctx *c, double *tbps) {
  if (!c || !tbps) return KC_ERR_INVALID_ARG;
  *tbps = c->arena_probe_tbps;
  return KC_OK;
}

extern "C" int kc_ctg_stats(kc_ctx *c, uint64_t *distinct, uint64_t *positions) {
  if (!c) return KC_ERR_INVALID_ARG;
  if (distinct) *distinct = c->ctg_new;
  if (positions) *positions = c->ctg_attempted;
  return KC_OK;
}

template <int NL>
static int ctg_merge_t(kc_ctx *c) {
  int rc = ensure_index<NL>(c);  // over the reads' results
  if (rc) return rc;
  const uint64_t n_res = c->out_n;
  uint64_t *cur = c->d_ctg_status + 2;  // [2] cursor, [3] sum of the appended counts
  const unsigned nblk = (unsigned)((c->ctg_cap + 255) / 256);
  for (int pass = 0; pass < 2; pass++) {
    uint64_t init[2] = {n_res, 0};
    HIPCHK(hipMemcpyAsync(cur, init, 16, hipMemcpyHostToDevice, c->stream));
    hipLaunchKernelGGL(kc_ctg_merge_kernel<NL>, dim3(nblk), dim3(256), 0, c->stream, c->ctg_table, c->ctg_cap, c->d_index, c->index_cap - 1, n_res,
                       c->d_out_keys, c->d_out_counts, c->d_out_left, c->d_out_right, pass == 0 ? (uint64_t)0 : c->out_cap, NL, cur, cur + 1);
    c->num_gpu_calls++;
    HIPCHK(hipGetLastError());
    uint64_t got[2];
    HIPCHK(hipMemcpyAsync(got, cur, 16, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (pass == 0) {
      if (got[0] == n_res) return KC_OK;  // nothing to add
      // (the index stays valid: it points at the first n_res entries, which keep their places)
      uint32_t *index = c->d_index;
      const uint64_t icap = c->index_cap;
      c->d_index = nullptr;
      rc = grow_results(c, got[0], n_res);
      c->d_index = index;
      c->index_cap = icap;
      if (rc) return rc;
      c->out_n = n_res;
    } else {
      c->out_n = got[0];
      c->sum_counts += got[1];
    }
  }
  free_index(c);  // the results have changed: a later kc_lookup builds it again
  return KC_OK;
}

static int ctg_merge(kc_ctx *c) {
  switch (c->nl) {
    case 1: return ctg_merge_t<1>(c);
    case 2: return ctg_merge_t<2>(c);
    case 3: return ctg_merge_t<3>(c);
    default: return ctg_merge_t<4>(c);
  }
}

extern "C" int kc_get_kernel_times(kc_ctx *c, kc_kernel_time *out, int max, int *n) {
  if (!c || !n) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  drain_kernel_times(c);
  int m = 0;
  for (int i = 0; i < KT_COUNT; i++) {
    if (!c->kt_launches[i]) continue;
    if (out && m < max) {
      memset(&out[m], 0, sizeof(out[m]));
      snprintf(out[m].name, sizeof(out[m].name), "%s", kt_names[i]);
      out[m].launches = c->kt_launches[i];
      out[m].total_ms = c->kt_ms[i];
    }
    m++;
  }
  *n = m;
  return KC_OK;
}

extern "C" int kc_clear_kernel_times(kc_ctx *c) {
  if (!c) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  HIPCHK(hipStreamSynchronize(c->stream));
  drain_kernel_times(c);
  for (int i = 0; i < KT_COUNT; i++) {
    c->kt_launches[i] = 0;
    c->kt_ms[i] = 0;
  }
  return KC_OK;
}

// ---- synthetic reads ---------------------------------------------------------------------------
extern "C" void kc_synth_default_params(kc_synth_params *p) {
  if (!p) return;
  p->seed = 1234;
  p->num_genomes = 64;
  p->read_len = 150;
  p->min_genome_len = 2000000;
  p->max_genome_len = 6000000;
  p->sub_error_rate = 0.005;
  p->lowq_rate = 0.01;
  p->n_rate = 0.0;
  p->abundance_sigma = 1.0;
}

static uint64_t rate_to_thr(double r) {
  if (r <= 0) return 0;
  if (r >= 1) return ~0ULL;
  return (uint64_t)((long double)r * 18446744073709551616.0L);
}

static int build_synth_table(const kc_synth_params *p, kc_synth_table *t) {
  if (!p || p->num_genomes < 1 || p->num_genomes > KC_SYNTH_MAX_GENOMES || p->read_len < 1 || p->min_genome_len < p->read_len ||
      p->max_genome_len < p->min_genome_len)
    return KC_ERR_INVALID_ARG;
  memset(t, 0, sizeof(*t));
  t->seed = p->seed;
  t->num_genomes = p->num_genomes;
  t->read_len = p->read_len;
  t->err_thr = rate_to_thr(p->sub_error_rate);
  t->lowq_thr = rate_to_thr(p->sub_error_rate + p->lowq_rate);
  t->n_thr = rate_to_thr(p->n_rate);
  std::vector<long double> w(p->num_genomes);
  long double tot = 0;
  for (uint32_t g = 0; g < p->num_genomes; g++) {
    uint64_t s = kc_splitmix(p->seed * 0x2545F4914F6CDD1DULL + g);
    t->genome_len[g] = p->min_genome_len + s % (p->max_genome_len - p->min_genome_len + 1);
    t->genome_seed[g] = kc_splitmix(s ^ 0x5851F42D4C957F2DULL);
    // log-normal abundance (seed+1 stream), Box-Muller
    uint64_t a = kc_splitmix((p->seed + 1) * 0x9E3779B97F4A7C15ULL + 2 * g), b = kc_splitmix((p->seed + 1) * 0x9E3779B97F4A7C15ULL + 2 * g + 1);
    double u1 = ((double)(a >> 11) + 1.0) / 9007199254740993.0, u2 = (double)(b >> 11) / 9007199254740992.0;
    double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
    w[g] = (long double)exp(p->abundance_sigma * z) * (long double)t->genome_len[g];
    tot += w[g];
  }
  long double acc = 0;
  for (uint32_t g = 0; g < p->num_genomes; g++) {
    acc += w[g];
    long double f = acc / tot;
    t->cum[g] = (f >= 1.0L) ? ~0ULL : (uint64_t)(f * 18446744073709551615.0L);
  }
  t->cum[p->num_genomes - 1] = ~0ULL;
  return KC_OK;
}

extern "C" int kc_synth_reads_host(const kc_synth_params *p, uint64_t first_read, uint64_t nreads, uint8_t *bases, uint8_t *quals,
                                   uint64_t *offsets) {
  kc_synth_table *t = new (std::nothrow) kc_synth_table;
  if (!t) return KC_ERR_OUT_OF_MEMORY;
  int rc = build_synth_table(p, t);
  if (rc) {
    delete t;
    return rc;
  }
  const uint32_t L = t->read_len;
  for (uint64_t r = 0; r < nreads; r++) {
    uint64_t rstate, start;
    uint32_t g;
    bool rev;
    kc_synth_read_header(t, first_read + r, &rstate, &g, &start, &rev);
    for (uint32_t i = 0; i < L; i++) kc_synth_base(t, rstate, g, start, rev, i, &bases[r * L + i], &quals[r * L + i]);
    offsets[r] = r * L;
  }
  offsets[nreads] = nreads * L;
  delete t;
  return KC_OK;
}

extern "C" int kc_synth_reads_device(kc_ctx *c, const kc_synth_params *p, uint64_t first_read, uint64_t nreads, uint8_t *d_bases,
                                     uint8_t *d_quals, uint64_t *d_offsets) {
  if (!c || !d_bases || !d_quals || !d_offsets) return KC_ERR_INVALID_ARG;
  HIPCHK(hipSetDevice(c->cfg.device));
  kc_synth_table *t = new (std::nothrow) kc_synth_table;
  if (!t) return KC_ERR_OUT_OF_MEMORY;
  int rc = build_synth_table(p, t);
  if (rc) {
    delete t;
    return rc;
  }
  if (!c->d_synth) {
    hipError_t e = hipMalloc((void **)&c->d_synth, sizeof(kc_synth_table));
    if (e != hipSuccess) {
      delete t;
      return hip_fail(e, "hipMalloc(synth table)", __LINE__);
    }
  }
  hipError_t e = hipMemcpy(c->d_synth, t, sizeof(kc_synth_table), hipMemcpyHostToDevice);
  delete t;
  if (e != hipSuccess) return hip_fail(e, "hipMemcpy(synth table)", __LINE__);
  const uint64_t nquads = nreads * ((p->read_len + 3) / 4);
  unsigned nblk = (unsigned)std::min<uint64_t>((nquads + 255) / 256, 256 * 64);
  if (!nblk) nblk = 1;
  hipLaunchKernelGGL(kc_synth_kernel, dim3(nblk), dim3(256), 0, c->stream, c->d_synth, first_read, nreads, d_bases, d_quals, d_offsets);
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(c->stream));
  return KC_OK;
}
