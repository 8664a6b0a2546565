// kc_kernels.hpp -- gfx950 kernels of libkcount_mi355 (wave64, no MFMA: integer / hash work).
//
//   kc_extract_kernel      reads -> canonical k-mer records (+ packed extension codes), inserted into this
//                          shard's global table (the bucketed path and the sender-side binning share its tile
//                          staging and k-mer cutting: kc_bucketed.hpp).
//                          Replaces parse_and_pack / build_supermers / pack_seqs and
//                          gpu_unpack_supermer_block / get_kmer_from_supermer of the reference
//                          (src/kcount/kcount-gpu/parse_and_pack.cpp:127-237,
//                          gpu_hash_table.cpp:281-355) with the CPU backend's semantics (S1-S5).
//   kc_insert_records_kernel  records -> table (receiver side; gpu_insert_kmer's role,
//                          gpu_hash_table.cpp:357-424, with S6 saturation instead of the GPU variant's).
//   kc_finalize_kernel     S7 vote + S8 purge + dense compaction (replaces gpu_purge_invalid +
//                          gpu_compact_ht + the host loop of kcount_gpu.cpp:438-471).
//
// A tile is TILE consecutive base positions of the concatenated read array.  Each workgroup stages
// its tile once in LDS as 2-bit codes (32 bases per u64 word), an "extension usable" bit per base and
// a "read boundary" bit per gap; every k-mer is then cut out of the packed words with one funnel
// shift per word instead of being re-packed from k ASCII bytes per thread.
#pragma once
#include "kc_common.hpp"

namespace kc {

constexpr int TPB = 256;
constexpr int PPT = 16;                 // positions per thread
constexpr int TILE = TPB * PPT;         // 4096 positions per workgroup
constexpr int PRE = 32;                 // staged positions before the tile (left neighbour of its first k-mer)
constexpr int POST = 192;               // staged positions after it: k+1 <= 128, + one word for the funnel, + slack
// geometry of a staged tile: THREADS threads share it, the k-mers starting at its SPAN positions are cut from it
template <int THREADS_, int SPAN_>
struct TileGeo {
  static constexpr int THREADS = THREADS_;
  static constexpr int SPAN = SPAN_;
  static constexpr int LSPAN = PRE + SPAN_ + POST;                // staged positions
  static constexpr int NGROUP = LSPAN / 16;                       // sixteen-base groups
  static constexpr int NWORD = LSPAN / 32;
  static constexpr int GPT = (NGROUP + THREADS_ - 1) / THREADS_;  // groups a thread stages
  static_assert(LSPAN % 32 == 0, "staged span must be whole words");
};
using TileSmall = TileGeo<TPB, TILE>;  // the global-table kernel: 4320 staged positions, 270 groups, 2 per thread

constexpr uint64_t KEY_EMPTY = ~0ULL;
constexpr uint64_t KEY_BUSY = ~0ULL - 1;
constexpr uint64_t KEY_NEVER = ~0ULL - 2;  // no claim word ever holds it (a key's last word ends in six zero bits)

enum { MODE_INSERT = 0, MODE_BIN = 1 };
// input formats: ASCII bases + qualities; the reference's '_'-joined case-masked block; the reference's read
// cache bytes (3-bit base | 5-bit quality << 3, src/packed_reads.cpp:99-126)
// FMT_READS_UQ: FMT_READS whose quality array is not 16-byte co-aligned with the base array (its own
// instantiation: the byte loads it needs would otherwise cost the common case registers)
enum { FMT_READS = 0, FMT_SEQBLOCK = 1, FMT_PACKED = 2, FMT_READS_UQ = 3 };
constexpr bool fmt_is_reads(int fmt) { return fmt == FMT_READS || fmt == FMT_READS_UQ; }

// device-side counters, one u64 each (host mirror in kc_api)
enum {
  CTR_ENTRIES = 0,   // occupied table slots
  CTR_INSERTED,      // k-mer occurrences inserted
  CTR_BAD_BASE,      // != 0: a byte outside the alphabet was seen
  CTR_OUT,           // finalize: results written
  CTR_PURGED,
  CTR_SUM_COUNTS,
  CTR_RAW_KMERS,
  CTR_EXPECT,        // k-mer occurrences with two neighbours submitted so far (upper bound of what is buffered)
  CTR_OVERFLOW,      // bin mode: a segment was too small
  CTR_BIN0,          // [CTR_BIN0 + d]: records binned for shard d (64 slots)
  CTR_COUNT = CTR_BIN0 + 64
};

struct Table {
  uint64_t *keys;   // capacity * NL words, 0xFF-filled when empty
  uint32_t *vals;   // capacity * 9: count, left ACGT, right ACGT (clipped to 65535 when read out: S6)
  uint64_t mask;    // capacity - 1 (capacity is a power of two)
};

struct ExtractArgs {
  const uint8_t *bases;     // 16-byte aligned base address (bases - align)
  const uint8_t *quals;     // same shift applied; only FMT_READS
  const uint64_t *offsets;  // nreads+1, relative to the first real byte; only FMT_READS
  const uint64_t *tile_first;  // per tile: first read whose start is at or after the tile's first position
  uint64_t nreads;
  uint64_t total;           // bytes of real data
  uint32_t align;           // real data starts at aligned coordinate `align` (0..15)
  int64_t pos0;             // aligned coordinate where the first tile of this launch starts
  int k;
  int qual_cut;             // qual_offset + KC_QUAL_CUTOFF
  uint32_t rank_me, rank_n;
  uint32_t reference_owner;  // owner shard = the reference's get_kmer_target_rank instead of the k-mer hash
  // MODE_BIN
  uint64_t *records;
  uint64_t seg_capacity;
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// ---- table ------------------------------------------------------------------------------
__device__ __forceinline__ void sat_inc(uint32_t *p) {
  uint32_t old = atomicAdd(p, 1u);
  if (old >= 0x80000000u) atomicSub(p, 1u);  // far above 65535 already; keeps the u32 from wrapping
}

// The slot of a canonical k-mer (extension bits cleared), claimed if the k-mer is new.  Linear probing, power-of-two
// capacity; the host keeps the load below 0.9 by growing the table, so the probe always terminates and
// nothing is ever dropped (the reference drops after KCOUNT_HT_MAX_PROBE, kcount_cpu.cpp:232-268).
template <int NL>
__device__ __forceinline__ uint64_t table_slot(const Table &t, const uint64_t (&key)[NL], bool &is_new) {
  uint64_t slot = kc_hash<NL>(key) & t.mask;
  is_new = false;
  if constexpr (NL == 1) {
    for (;;) {
      // plain load as a hint: a slot only ever goes EMPTY -> key, so a stale line can only claim
      // EMPTY, and then the CAS below tells the truth
      uint64_t cur = t.keys[slot];
      if (cur == key[0]) break;
      if (cur == KEY_EMPTY) {
        uint64_t old = atomicCAS((unsigned long long *)&t.keys[slot], (unsigned long long)KEY_EMPTY,
                                 (unsigned long long)key[0]);
        if (old == KEY_EMPTY) { is_new = true; break; }
        if (old == key[0]) break;
      }
      slot = (slot + 1) & t.mask;
    }
  } else {
    // the last word is lock + publish: EMPTY -> BUSY -> value.  Every read of shared key words is a
    // device-scope RMW so that it is served by the coherent memory side, never by another XCD's L2.
    // Predicated form (a flag per lane, no divergent exit): the lane that claimed a slot publishes inside the very
    // trip in which another lane of its wave may have seen BUSY, whatever the compiler makes of the control flow.
    bool done = false;
    do {
      unsigned long long *ks = (unsigned long long *)(t.keys + slot * NL);
      unsigned long long old = KEY_BUSY;
      if (!done) old = atomicCAS(&ks[NL - 1], (unsigned long long)KEY_EMPTY, (unsigned long long)KEY_BUSY);
      const bool won = !done && old == KEY_EMPTY;
      if (won) {
#pragma unroll
        for (int j = 0; j < NL - 1; j++)
          __hip_atomic_store(&ks[j], (unsigned long long)key[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the word stores must be performed before the publish below; inline asm is not reordered or dropped
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        atomicExch(&ks[NL - 1], (unsigned long long)key[NL - 1]);
        is_new = true;
      }
      bool same = !done && old == key[NL - 1];
      if (same) {
#pragma unroll
        for (int j = 0; j < NL - 1; j++) same &= (atomicOr(&ks[j], 0ULL) == key[j]);
      }
      const bool busy = !done && old == KEY_BUSY;  // its owner publishes within this same trip of its own wave: look again
      if (!done && !won && !same && !busy) slot = (slot + 1) & t.mask;
      done = done || won || same;
    } while (__any(!done));
  }
  return slot;
}

// Insert one record (canonical k-mer + extension codes).
template <int NL>
__device__ __forceinline__ void table_insert(const Table &t, const uint64_t (&rec)[NL], uint64_t *ctrs) {
  uint64_t key[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) key[j] = rec[j];
  const uint32_t le = (uint32_t)(rec[NL - 1] & 7u), re = (uint32_t)((rec[NL - 1] >> 3) & 7u);
  key[NL - 1] &= ~KC_EXT_MASK;
  bool is_new;
  const uint64_t slot = table_slot<NL>(t, key, is_new);
  uint32_t *v = t.vals + slot * 9;
  sat_inc(v);                              // S6 count
  if (le < 4u) sat_inc(v + 1 + le);        // S5/S6 only ACGT extensions are counted
  if (re < 4u) sat_inc(v + 5 + re);
  if (is_new) atomicAdd((unsigned long long *)&ctrs[CTR_ENTRIES], 1ULL);
}

// Add an already counted k-mer (its occurrences and its eight extension counters, each clipped to 65535 by whoever
// counted them: a sum of clipped values clips to the same 65535 as the sum of the unclipped ones, S6).
__device__ __forceinline__ void sat_add(uint32_t *p, uint32_t x) {
  if (!x) return;
  const uint32_t old = atomicAdd(p, x);
  if (old >= 0x80000000u) atomicSub(p, x);  // far above 65535 already; keeps the u32 from wrapping
}
template <int NL>
__global__ __launch_bounds__(TPB) void kc_merge_entries_kernel(const uint64_t *keys, const uint16_t *counts, const uint16_t *exts, uint64_t n,
                                                               Table t, uint64_t *ctrs) {
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
    uint64_t key[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) key[j] = keys[i * NL + j];
    bool is_new;
    const uint64_t slot = table_slot<NL>(t, key, is_new);
    uint32_t *v = t.vals + slot * 9;
    sat_add(v, counts[i]);
#pragma unroll
    for (int x = 0; x < 8; x++) sat_add(v + 1 + x, exts[i * 8 + x]);
    if (is_new) atomicAdd((unsigned long long *)&ctrs[CTR_ENTRIES], 1ULL);
  }
}

// ---- tile staging --------------------------------------------------------------------------
template <class G>
struct alignas(16) TileLDS {
  uint32_t codes[G::NGROUP];   // 2-bit codes; u64 word w = {codes[2w+1] (first 16 bases), codes[2w]}
  uint16_t ok[G::NGROUP];      // bit i: base 16g+i may serve as an extension (high quality, ACGT)
  uint32_t gap[G::NWORD + 1];  // bit lp: a read boundary lies between local positions lp-1 and lp
  uint64_t far_off;            // offsets[first read of the tile + THREADS - 1] (~0 past the end): see tile_encode
};

__device__ __forceinline__ uint32_t pack4(uint32_t v) {  // 4 ASCII bytes -> 4 codes, first byte highest
  uint32_t b = (v >> 1) & 0x03030303u;
  uint32_t c = b ^ ((b >> 1) & 0x01010101u);
  return (c * 0x40100401u) >> 24;
}

__device__ __forceinline__ uint32_t pack4_cache(uint32_t v) {  // 4 read-cache bytes (base 0-4 = ACGTN) -> 4 codes, N -> G
  uint32_t c = (v & 0x03030303u) | ((v >> 1) & 0x02020202u);
  return (c * 0x40100401u) >> 24;
}

constexpr uint32_t BM_ACGT = (1u << 1) | (1u << 3) | (1u << 7) | (1u << 20);
constexpr uint32_t BM_ACGTN = BM_ACGT | (1u << 14);
__device__ __forceinline__ bool in_bitmap(uint32_t c, uint32_t bm) { return ((c & 0xC0u) == 0x40u) && ((bm >> (c & 31u)) & 1u); }

// ---- sixteen bytes at a time -----------------------------------------------------------------------------------
// Byte-parallel predicates: a result word carries its answer in bit 7 of every byte, the other bits are garbage
// until the final gather.  nz7(x): bit 7 set where the byte of x is not zero.
__device__ __forceinline__ uint32_t nz7(uint32_t x) { return ((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x; }
// S2/S5 for a group whose sixteen bytes are all real data: okm bit i = byte i may serve as an extension,
// sepm bit i = byte i is a separator (FMT_SEQBLOCK), bad |= a byte outside the alphabet
template <int FMT>
__device__ __forceinline__ void encode_group_swar(const uint32_t (&bw)[4], const uint32_t (&qw)[4], uint32_t qual_cut, uint32_t &okm,
                                                  uint32_t &sepm, bool &bad) {
  // one word after the other, each folded into the running gathers at once (few values live at a time)
  uint32_t badacc = 0, ok_lo = 0, ok_hi = 0, sp_lo = 0, sp_hi = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const uint32_t w = bw[j];
    const uint32_t weight = (j & 1) ? 0x80402010u : 0x08040201u;  // byte i of the half-group -> bit i (times 0x80)
    uint32_t ok, sp = 0;
    if (FMT == FMT_PACKED) {
      // base = low 3 bits (0-4), quality = high 5: quality >= 20 <=> byte >= 0xA0; base > 4 <=> bit2 & (bit1 | bit0)
      const uint32_t x = w & ((w << 1) | (w << 2));  // bit 7: high quality, bit 2: bad base code
      badacc |= x << 5;
      ok = x & ~(w << 5);                            // high quality and base < 4
    } else {
      const uint32_t u = w & 0xDFDFDFDFu;  // upper case
      const uint32_t not_acgt = nz7(u ^ 0x41414141u) & nz7(u ^ 0x43434343u) & nz7(u ^ 0x47474747u) & nz7(u ^ 0x54545454u);
      const uint32_t not_n = nz7(u ^ 0x4E4E4E4Eu);
      uint32_t hq;
      if (fmt_is_reads(FMT)) {
        const uint32_t q = qw[j];
        hq = (((q & 0x7F7F7F7Fu) | 0x80808080u) - qual_cut * 0x01010101u) | q;  // S2: byte >= qual_cut (<= 128)
        badacc |= not_acgt & not_n;
      } else {
        hq = ~(w << 2);                    // case carries the quality: bit 5 clear
        const uint32_t not_sep = nz7(w ^ 0x5F5F5F5Fu);
        sp = ~not_sep;
        badacc |= not_acgt & not_n & not_sep;
      }
      ok = hq & ~not_acgt;
    }
    if (j < 2) {
      ok_lo = __builtin_amdgcn_udot4(ok & 0x80808080u, weight, ok_lo, false);
      if (FMT == FMT_SEQBLOCK) sp_lo = __builtin_amdgcn_udot4(sp & 0x80808080u, weight, sp_lo, false);
    } else {
      ok_hi = __builtin_amdgcn_udot4(ok & 0x80808080u, weight, ok_hi, false);
      if (FMT == FMT_SEQBLOCK) sp_hi = __builtin_amdgcn_udot4(sp & 0x80808080u, weight, sp_hi, false);
    }
  }
  okm = (ok_lo >> 7) | ((ok_hi >> 7) << 8);
  if (FMT == FMT_SEQBLOCK) sepm = (sp_lo >> 7) | ((sp_hi >> 7) << 8);
  if (badacc & 0x80808080u) bad = true;
}

// Staging a tile has two halves so that a kernel can keep the next tile's bytes in flight while it works on the
// current one: tile_prefetch issues the global loads into registers (nothing waits on them), tile_encode turns
// the registers into the LDS image.  Aligned coordinate x = byte index from the 16-byte aligned base address;
// local position lp = x - (T0 - PRE).  Thread tid (of the G::THREADS threads sharing the tile) owns the
// 16-byte groups tid, tid + G::THREADS, ...
// Barrier for data exchanged through LDS only: __syncthreads() would also drain the vector-memory counter, i.e.
// wait for every global store and prefetch load this wave still has in flight.
__device__ __forceinline__ void tile_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class G>
struct TileRaw {
  uint32_t bw[G::GPT][4];
  uint32_t qw[G::GPT][4];
  uint64_t first_read;  // first read starting at or after the tile's first position
  uint64_t off;         // offsets[first_read + tid], or ~0 past the end
};

template <int FMT, class G>
__device__ __forceinline__ void tile_prefetch(TileRaw<G> &R, const ExtractArgs &a, int64_t T0, int tid, uint64_t tile_first_read, bool active) {
  const int64_t lo = a.align, hi = (int64_t)a.align + (int64_t)a.total;  // real data is [lo, hi)
#pragma unroll
  for (int gi = 0; gi < G::GPT; gi++) {
#pragma unroll
    for (int w = 0; w < 4; w++) {
      R.bw[gi][w] = 0;
      R.qw[gi][w] = 0;
    }
    const int g = tid + gi * G::THREADS;
    const int64_t X0 = T0 - PRE + 16 * g;
    if (active && g < G::NGROUP && (X0 + 16 > lo) && (X0 < hi)) {
      // the 16-byte line holding the group is inside the allocation's aligned span whenever any
      // of its bytes is real data, so the vector load is safe; bytes outside [lo,hi) are masked when encoding
      const uint4 v = *reinterpret_cast<const uint4 *>(a.bases + X0);
      R.bw[gi][0] = v.x; R.bw[gi][1] = v.y; R.bw[gi][2] = v.z; R.bw[gi][3] = v.w;
      if (fmt_is_reads(FMT)) {
        if (FMT == FMT_READS) {
          const uint4 q = *reinterpret_cast<const uint4 *>(a.quals + X0);
          R.qw[gi][0] = q.x; R.qw[gi][1] = q.y; R.qw[gi][2] = q.z; R.qw[gi][3] = q.w;
        } else {
          for (int i = 0; i < 16; i++) {
            const int64_t x = X0 + i;
            if (x >= lo && x < hi) R.qw[gi][i >> 2] |= (uint32_t)a.quals[x] << (8 * (i & 3));
          }
        }
      }
    }
  }
  R.first_read = tile_first_read;
  R.off = ~0ULL;
  if (FMT != FMT_SEQBLOCK && active && tile_first_read + tid <= a.nreads) R.off = a.offsets[tile_first_read + tid];
}

// tile_encode in two halves for a kernel that has barriers of its own to put between and behind them (kc_l1_reads16_kernel):
// tile_encode_clear, BARRIER, tile_encode_fill, BARRIER.
template <class G>
__device__ __forceinline__ void tile_encode_clear(TileLDS<G> &L, const TileRaw<G> &R, int tid) {
  for (int i = tid; i < G::NWORD + 1; i += G::THREADS) L.gap[i] = 0;
  if (tid == G::THREADS - 1) L.far_off = R.off;
}
template <int FMT, class G>
__device__ __forceinline__ void tile_encode_fill(TileLDS<G> &L, const TileRaw<G> &R, const ExtractArgs &a, int64_t T0, uint64_t *ctrs, int tid,
                                                 bool active);

template <int FMT, class G>
__device__ __forceinline__ void tile_encode(TileLDS<G> &L, const TileRaw<G> &R, const ExtractArgs &a, int64_t T0, uint64_t *ctrs, int tid,
                                            bool active) {
  // every thread of the workgroup reaches both barriers
  tile_encode_clear<G>(L, R, tid);
  tile_barrier();
  tile_encode_fill<FMT, G>(L, R, a, T0, ctrs, tid, active);
  tile_barrier();
}

template <int FMT, class G>
__device__ __forceinline__ void tile_encode_fill(TileLDS<G> &L, const TileRaw<G> &R, const ExtractArgs &a, int64_t T0, uint64_t *ctrs, int tid,
                                                 bool active) {
#ifdef KC_STAMPS
  unsigned long long ts_ = __builtin_amdgcn_s_memtime();
#define KC_ENC_STAMP(k)                                                              \
  if (threadIdx.x == 0) {                                                            \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime();                     \
    atomicAdd((unsigned long long *)&ctrs[CTR_BIN0 + 40 + k], tn_ - ts_);            \
    ts_ = tn_;                                                                       \
  }
#else
#define KC_ENC_STAMP(k)
#endif
  KC_ENC_STAMP(0)
  if (!active) return;
  const int64_t lo = a.align, hi = (int64_t)a.align + (int64_t)a.total;
  bool bad = false;
#pragma unroll
  for (int gi = 0; gi < G::GPT; gi++) {
    const int g = tid + gi * G::THREADS;
    if (g >= G::NGROUP) break;
    const int64_t X0 = T0 - PRE + 16 * g;
    const uint32_t(&bw)[4] = R.bw[gi];
    const uint32_t(&qw)[4] = R.qw[gi];
    const bool any = (X0 + 16 > lo) && (X0 < hi);
    const bool full = (X0 >= lo) && (X0 + 16 <= hi);
    uint32_t code = FMT == FMT_PACKED ? (pack4_cache(bw[0]) << 24) | (pack4_cache(bw[1]) << 16) | (pack4_cache(bw[2]) << 8) | pack4_cache(bw[3])
                                      : (pack4(bw[0]) << 24) | (pack4(bw[1]) << 16) | (pack4(bw[2]) << 8) | pack4(bw[3]);
    uint32_t okm = 0, sepm = 0;
    if (full && (!fmt_is_reads(FMT) || a.qual_cut <= 128)) {
      encode_group_swar<FMT>(bw, qw, (uint32_t)a.qual_cut, okm, sepm, bad);
    } else {
      // groups at the edges of the data: byte by byte, bytes outside [lo, hi) masked
#pragma unroll
      for (int j = 0; j < 4; j++) {
        uint32_t wb = bw[j], wq = qw[j];
#pragma unroll 1
        for (int i = 4 * j; i < 4 * j + 4; i++, wb >>= 8, wq >>= 8) {
          const uint32_t c = wb & 0xFFu;
          const bool real = full || (any && (X0 + i >= lo) && (X0 + i < hi));
          bool hq;
          if (FMT == FMT_PACKED) {
            hq = (c >> 3) >= KC_QUAL_CUTOFF;                      // S2 on the stored quality (already relative to qual_offset)
            if (real && (c & 7u) > 4u) bad = true;
            if (real && hq && (c & 7u) < 4u) okm |= 1u << i;
            continue;
          }
          if (fmt_is_reads(FMT)) {
            const int q = (int)(wq & 0xFFu);
            hq = q >= a.qual_cut;                                 // S2
            if (real && !in_bitmap(c, BM_ACGTN)) bad = true;
          } else {
            hq = (c & 0x20u) == 0;                                // case carries the quality
            const bool sep = (c == '_');
            if (real && sep) sepm |= 1u << i;
            if (real && !sep && !in_bitmap(c, BM_ACGTN)) bad = true;
          }
          if (real && hq && in_bitmap(c, BM_ACGT)) okm |= 1u << i;
        }
      }
    }
    L.codes[g ^ 1] = code;
    L.ok[g] = (uint16_t)okm;
    if (!full) {
      // positions outside the data count as read boundaries, so that no window can reach into them
      uint32_t nonreal = 0xFFFFu;
      if (any) {
        const int i0 = (int)(lo > X0 ? lo - X0 : 0), i1 = (int)(hi - X0 < 16 ? hi - X0 : 16);
        nonreal = ~(((1u << i1) - 1u) & ~((1u << i0) - 1u)) & 0xFFFFu;
      }
      atomicOr(&L.gap[g >> 1], nonreal << (16 * (g & 1)));
    }
    if (FMT == FMT_SEQBLOCK && sepm) {
      // a separator at q kills every window [p-1, p+k] that contains q: gaps q and q+1
      uint64_t bits = ((uint64_t)sepm | ((uint64_t)sepm << 1)) << (16 * (g & 1));
      atomicOr(&L.gap[g >> 1], (uint32_t)bits);
      if (bits >> 32) atomicOr(&L.gap[(g >> 1) + 1], (uint32_t)(bits >> 32));
    }
  }
  KC_ENC_STAMP(1)
  if (FMT != FMT_SEQBLOCK) {
    // boundaries from the read offsets (the end of the data is offsets[nreads]); the first one of this thread
    // came with the prefetch.  More are loaded only when the tile holds more reads than it has threads, which the
    // last thread's offset tells everybody (offsets only grow): otherwise the threads that own a boundary would
    // each wait for one more load just to learn that it lies beyond the tile.
    const int64_t first = T0, last = T0 + G::SPAN + a.k;  // gaps that any window of this tile can contain
    const uint64_t far = L.far_off;
    const bool more = far != ~0ULL && (int64_t)far + lo <= last;
    uint64_t off = R.off;
    for (uint64_t r = R.first_read + tid; r <= a.nreads; r += G::THREADS) {
      if (r != R.first_read + tid) off = a.offsets[r];
      const int64_t s = (int64_t)off + lo;
      if (s > last) break;
      if (s >= first) {
        const int lp = (int)(s - (T0 - PRE));
        atomicOr(&L.gap[lp >> 5], 1u << (lp & 31));
      }
      if (!more) break;
    }
  } else {
    // start and end of the block are boundaries too
    if (tid == 0) {
      int64_t s0 = lo - (T0 - PRE), s1 = hi - (T0 - PRE);
      if (s0 >= 0 && s0 < G::LSPAN) atomicOr(&L.gap[s0 >> 5], 1u << (s0 & 31));
      if (s1 >= 0 && s1 < G::LSPAN) atomicOr(&L.gap[s1 >> 5], 1u << (s1 & 31));
    }
  }
  if (bad) ctrs[CTR_BAD_BASE] = 1;
  KC_ENC_STAMP(2)
}

// both halves back to back (kernels that do not pipeline their tiles)
template <int FMT, class G>
__device__ __forceinline__ void stage_tile(TileLDS<G> &L, const ExtractArgs &a, int64_t T0, uint64_t *ctrs, int tid, uint64_t tile_first_read,
                                           bool active) {
  TileRaw<G> R;
  tile_prefetch<FMT, G>(R, a, T0, tid, tile_first_read, active);
  tile_encode<FMT, G>(L, R, a, T0, ctrs, tid, active);
}

// Cut the k-mer that starts at local position lp out of the staged tile.  Returns false if the
// window [lp-1, lp+k] crosses a read boundary (S1, S5).  rec = canonical k-mer (S3, S4) with the
// extension codes (S5) in the low 6 bits of its last word; h = hash of the bare k-mer.
template <int NL, class TL>
__device__ __forceinline__ bool tile_kmer(const TL &L, int lp, int k, uint64_t (&rec)[NL], uint64_t &h, uint32_t rank_n = 1,
                                          uint32_t reference_owner = 0, uint32_t *owner = nullptr) {
  {  // any boundary among gaps lp .. lp+k ?
    int rem = k + 1, w = lp >> 5, s = lp & 31;
    uint32_t badbits = L.gap[w] >> s;
    int got = 32 - s;
    if (got >= rem) {
      if (rem < 32) badbits &= (1u << rem) - 1u;
    } else {
      rem -= got;
      w++;
      while (rem >= 32) { badbits |= L.gap[w++]; rem -= 32; }
      if (rem) badbits |= L.gap[w] & ((1u << rem) - 1u);
    }
    if (badbits) return false;
  }
  const uint64_t *W = reinterpret_cast<const uint64_t *>(L.codes);
  const int q = lp >> 5, sh = 2 * (lp & 31);
  uint64_t f[NL], r[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) {
    uint64_t hi = W[q + j], lo = W[q + j + 1];
    f[j] = (sh ? ((hi << sh) | (lo >> (64 - sh))) : hi) & kc_word_mask(k, j);
  }
  kc_revcomp<NL>(f, k, r);
  // extension codes of the two neighbours
  const int pl = lp - 1, pr = lp + k;
  uint32_t lc = (uint32_t)(W[pl >> 5] >> (62 - 2 * (pl & 31))) & 3u;
  uint32_t rc = (uint32_t)(W[pr >> 5] >> (62 - 2 * (pr & 31))) & 3u;
  uint32_t le = ((L.ok[pl >> 4] >> (pl & 15)) & 1u) ? lc : KC_EXT_NONE;
  uint32_t re = ((L.ok[pr >> 4] >> (pr & 15)) & 1u) ? rc : KC_EXT_NONE;
  if (owner) *owner = 0;
  if (owner && rank_n > 1 && reference_owner) *owner = kc_reference_owner<NL>(f, r, k, rank_n);
  if (kc_less<NL>(r, f)) {  // strict: a palindrome keeps the forward extensions
#pragma unroll
    for (int j = 0; j < NL; j++) f[j] = r[j];
    uint32_t nl = (re == KC_EXT_NONE) ? KC_EXT_NONE : 3u - re;
    uint32_t nr = (le == KC_EXT_NONE) ? KC_EXT_NONE : 3u - le;
    le = nl;
    re = nr;
  }
  h = kc_hash<NL>(f);
  if (owner && rank_n > 1 && !reference_owner) *owner = kc_owner_of_hash(h, rank_n);
#pragma unroll
  for (int j = 0; j < NL; j++) rec[j] = f[j];
  rec[NL - 1] |= (uint64_t)(le | (re << 3));
  return true;
}

// Rolling form of tile_kmer for a run of consecutive positions lp0, lp0+1, ...: the first k-mer and its reverse
// complement are cut out once, every further one costs a 2-bit shift of both; the boundary test becomes "no boundary
// gap at or after the position among those seen so far".  Same results as tile_kmer position by position.
template <int NL>
struct KmerRun {
  uint64_t f[NL], r[NL];  // forward k-mer and reverse complement at the current position
  uint64_t nextb;         // bases lp+k, lp+k+1, ...: 2 bits each, the first one highest
  uint32_t okl, okr;      // bit j: base lp0-1+j / base lp0+k+j may serve as an extension
  uint32_t gin;           // bit j: read boundary at gap lp0+k+1+j
  uint32_t lc;            // code of base lp-1
  int lastgap;            // highest boundary gap <= lp+k (-1: none)
};

template <int NL, class TL>
__device__ __forceinline__ void run_begin(KmerRun<NL> &s, const TL &L, int lp0, int k) {
  const uint64_t *W = reinterpret_cast<const uint64_t *>(L.codes);
  {
    const int q = lp0 >> 5, sh = 2 * (lp0 & 31);
#pragma unroll
    for (int j = 0; j < NL; j++) {
      const uint64_t hi = W[q + j], lo = W[q + j + 1];
      s.f[j] = (sh ? ((hi << sh) | (lo >> (64 - sh))) : hi) & kc_word_mask(k, j);
    }
  }
  kc_revcomp<NL>(s.f, k, s.r);
  const int pl = lp0 - 1, pr = lp0 + k;
  {
    const int q = pr >> 5, sh = 2 * (pr & 31);
    const uint64_t hi = W[q], lo = W[q + 1];
    s.nextb = sh ? ((hi << sh) | (lo >> (64 - sh))) : hi;
  }
  s.lc = (uint32_t)(W[pl >> 5] >> (62 - 2 * (pl & 31))) & 3u;
  s.okl = ((uint32_t)L.ok[pl >> 4] | ((uint32_t)L.ok[(pl >> 4) + 1] << 16)) >> (pl & 15);
  s.okr = ((uint32_t)L.ok[pr >> 4] | ((uint32_t)L.ok[(pr >> 4) + 1] << 16)) >> (pr & 15);
  {
    const int a = pr + 1, w = a >> 5, sh = a & 31;
    const uint32_t g0 = L.gap[w], g1 = L.gap[w + 1];
    s.gin = sh ? ((g0 >> sh) | (g1 << (32 - sh))) : g0;
  }
  int last = -1;
  const int w0 = lp0 >> 5, w1 = pr >> 5;
  for (int w = w0; w <= w1; w++) {
    uint32_t bits = L.gap[w];
    if (w == w0) bits &= ~0u << (lp0 & 31);
    if (w == w1 && (pr & 31) < 31) bits &= (2u << (pr & 31)) - 1u;
    if (bits) last = 32 * w + 31 - __clz(bits);
  }
  s.lastgap = last;
}

// the k-mer at position lp0 + j (the run has been advanced j times); same outputs as tile_kmer
// HASH = false: the caller has no use for the k-mer hash (h is left alone; *owner is only set for the reference's
// partition function)
template <int NL, bool HASH = true>
__device__ __forceinline__ bool run_kmer(const KmerRun<NL> &s, int j, int lp0, int k, uint64_t (&rec)[NL], uint64_t &h, uint32_t rank_n = 1,
                                         uint32_t reference_owner = 0, uint32_t *owner = nullptr) {
  const uint32_t c = (uint32_t)(s.nextb >> 62);
  uint32_t le = ((s.okl >> j) & 1u) ? s.lc : KC_EXT_NONE;
  uint32_t re = ((s.okr >> j) & 1u) ? c : KC_EXT_NONE;
  if (owner) *owner = 0;
  if (owner && rank_n > 1 && reference_owner) *owner = kc_reference_owner<NL>(s.f, s.r, k, rank_n);
  uint64_t f[NL];
  const bool swap = kc_less<NL>(s.r, s.f);  // strict: a palindrome keeps the forward extensions
#pragma unroll
  for (int w = 0; w < NL; w++) f[w] = swap ? s.r[w] : s.f[w];
  const uint32_t nl = (re == KC_EXT_NONE) ? KC_EXT_NONE : 3u - re;
  const uint32_t nr = (le == KC_EXT_NONE) ? KC_EXT_NONE : 3u - le;
  le = swap ? nl : le;
  re = swap ? nr : re;
  if (HASH) {
    h = kc_hash<NL>(f);
    if (owner && rank_n > 1 && !reference_owner) *owner = kc_owner_of_hash(h, rank_n);
  }
#pragma unroll
  for (int w = 0; w < NL; w++) rec[w] = f[w];
  rec[NL - 1] |= (uint64_t)(le | (re << 3));
  return s.lastgap < lp0 + j;
}

// from position lp0 + j to lp0 + j + 1
template <int NL>
__device__ __forceinline__ void run_advance(KmerRun<NL> &s, int j, int lp0, int k) {
  const uint64_t c = s.nextb >> 62;
  s.nextb <<= 2;
  s.lc = (uint32_t)(s.f[0] >> 62);
  const int wi = (k - 1) >> 5, bit = 62 - 2 * ((k - 1) & 31);  // where the last base of a k-mer sits
#pragma unroll
  for (int w = 0; w < NL; w++) {
    s.f[w] = (s.f[w] << 2) | (w + 1 < NL ? s.f[w + 1] >> 62 : 0ULL);
    if (w == wi) s.f[w] |= c << bit;
  }
#pragma unroll
  for (int w = NL - 1; w >= 0; w--) {
    s.r[w] = (s.r[w] >> 2) | (w > 0 ? s.r[w - 1] << 62 : (3ULL - c) << 62);
    s.r[w] &= kc_word_mask(k, w);
  }
  if ((s.gin >> j) & 1u) s.lastgap = lp0 + j + k + 1;
}

template <int NL, int FMT>
__global__ __launch_bounds__(TPB) void kc_extract_kernel(ExtractArgs a, Table t, uint64_t *ctrs) {
  __shared__ TileLDS<TileSmall> L;
  const int64_t T0 = a.pos0 + (int64_t)blockIdx.x * TILE;
  stage_tile<FMT, TileSmall>(L, a, T0, ctrs, threadIdx.x, FMT != FMT_SEQBLOCK ? a.tile_first[blockIdx.x] : 0, true);
  const int64_t lo = a.align, hi = (int64_t)a.align + (int64_t)a.total;
  uint32_t n_ins = 0;
#pragma unroll 1
  for (int it = 0; it < PPT; it++) {
    const int off = it * TPB + threadIdx.x;
    const int64_t x = T0 + off;
    uint64_t rec[NL], h = 0;
    uint32_t owner = 0;
    bool valid = (x > lo) && (x + a.k < hi) && tile_kmer<NL>(L, PRE + off, a.k, rec, h, a.rank_n, a.reference_owner, &owner);
    if (valid && owner == a.rank_me) {
      table_insert<NL>(t, rec, ctrs);
      n_ins++;
    }
  }
  // wave-reduce then one atomic
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (lane_id() == 0 && n_ins) atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
}

template <int NL>
__global__ __launch_bounds__(TPB) void kc_insert_records_kernel(const uint64_t *recs, uint64_t n, Table t, uint64_t *ctrs,
                                                                 uint32_t count_inserted) {
  const uint64_t stride = (uint64_t)gridDim.x * TPB;
  uint32_t n_ins = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += stride) {
    uint64_t rec[NL];
#pragma unroll
    for (int j = 0; j < NL; j++) rec[j] = recs[i * NL + j];
    table_insert<NL>(t, rec, ctrs);
    n_ins++;
  }
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (count_inserted && lane_id() == 0 && n_ins) atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
}

// first read whose start lies at or after each tile's first position (lower bound on the offsets)
__global__ void kc_tile_first_kernel(const uint64_t *offsets, uint64_t nreads, uint32_t align, int64_t pos0, uint32_t span, uint64_t ntiles,
                                     uint64_t *tile_first) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= ntiles) return;
  int64_t T0 = pos0 + (int64_t)i * span - (int64_t)align;  // in offset coordinates
  uint64_t lo = 0, hi = nreads + 1;                            // offsets has nreads+1 entries
  while (lo < hi) {
    uint64_t mid = (lo + hi) >> 1;
    if ((int64_t)offsets[mid] < T0) lo = mid + 1; else hi = mid;
  }
  tile_first[i] = lo;
}

// raw k-mers of the block: sum over reads of max(0, len-k+1) (kcount.cpp:78,86)
__global__ void kc_read_stats_kernel(const uint64_t *offsets, uint64_t nreads, int k, uint64_t *ctrs, uint32_t count_expect) {
  uint64_t acc = 0, exp = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; r < nreads; r += stride) {
    uint64_t len = offsets[r + 1] - offsets[r];
    if (len >= (uint64_t)k) acc += len - k + 1;
    if (len >= (uint64_t)k + 2) exp += len - k - 1;
  }
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_down(acc, o);
    exp += __shfl_down(exp, o);
  }
  // one pair of bumps per workgroup: the sixteen thousand waves of a 50 M-read block bumping the two counters one by one
  // were most of this kernel's 0.44 ms (a returning or not, an atomic on one address is served about every ten ns)
  __shared__ unsigned long long s_acc, s_exp;
  if (threadIdx.x == 0) s_acc = s_exp = 0;
  __syncthreads();
  if (lane_id() == 0) {
    if (acc) atomicAdd(&s_acc, (unsigned long long)acc);
    if (exp) atomicAdd(&s_exp, (unsigned long long)exp);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_acc) atomicAdd((unsigned long long *)&ctrs[CTR_RAW_KMERS], s_acc);
    if (count_expect && s_exp) atomicAdd((unsigned long long *)&ctrs[CTR_EXPECT], s_exp);
  }
}

// the same for a '_'-joined block: count runs of non-separator bytes.  A wave owns SEQSTAT_SPAN consecutive bytes and
// takes them 64 at a time, one per lane: the separators of a step are a ballot, a run that ends in the step began behind
// the last separator below it or, failing one, as many bytes before the step as the run that reaches into it is long.
// (One thread per run end that walked back byte by byte took 8 ms per 268 MB: 150 dependent loads a run.)
constexpr uint64_t SEQSTAT_SPAN = 4096;
__global__ void kc_seqblock_stats_kernel(const uint8_t *seqs, uint64_t len, int k, uint64_t *ctrs, uint32_t count_expect) {
  const uint32_t lane = lane_id();
  const uint64_t nwaves = ((uint64_t)gridDim.x * blockDim.x) >> 6, wave0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint64_t acc = 0, exp = 0;
  for (uint64_t w = wave0; w * SEQSTAT_SPAN < len; w += nwaves) {
    const uint64_t lo = w * SEQSTAT_SPAN, hi = lo + SEQSTAT_SPAN < len ? lo + SEQSTAT_SPAN : len;
    // the run that reaches lo from before it
    uint64_t carry = 0;
    for (uint64_t q = lo; q > 0;) {
      const uint64_t b = q >= 64 ? q - 64 : 0, n = q - b;
      const bool sep = lane < n && seqs[b + lane] == '_';
      const uint64_t m = __ballot(sep);
      if (m) {
        carry += n - 1 - (uint64_t)(63 - __clzll((long long)m));
        break;
      }
      carry += n;
      q = b;
    }
    for (uint64_t at = lo; at < hi; at += 64) {
      const uint64_t pos = at + lane;
      const bool in = pos < hi;
      const uint8_t c = in ? seqs[pos] : (uint8_t)'_';
      const bool next_sep = in && (pos + 1 >= len || seqs[pos + 1] == '_');
      const uint64_t n = hi - at < 64 ? hi - at : 64;
      const uint64_t m = __ballot(in && c == '_');
      if (in && c != '_' && next_sep) {  // a run ends here
        const uint64_t below = m & ((1ULL << lane) - 1ULL);
        const uint64_t rl = below ? (uint64_t)lane - (uint64_t)(63 - __clzll((long long)below)) : (uint64_t)lane + 1 + carry;
        if (rl >= (uint64_t)k) acc += rl - k + 1;
        if (rl >= (uint64_t)k + 2) exp += rl - k - 1;
      }
      carry = m ? n - 1 - (uint64_t)(63 - __clzll((long long)m)) : carry + n;
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    acc += __shfl_down(acc, o);
    exp += __shfl_down(exp, o);
  }
  if (lane == 0 && acc) atomicAdd((unsigned long long *)&ctrs[CTR_RAW_KMERS], (unsigned long long)acc);
  if (count_expect && lane == 0 && exp) atomicAdd((unsigned long long *)&ctrs[CTR_EXPECT], (unsigned long long)exp);
}

// ---- finalize ----------------------------------------------------------------------------------
// S7: ExtCounts::get_ext (kcount_cpu.cpp:135-145,173-182).  Returns 0-3 = ACGT, 4 = 'X', 5 = 'F'.
__device__ __forceinline__ uint32_t vote_ext(const uint32_t (&c)[4], uint32_t count, int dmin_thres) {
  uint32_t top = 0;
#pragma unroll
  for (uint32_t i = 1; i < 4; i++)
    if (c[i] >= c[top]) top = i;  // ties go to the later letter
  uint32_t runner = 0;
#pragma unroll
  for (uint32_t i = 0; i < 4; i++)
    if (i != top && c[i] > runner) runner = c[i];
  // the reference's expression, in double, truncated toward zero: (int)((1.0 - DYN_MIN_DEPTH) * count)
  int dmin_dyn = (int)((1.0 - 0.9) * (double)count);
  if (dmin_dyn < dmin_thres) dmin_dyn = dmin_thres;
  if ((int)c[top] < dmin_dyn) return 4u;
  if ((int)runner >= dmin_dyn) return 5u;
  return top;
}

template <int NL>
__global__ __launch_bounds__(TPB) void kc_finalize_kernel(Table t, uint64_t capacity, int dmin_thres, uint64_t *out_keys,
                                                           uint16_t *out_counts, uint8_t *out_left, uint8_t *out_right,
                                                           uint64_t *ctrs) {
  const uint64_t stride = (uint64_t)gridDim.x * TPB;
  uint64_t purged = 0, sum = 0;
  const uint64_t nround = (capacity + stride - 1) / stride;
  for (uint64_t rnd = 0; rnd < nround; rnd++) {
    const uint64_t s = rnd * stride + (uint64_t)blockIdx.x * TPB + threadIdx.x;
    bool keep = false;
    uint32_t count = 0, l = 0, r = 0;
    if (s < capacity && t.keys[s * NL + NL - 1] != KEY_EMPTY) {
      const uint32_t *v = t.vals + s * 9;
      count = min(v[0], KC_COUNT_MAX);
      uint32_t lc[4], rc[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        lc[i] = min(v[1 + i], KC_COUNT_MAX);
        rc[i] = min(v[5 + i], KC_COUNT_MAX);
      }
      if (count >= 2) {  // S8
        l = vote_ext(lc, count, dmin_thres);
        r = vote_ext(rc, count, dmin_thres);
        keep = (l < 4u) && (r < 4u);
      }
      if (!keep) purged++;
    }
    const uint64_t m = __ballot(keep);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      uint64_t base = 0;
      if ((int)lane_id() == leader) base = atomicAdd((unsigned long long *)&ctrs[CTR_OUT], (unsigned long long)__popcll(m));
      base = __shfl(base, leader);
      if (keep) {
        const uint64_t o = base + (uint64_t)__popcll(m & ((1ULL << lane_id()) - 1ULL));
#pragma unroll
        for (int j = 0; j < NL; j++) out_keys[o * NL + j] = t.keys[s * NL + j];
        out_counts[o] = (uint16_t)count;
        out_left[o] = (uint8_t)("ACGT"[l]);
        out_right[o] = (uint8_t)("ACGT"[r]);
        sum += count;
      }
    }
  }
  for (int o = 32; o > 0; o >>= 1) {
    purged += __shfl_down(purged, o);
    sum += __shfl_down(sum, o);
  }
  if (lane_id() == 0) {
    if (purged) atomicAdd((unsigned long long *)&ctrs[CTR_PURGED], (unsigned long long)purged);
    if (sum) atomicAdd((unsigned long long *)&ctrs[CTR_SUM_COUNTS], (unsigned long long)sum);
  }
}

// every entry, unfiltered (tests of S5/S6)
template <int NL>
__global__ void kc_dump_kernel(Table t, uint64_t capacity, uint64_t *out_keys, uint16_t *out_counts, uint16_t *out_exts,
                               uint64_t *cursor) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= capacity || t.keys[s * NL + NL - 1] == KEY_EMPTY) return;
  const uint64_t o = atomicAdd((unsigned long long *)cursor, 1ULL);
  for (int j = 0; j < NL; j++) out_keys[o * NL + j] = t.keys[s * NL + j];
  const uint32_t *v = t.vals + s * 9;
  out_counts[o] = (uint16_t)min(v[0], KC_COUNT_MAX);
  for (int i = 0; i < 8; i++) out_exts[o * 8 + i] = (uint16_t)min(v[1 + i], KC_COUNT_MAX);
}

// move every entry of a full table into a bigger one (keys are unique, so values can be stored plainly)
template <int NL>
__global__ void kc_rehash_kernel(Table from, uint64_t from_capacity, Table to) {
  const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= from_capacity || from.keys[s * NL + NL - 1] == KEY_EMPTY) return;
  uint64_t key[NL];
  for (int j = 0; j < NL; j++) key[j] = from.keys[s * NL + j];
  uint64_t slot = kc_hash<NL>(key) & to.mask;
  for (;;) {
    unsigned long long *ks = (unsigned long long *)(to.keys + slot * NL);
    if (atomicCAS(&ks[NL - 1], (unsigned long long)KEY_EMPTY, (unsigned long long)key[NL - 1]) == KEY_EMPTY) {
      for (int j = 0; j < NL - 1; j++) ks[j] = key[j];
      break;
    }
    slot = (slot + 1) & to.mask;
  }
  for (int i = 0; i < 9; i++) to.vals[slot * 9 + i] = from.vals[s * 9 + i];
}

// ---- lookups over the results (KmerDHT::kmer_exists / get_kmer_count, src/kcount/kmer_dht.cpp:198-245) -------------
// index: open addressing over result numbers (slot = result index + 1, 0 = empty), built once after finalize
template <int NL>
__global__ void kc_index_build_kernel(const uint64_t *keys, uint64_t n, uint32_t *index, uint64_t mask) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t key[NL];
  for (int j = 0; j < NL; j++) key[j] = keys[i * NL + j];
  uint64_t s = kc_hash<NL>(key) & mask;
  while (atomicCAS(&index[s], 0u, (uint32_t)(i + 1)) != 0u) s = (s + 1) & mask;  // result keys are unique
}

// queries may be given in either orientation; counts[i] = 0 and left/right = 0 when the k-mer is not in the results
template <int NL>
__global__ void kc_lookup_kernel(const uint64_t *queries, uint64_t nq, int k, const uint32_t *index, uint64_t mask, const uint64_t *keys,
                                 const uint16_t *counts, const uint8_t *left, const uint8_t *right, uint16_t *out_counts,
                                 uint8_t *out_left, uint8_t *out_right) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nq) return;
  uint64_t f[NL], r[NL];
  for (int j = 0; j < NL; j++) f[j] = queries[i * NL + j] & kc_word_mask(k, j);
  kc_revcomp<NL>(f, k, r);
  if (kc_less<NL>(r, f))
    for (int j = 0; j < NL; j++) f[j] = r[j];
  uint64_t s = kc_hash<NL>(f) & mask;
  uint16_t c = 0;
  uint8_t l = 0, rr = 0;
  for (;;) {
    const uint32_t e = index[s];
    if (!e) break;
    bool same = true;
    for (int j = 0; j < NL; j++) same &= keys[(uint64_t)(e - 1) * NL + j] == f[j];
    if (same) {
      c = counts[e - 1];
      l = left[e - 1];
      rr = right[e - 1];
      break;
    }
    s = (s + 1) & mask;
  }
  out_counts[i] = c;
  if (out_left) out_left[i] = l;
  if (out_right) out_right[i] = rr;
}

// ---- synthetic reads ----------------------------------------------------------------------------
__global__ void kc_synth_kernel(const kc_synth_table *tab, uint64_t first_read, uint64_t nreads, uint8_t *bases,
                                uint8_t *quals, uint64_t *offsets) {
  const uint32_t L = tab->read_len;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  // one thread per 4 bases: a read's header (genome, start, strand) is cheap to recompute
  const uint64_t nquads = nreads * ((L + 3) / 4);
  for (uint64_t qd = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; qd < nquads; qd += stride) {
    const uint64_t r = qd / ((L + 3) / 4);
    const uint32_t i0 = (uint32_t)(qd % ((L + 3) / 4)) * 4;
    uint64_t rstate, start;
    uint32_t g;
    bool rev;
    kc_synth_read_header(tab, first_read + r, &rstate, &g, &start, &rev);
    for (uint32_t i = i0; i < i0 + 4 && i < L; i++) {
      uint8_t b, q;
      kc_synth_base(tab, rstate, g, start, rev, i, &b, &q);
      bases[r * L + i] = b;
      quals[r * L + i] = q;
    }
    if (i0 == 0) offsets[r] = r * L;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) offsets[nreads] = nreads * L;
}

}  // namespace kc
