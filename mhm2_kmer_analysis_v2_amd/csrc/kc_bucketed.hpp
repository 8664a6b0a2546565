// kc_bucketed.hpp -- the bucketed insert path: the HBM-bound core of libkcount_mi355.
//
// Random per-k-mer atomics on an HBM table move a 64-byte sector per 8-byte key and run at the
// fabric's scattered-atomic rate (the global-table path in kc_kernels.hpp measures ~5 G k-mers/s).
// Here every k-mer record instead makes two coalesced trips through HBM and is counted in LDS:
//
//   level 1  kc_l1_reads_kernel / kc_l1_records_kernel
//            reads (or records received from other shards) -> canonical k-mer records, multisplit in LDS
//            into P1 buckets; every persistent workgroup appends its runs to segments it alone owns
//            (writer g, bucket b), so no global atomics and no ordering hazards.
//   level 2  kc_l2_split_kernel
//            one workgroup per bucket streams the bucket's G segments, multisplits into P2 regions and
//            appends to region arrays it alone owns.
//   count    kc_count_kernel
//            one workgroup per region: an open-addressed (linear probe) table of S slots lives in LDS
//            -- the "probe window" of the region; records are streamed once, keys claimed by LDS
//            compare-and-swap, count / extension votes bumped by LDS atomics; then S7 vote, S8 purge
//            and a ballot/prefix compaction straight into the dense result arrays.
//
// Bucket, region and probe start are fields of the k-mer's 64-bit hash (hash_b1, hash_b2, hash_slot) -- or, for
// short one-word k-mers, bits of an invertible mix of the k-mer itself (Geom::cp, "compact records"): the region
// then implies part of the k-mer, level 2 stores the rest in 32 bits, and nobody hashes a record again.
//
// A region that does not fit (too many records for its array, or more distinct k-mers than LDS slots)
// is flagged and handled, whole, by the global-table kernels of kc_kernels.hpp, so every k-mer lives in
// exactly one structure and nothing is ever dropped.
//
// Replaces gpu_insert_supermer_block / gpu_insert_kmer / gpu_purge_invalid / gpu_compact_ht of the
// reference (src/kcount/kcount-gpu/gpu_hash_table.cpp:205-268,357-475) with CPU-backend semantics.
#pragma once
#include "kc_kernels.hpp"

namespace kc {

constexpr int WGB = 1024;        // threads per workgroup in this file (16 waves: one workgroup per CU)
// The reads kernels stage one "super-tile" per workgroup and step: exactly one sixteen-base group per thread (the
// staged span, PRE + SUPER_SPAN + POST, is 16 * WGB positions), k-mers starting at its SUPER_SPAN positions.
constexpr int SUPER_SPAN = 16 * WGB - PRE - POST;  // 16160
using TileSuper = TileGeo<WGB, SUPER_SPAN>;
static_assert(TileSuper::GPT == 1 && TileSuper::NGROUP == WGB, "one group per thread");
constexpr int PMAX = 1024;       // max fan-out of either level
constexpr int GMAX = 512;        // max level-1 writers

// records a thread holds per round: RPOS when they come from memory, RPOS_READS when each is cut out of a staged tile
// (more live registers)
template <int NL> struct Rnd {
  static constexpr int RPOS = NL == 1 ? 16 : NL == 2 ? 8 : 4;
  static constexpr int RPOS_READS = NL == 1 ? 8 : NL <= 3 ? 4 : 2;
  // sorted staging of one round (+ one slot per lane of a wave for the positions that hold no record):
  // <= 128 KiB from memory, <= 96 KiB from tiles (the tiles need LDS too)
  static constexpr size_t STAGE = ((size_t)WGB * RPOS + 64) * NL * 8;
  static constexpr size_t STAGE_READS = ((size_t)WGB * RPOS_READS + 64) * NL * 8;
};

// Destination arrays are chains of fixed-size chunks taken from an arena that only its owner allocates
// from (a level-1 writer; a level-2 bucket), so appends need no global atomics and a destination that
// grows far beyond the mean (k-mer counts are heavy-tailed) costs no memory up front.
struct Geom {
  uint32_t G, P1, P2, S;       // writers, fan-outs (any value up to PMAX), LDS slots per region (power of two)
  uint32_t log2CH1, log2CH2;   // records per chunk (powers of two)
  uint32_t L1MAX, L2MAX;       // longest chain of a (writer,bucket) segment / of a region, in chunks
  uint32_t A1;                 // chunks in each writer's arena
  uint32_t A2;                 // chunks in the level-2 arena (all buckets)
  // compact records (one-word k-mers with k <= KC_COMPACT_MAX_K, both fan-outs powers of two): the k-mer travels as
  // its invertible mix (kc_feistel_fwd), the top la bits of which are the bucket, the next lb the region's index in
  // the bucket and the low bits the probe start; level 2 then stores only what the region does not imply, in 32 bits
  uint32_t cp;                 // 0: records carry the k-mer itself, buckets come from its hash
  uint32_t la, lb;             // log2 P1, log2 P2
  uint32_t k2;                 // 2k: bits of the mixed k-mer
  uint32_t own_lo, own_hi;     // level-1 buckets this shard owns (every one unless the context is in the shard flow): ChainDest::own_lo
  // Level-1 records of SIX bytes (the kernels that stage the short form of compact records, kc_l1_reads16_kernel and
  // kc_l1_records16_kernel, write nothing else): the 32 bits of the mix below the bucket, then bucket | extension codes << 10
  // in 16 bits -- what the staging holds; the bucket is the chain's anyway.  A chunk is CH1 records of six bytes.  Two
  // records are one 12-byte store for the writer and one aligned 12-byte load for level 2 (kc_l2_rec6_kernel): a third more
  // records per second than 8-byte ones on either side (scripts/ubench_rec6.hip, profiles/r04_ubench_rec6.txt).
  uint32_t rec6;
#ifdef KC_ABLATE
  uint32_t abl;                // experiment builds (-DKC_ABLATE): which part of a kernel to leave out (scripts/ablate.py)
#endif
};
#ifdef KC_ABLATE
#define KC_ABL(x, k) ((x).abl == (k))
#else
#define KC_ABL(x, k) false
#endif

struct BucketBufs {
  uint64_t *rec1;      // [G][A1] chunks of CH1 records
  uint32_t *chain1;    // [G*P1][L1MAX] chunk index within the writer's arena
  uint32_t *cnt1;      // [G*P1] records in the chain
  uint32_t *used1;     // [2 * G] chunks taken from each writer's arena: [g] from the bottom, [G + g] from the top (ChainDest::own_lo)
  uint64_t *rec2;      // [A2] chunks of CH2 records
  uint32_t *chain2;    // [P1*P2][L2MAX] chunk index in rec2
  uint32_t *cnt2;      // [P1*P2] records in the chain
  uint32_t *base2;     // [P1+1] first chunk of each bucket's private part of the level-2 arena
  uint32_t *flag;      // [P1*P2]: 0 fine, 1 chain overflowed (level 2), 2 more distinct k-mers than slots (count)
  uint64_t *ovf1;      // level-1 overflow: records without a home segment
  uint64_t *ovf2;      // level-2 overflow: records of flagged regions
  uint64_t ovf1_cap, ovf2_cap;
  // level 2 in instalments (kc_l2_split_kernel<..., INC>): what an earlier instalment has already taken
  uint32_t *done1;     // [G*P1] records of the chain that have been to level 2
  uint32_t *used2;     // [P1] chunks taken from the bucket's part of the level-2 arena
};

// The shard flow (kc_shard.hpp): besides its own G chains a bucket may have flat sources -- dense runs of records
// received from other shards, one per received segment -- and a shard builds regions only for the buckets it owns.
constexpr uint32_t FLAT_MAX = 256;  // flat sources level 2 can walk per bucket (L2LDS::flo)
struct FlatSrc {
  const uint32_t *cnt;  // [F][nbo] records of bucket (b_lo + i) in flat source f
  const uint64_t *at;   // [F][nbo] device address of the first of them
  uint32_t F, nbo;      // flat sources so far; buckets this shard owns
  uint32_t b_lo, b_hi;  // ... which are these
};

enum {  // counters of this path, one u64 each
  CB_OVF1 = 0, CB_OVF2, CB_FATAL, CB_FLAGGED_RECS, CB_DUMP, CB_ENTRIES, CB_OUT_RESERVED, CB_COUNT = 16
};
// bits of cb[CB_FATAL]: records were lost, the pass that set one cannot be used
enum { FATAL_OVF1 = 1,    // level-1 overflow list full (the host bounds every launch by the list's free room: cannot happen)
       FATAL_ARENA2 = 2,  // level-2 arena smaller than the buffered records (cannot happen either)
       FATAL_OVF2 = 4 };  // level-2 overflow list full: the host makes it as large as cb[CB_OVF2] says and runs level 2 again


// Four independent fields of the 64-bit k-mer hash: bits 0-15 pick the level-1 bucket, 16-31 the level-2 bucket
// (each by multiply-shift, so the fan-outs need not be powers of two; 65536 / P >= 64 values per bucket keeps the
// imbalance under 1.6 %), bits 32-47 the slot inside the region table, and the top bits the owner shard
// (kc_owner_of_hash weighs bits 32-63 towards the top).
__device__ __forceinline__ uint32_t hash_b1(uint64_t h, const Geom &g) { return (((uint32_t)h & 0xFFFFu) * g.P1) >> 16; }
__device__ __forceinline__ uint32_t hash_b2(uint64_t h, const Geom &g) { return ((((uint32_t)h >> 16) & 0xFFFFu) * g.P2) >> 16; }
__device__ __forceinline__ uint32_t hash_slot(uint64_t h, uint32_t S) {  // S is a power of two <= 4096
  return (uint32_t)(h >> 32) & (S - 1u);
}

// compact records: rec = mix << (64 - k2) | extension codes
__device__ __forceinline__ uint64_t cp_mix_rec(uint64_t keyrec, const Geom &g) {  // k-mer record -> mixed record
  const uint32_t sh = 64u - g.k2;
  return (kc_feistel_fwd(keyrec >> sh, (int)(g.k2 >> 1)) << sh) | (keyrec & KC_EXT_MASK);
}
__device__ __forceinline__ uint64_t cp_unmix_rec(uint64_t mixrec, const Geom &g) {
  const uint32_t sh = 64u - g.k2;
  return (kc_feistel_inv(mixrec >> sh, (int)(g.k2 >> 1)) << sh) | (mixrec & KC_EXT_MASK);
}
__device__ __forceinline__ uint32_t cp_b1(uint64_t rec, const Geom &g) { return (uint32_t)(rec >> (64u - g.la)); }
__device__ __forceinline__ uint32_t cp_b2(uint64_t rec, const Geom &g) { return (uint32_t)(rec >> (64u - g.la - g.lb)) & (g.P2 - 1u); }
// what a region does not imply: the low bits of the mix, above the six extension bits
__device__ __forceinline__ uint32_t cp_pack32(uint64_t rec, const Geom &g) {
  const uint32_t rb = g.k2 - g.la - g.lb;
  return (((uint32_t)(rec >> (64u - g.k2)) & ((1u << rb) - 1u)) << 6) | ((uint32_t)rec & (uint32_t)KC_EXT_MASK);
}
// back to the k-mer record from a region and its 32-bit record
__device__ __forceinline__ uint64_t cp_unpack_rec(uint32_t rec32, size_t region, const Geom &g) {
  const uint32_t rb = g.k2 - g.la - g.lb;
  const uint64_t mix = ((uint64_t)region << rb) | (uint64_t)(rec32 >> 6);
  return (kc_feistel_inv(mix, (int)(g.k2 >> 1)) << (64u - g.k2)) | (uint64_t)(rec32 & (uint32_t)KC_EXT_MASK);
}

template <int NL>
__device__ __forceinline__ uint64_t rec_hash(const uint64_t (&rec)[NL]) {
  uint64_t key[NL];
#pragma unroll
  for (int j = 0; j < NL; j++) key[j] = rec[j];
  key[NL - 1] &= ~KC_EXT_MASK;
  return kc_hash<NL>(key);
}

// Barrier for data exchanged through LDS only.  __syncthreads() also drains the vector-memory counter,
// which would serialise the global loads and stores these kernels keep in flight across their LDS phases.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The thread id as a value the compiler cannot trace back to the id register.  Addresses that depend only on the thread
// (its word of the histogram, its entry of dst[], its chain) are otherwise computed once before the outermost loop, kept
// in registers for the whole kernel and, in kernels that use every register they may, SPILLED -- and a reload from scratch is
// a vector-memory load whose wait (vmcnt(0): the counter is in order) is a wait for every prefetch load and every copy-out
// store the wave has in flight.  Round 2's level-2 kernel reloaded two such addresses per round: one before writing
// dst[] and one right behind the next round's sixteen prefetch loads.  Recomputing them costs an instruction or two.
__device__ __forceinline__ int fresh_tid() {
  int t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}

// ---- workgroup exclusive scan over WGB values --------------------------------------------------
struct ScanLDS {
  uint32_t wsum[32];
  uint32_t total;
};

// Inclusive scan over the 64 lanes of a wave with data-parallel-primitive moves (no LDS traffic, unlike a shuffle,
// which goes through the LDS crossbar): four shifted adds scan each row of 16 lanes, two broadcasts carry the row totals
// on.  A lane whose source lies outside its row (or whose row a broadcast does not address) adds the zero given as `old`.
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xF, 0xF, false);  // row_shr:1
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xF, 0xF, false);  // row_shr:2
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xF, 0xF, false);  // row_shr:4
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xF, 0xF, false);  // row_shr:8
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xA, 0xF, false);  // row_bcast:15 into rows 1 and 3
  v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xC, 0xF, false);  // row_bcast:31 into rows 2 and 3
  return v;
}

// every thread of the workgroup calls this; returns the exclusive prefix of v, total in S.total (valid after the
// caller's next barrier).  One barrier inside: every wave scans the 16 wave totals itself instead of waiting for
// wave 0 to do it.
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, ScanLDS &S) {
  static_assert(WGB / 64 <= 16, "the wave totals must fit one row of 16 lanes");
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t incl = wave_incl_scan(v);
  if (lane_id() == 63) S.wsum[wave] = incl;
  lds_barrier();
  const uint32_t w = (lane_id() < WGB / 64) ? S.wsum[lane_id()] : 0u;
  uint32_t wi = w;  // the 16 totals sit in row 0: a row scan is enough
  wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x111, 0xF, 0xF, false);
  wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x112, 0xF, 0xF, false);
  wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x114, 0xF, 0xF, false);
  wi += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)wi, 0x118, 0xF, 0xF, false);
  const uint32_t wave_pre = (uint32_t)__builtin_amdgcn_readlane((int)(wi - w), wave);  // exclusive prefix of this wave's total
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)wi, WGB / 64 - 1);
  if (tid == 0) S.total = total;
  return incl - v + wave_pre;
}

// ---- multisplit of one round of records held in registers ---------------------------------------
// A round: every thread ranks its records inside their destinations with LDS atomics on a histogram (br[] = bucket |
// rank << 10); the histogram is scanned, the thread that owns a destination (thread b owns destination b) reserves room
// for the round's run at the end of that destination's chain; the records are scattered into an LDS staging sorted by
// destination and copied out run by run, so that HBM sees stores of consecutive records.  (Storing the records from
// the registers straight to their scattered places was tried: every lane's store is then a memory request of its own,
// and both split kernels took 1.3-1.8x as long.)
// No LDS read sits inside a branch in the scatter and the copy-out: a read inside a conditional block is waited for at
// the end of that block, which would put the reads of a thread's records one behind the other.
struct SplitLDS {
  // per-bucket counts of this round (double-buffered); behind each, one word per lane for the positions that hold no
  // record (hist_rank)
  uint32_t hist[2][PMAX + 64];
  // what scatter and copy-out need to know about a destination's run, in one 16-byte read.  i = an element's index in
  // the sorted staging, s = where the run starts there (exclusive prefix):
  //   x, y = record index (relative to the owner's part of the arena) of element i, minus i, as seen from the chain's
  //          old last chunk and from the round's new chunks (the new chunks of a round are consecutive, so inside
  //          either the index is linear in i);  z = s | how many of the run's records found room << 16;
  //   w = how many of those still go into the old last chunk
  uint4 dst[PMAX];
  uint32_t arena_used;     // chunks taken from the bottom of the owner's arena
  uint32_t arena_top;      // ... from its top (shard flow: the chains that leave with every block, ChainDest::own_lo)
  ScanLDS scan;
};

// Rank of a record inside its destination's run of this round: one LDS add, all of a thread's adds in flight together and
// no branch around them.  A position without a record bumps a word of its own lane's behind the histogram -- chosen by
// INDEX: a select between two LDS pointers is compiled into a branch around either address computation (seven vector and
// five scalar instructions per record in round 2's kernels), a select between two indices is one v_cndmask.
__device__ __forceinline__ uint32_t hist_rank(SplitLDS &L, int buf, uint32_t b, bool valid) {
  return atomicAdd(&L.hist[buf][valid ? b : (uint32_t)PMAX + lane_id()], 1u);
}

// where the destinations of the current owner live
#ifdef KC_STAMPS
#define KC_SPLIT_STAMP(k)                                                                 \
  if (threadIdx.x == 0 && D.stamps) {                                                     \
    const unsigned long long tn_ = __builtin_amdgcn_s_memtime();                          \
    atomicAdd((unsigned long long *)&D.stamps[k], tn_ - *D.tprev);                        \
    *D.tprev = tn_;                                                                       \
  }
#else
#define KC_SPLIT_STAMP(k)
#endif

struct ChainDest {
#ifdef KC_STAMPS
  uint64_t *stamps;                 // diagnostic builds: cb + 8
  unsigned long long *tprev;        // thread 0's last stamp
#endif
  uint64_t *arena;      // chunk id c starts at record (c << log2CH) of the arena
  uint32_t *chain;      // destination b's chain: chain[b*LMAX + i]
  uint32_t log2CH, LMAX;
  uint32_t arena_cap;   // chunks this owner may take
  uint32_t arena_base;  // id of the owner's first chunk
  // Destinations [own_lo, own_hi) take their chunks from the bottom of the arena, the others from its top.  The shard
  // flow (kc_shard.hpp) empties every chain of a bucket another shard owns after every block: those chains live at the
  // top, and giving the top back whole (kc_shard_release_kernel) recycles their chunks, the partly filled last ones
  // included -- a bump allocator that only grew lost about half a chunk per foreign chain and block.
  uint32_t own_lo, own_hi;
#ifdef KC_ABLATE
  uint32_t abl;         // experiment builds: see Geom::abl
  uint32_t abl_a;       // experiment builds: records per 64 bytes of this destination
#endif
};

// state of destination `tid`, kept in thread tid's registers (only that thread ever touches it)
struct ChainState {
  uint32_t cur;   // records in the chain
  uint32_t last;  // id of its last chunk (valid when cur is not chunk-aligned)
};

// A round has two halves.  split_stage: scan, reserve, scatter to LDS; the caller has already bumped hist[buf] with LDS
// atomics (bucket | rank<<10 in br[], ~0 for "no record") and hit a barrier; returns the number of staged records.
// split_copy_out: the staged records go to HBM.
// Between the two a kernel takes delivery of what it prefetched for its next round and sends the next prefetch off:
// the wait for a load is a wait for every older store of the wave too (one in-order counter, and the compiler, which
// cannot count the stores of a loop, waits for all of them), so a wait at the top of a round would expose the latency
// of the copy-out stores just issued; placed here the stores it meets are a whole round old.
//   bucket_of(rec) gives a staged record's bucket during copy-out where that is cheap (sbucket == nullptr); otherwise
//   sbucket, an LDS array parallel to `sorted`, remembers it; overflow(b, rec) takes what found no room;
//   store(i, rec) writes a record to position i (in records) of the destination arena
//   rec_of(j, out) hands over record j of the thread: a kernel may keep its records in a shorter form in its registers
//   (the level-1 record of a compact k-mer is 32 bits + the bucket and extension codes that ride in br[], RS = 16:
//   br = bucket | extension codes << 10 | rank << 16)
template <int NL, int R, int RS = 10, class RecFn>
__device__ __forceinline__ uint32_t split_stage(SplitLDS &L, uint64_t *sorted, uint16_t *sbucket, int buf, uint32_t P, RecFn rec_of,
                                                const uint32_t (&br)[R], const ChainDest &D, ChainState &st) {
  static_assert(R * WGB <= 32768, "positions in the staging must fit 16 bits");
  static_assert(RS == 10 || R * WGB <= 65536, "ranks must fit the bits above RS");
  const int tid = fresh_tid();
  KC_SPLIT_STAMP(1)  // barrier after the histogram
  const uint32_t v = ((uint32_t)tid < P) ? L.hist[buf][tid] : 0u;
  const uint32_t excl = block_excl_scan(v, L.scan);
  const uint32_t CHm = (1u << D.log2CH) - 1u;
  if ((uint32_t)tid < P) {
    const uint32_t base = st.cur;
    uint32_t fit = v;
    const uint64_t room = ((uint64_t)D.LMAX << D.log2CH) - base;  // the chain holds at most LMAX chunks
    if ((uint64_t)fit > room) fit = (uint32_t)room;
    const uint32_t have = (base + CHm) >> D.log2CH;
    uint32_t k = ((base + fit + CHm) >> D.log2CH) - have, a = 0;
    if (k) {
      const bool own = (uint32_t)tid >= D.own_lo && (uint32_t)tid < D.own_hi;
      if (own) {
        a = atomicAdd(&L.arena_used, k);
        const uint32_t left = D.arena_cap - min(D.arena_cap, L.arena_top);  // (the top only moves between launches of a shard's blocks)
        if (a + k > left) {  // arena exhausted: use what is left of it, the rest overflows
          k = a < left ? left - a : 0;
          const uint64_t cap = ((uint64_t)(have + k) << D.log2CH) - base;
          if ((uint64_t)fit > cap) fit = (uint32_t)cap;
        }
      } else {
        // k consecutive chunks below what the top has taken so far; the bottom is whatever the launch started with
        // plus what it takes meanwhile: the two ends may only meet in a launch that is about to overflow anyway, and
        // then both sides stop at the other's starting point of this launch
        const uint32_t t = atomicAdd(&L.arena_top, k);
        const uint32_t left = D.arena_cap - min(D.arena_cap, L.arena_used);
        if (t + k > left) {
          k = 0;
          const uint64_t cap = ((uint64_t)have << D.log2CH) - base;
          if ((uint64_t)fit > cap) fit = (uint32_t)cap;
        } else {
          a = D.arena_cap - t - k;
        }
      }
      uint32_t *ch = D.chain + (size_t)tid * D.LMAX + have;
      for (uint32_t i = 0; i < k; i++) ch[i] = D.arena_base + a + i;
    }
    // element j of the run sits at chain position base + j: in the old last chunk (id st.last) while that has room,
    // then in the new chunks a, a+1, ... whose first one holds chain positions have << log2CH onwards
    uint4 d;
    d.x = ((st.last - D.arena_base - (base >> D.log2CH)) << D.log2CH) + base - excl;
    d.y = ((a - have) << D.log2CH) + base - excl;
    d.z = excl | (fit << 16);
    d.w = (base & CHm) ? min(fit, (CHm + 1u) - (base & CHm)) : 0u;
    L.dst[tid] = d;
    if (k) st.last = D.arena_base + a + k - 1;
    st.cur = base + fit;
    // the scatter takes a run's start from the histogram's word, which has served its purpose: a dense array of words
    // spreads over all the banks, the .z of 16-byte entries over a quarter of them
    L.hist[buf][tid] = excl;
    L.hist[buf ^ 1][tid] = 0;  // next round's histogram
  }
  lds_barrier();
  KC_SPLIT_STAMP(2)  // scan + reserve
  const uint32_t total = L.scan.total;
  if (!KC_ABL(D, 3)) {
    // all the run starts first, then all the writes (eight records at a time: sixteen starts in flight cost the level-2
    // kernel registers it does not have); a position without a record goes to a slot of its lane's behind the staging
    constexpr int H = R < 8 ? R : 8;
#pragma unroll
    for (int j0 = 0; j0 < R; j0 += H) {
      uint32_t pos[H];
#pragma unroll
      for (int j = 0; j < H; j++) pos[j] = L.hist[buf][br[j0 + j] & (PMAX - 1)];
#pragma unroll
      for (int j = 0; j < H; j++) {
        const uint32_t bj = br[j0 + j];
        const uint32_t p = bj != ~0u ? pos[j] + (bj >> RS) : (uint32_t)(R * WGB) + lane_id();
        uint64_t r[NL];
        rec_of(j0 + j, r);
#pragma unroll
        for (int w = 0; w < NL; w++) sorted[(size_t)p * NL + w] = r[w];
        if (sbucket) sbucket[p] = (uint16_t)(bj & (PMAX - 1));
      }
    }
  }
  lds_barrier();
  KC_SPLIT_STAMP(3)  // scatter to LDS
  return total;
}

template <int NL, class BucketFn, class OvfFn, class StoreFn>
__device__ __forceinline__ void split_copy_out(SplitLDS &L, const uint64_t *sorted, const uint16_t *sbucket, uint32_t total, const ChainDest &D,
                                               BucketFn bucket_of, OvfFn overflow, StoreFn store) {
  const int tid = fresh_tid();
  // copy out, U elements per thread and trip: first all their records, then all their destinations, then the stores
  constexpr int U = NL == 1 ? 4 : 2;
  const size_t arena0 = (size_t)D.arena_base << D.log2CH;
  if (KC_ABL(D, 2) || KC_ABL(D, 3)) return;
  for (uint32_t i0 = tid; i0 < total; i0 += U * WGB) {
    uint64_t r[U][NL];
    uint32_t b[U];
    uint4 d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = i0 + u * WGB;
      const uint32_t ii = i < total ? i : i0;  // in range: i0 < total
#pragma unroll
      for (int w = 0; w < NL; w++) r[u][w] = sorted[(size_t)ii * NL + w];
      if (sbucket) b[u] = sbucket[ii];
    }
    if (!sbucket) {
#pragma unroll
      for (int u = 0; u < U; u++) b[u] = bucket_of(r[u]);
    }
#pragma unroll
    for (int u = 0; u < U; u++) d[u] = L.dst[b[u]];
    bool spill = false;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = i0 + u * WGB;
      const uint32_t j = i - (d[u].z & 0xFFFFu);
      const bool live = i < total, fits = j < (d[u].z >> 16);
      // x and y are "index minus staging position" modulo 2^32: the sum must wrap in 32 bits before it is widened
      const uint32_t at = (j < d[u].w ? d[u].x : d[u].y) + i;
#ifdef KC_ABLATE
      if (KC_ABL(D, 4)) {  // timing only (wrong results): every run starts on a 64-byte boundary and is whole 64-byte blocks long
        const uint32_t am = D.abl_a - 1u, nfit = ((d[u].z >> 16) + (D.abl_a >> 1)) & ~am;
        const uint32_t at2 = at - ((at - j) & am);
        if (live && j < nfit) store(arena0 + at2, r[u]);
      } else
#endif
      if (live && fits && !KC_ABL(D, 1)) store(arena0 + at, r[u]);
      spill |= live && !fits;
    }
    if (__any(spill)) {  // rare: a chain or the arena is full
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = i0 + u * WGB;
        if (i < total && i - (d[u].z & 0xFFFFu) >= (d[u].z >> 16)) overflow(b[u], r[u]);
      }
    }
  }
  KC_SPLIT_STAMP(4)  // copy-out
  // no barrier here: the next round only touches the other histogram buffer and registers until its own first
  // barrier, which every wave reaches after finishing this copy-out
}

// the plain store of split_round: NL words per record
template <int NL>
__device__ __forceinline__ void store_words(uint64_t *arena, size_t i, const uint64_t (&r)[NL]) {
#pragma unroll
  for (int w = 0; w < NL; w++) arena[i * NL + w] = r[w];
}

// load the persistent state of this owner's P chains (before its first round)
__device__ __forceinline__ ChainState split_load_state(SplitLDS &L, uint32_t P, const uint32_t *cnt, const uint32_t *chain,
                                                       uint32_t LMAX, uint32_t log2CH, uint32_t used, uint32_t top = 0) {
  const int tid = threadIdx.x;
  ChainState st;
  st.cur = 0;
  st.last = 0;
  if ((uint32_t)tid < P) {
    const uint32_t c = cnt ? cnt[tid] : 0u;
    st.cur = c;
    if (c & ((1u << log2CH) - 1u)) st.last = chain[(size_t)tid * LMAX + (c >> log2CH)];
    L.hist[0][tid] = 0;
    L.hist[1][tid] = 0;
  }
  if (tid == 0) {
    L.arena_used = used;
    L.arena_top = top;
  }
  return st;
}

// ---- how fast does this arena take level 1's write pattern? -------------------------------------------------------------
// Every workgroup appends 64-byte runs round-robin to 1024 open chunks of its own part of the arena, like a writer of
// level 1 -- its stores without the rest of it (pick_fast_arena times it).  The window of open chunks jumps through the
// workgroup's whole part (32 places, eight rounds at each): an arena can be slow in places.
__global__ __launch_bounds__(WGB) void kc_arena_probe_kernel(uint64_t *arena, size_t words_per_wg, uint32_t rounds) {
  uint64_t *mine = arena + (size_t)blockIdx.x * words_per_wg;
  const uint32_t t = threadIdx.x, b = t >> 3, w = t & 7u;  // eight lanes write one 64-byte run
  const size_t chunks = words_per_wg / 512u;               // 512-record chunks in this part
  const size_t hop = chunks > 1024u ? (chunks - 1024u) / 31u : 0u;
  for (uint32_t r = 0; r < rounds; r++) {
    const size_t first = (size_t)((r >> 3) & 31u) * hop;   // where the window of 1024 open chunks starts
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const size_t chunk = first + b + 128u * j;           // 1024 chunks per round
      const size_t at = chunk * 512u + (size_t)(r & 7u) * 8u + w;
      if (at < words_per_wg) mine[at] = (uint64_t)r;
    }
  }
}

// ---- level 1 from reads ---------------------------------------------------------------------------
struct L1LDS {
  TileLDS<TileSuper> tile;
  SplitLDS sp;
};

template <int NL>
__device__ __forceinline__ ChainDest l1_dest(const Geom &gm, const BucketBufs &bb, uint32_t g) {
  ChainDest D;
#ifdef KC_STAMPS
  D.stamps = nullptr;
  D.tprev = nullptr;
#endif
  D.arena = bb.rec1 + (((size_t)g * gm.A1) << gm.log2CH1) * NL;
  D.chain = bb.chain1 + (size_t)g * gm.P1 * gm.L1MAX;
  D.log2CH = gm.log2CH1;
  D.LMAX = gm.L1MAX;
  D.arena_cap = gm.A1;
  D.arena_base = 0;
  D.own_lo = gm.own_lo;
  D.own_hi = gm.own_hi;
#ifdef KC_ABLATE
  D.abl = gm.abl;
  D.abl_a = 8 / NL;
#endif
  return D;
}

// the same for a geometry of six-byte level-1 records (Geom::rec6): the writer's part starts at that many BYTES
__device__ __forceinline__ ChainDest l1_dest6(const Geom &gm, const BucketBufs &bb, uint32_t g) {
  ChainDest D = l1_dest<1>(gm, bb, g);
  D.arena = reinterpret_cast<uint64_t *>(reinterpret_cast<uint8_t *>(bb.rec1) + (((size_t)g * gm.A1) << gm.log2CH1) * 6);
  return D;
}

// the overflow lists always hold k-mer records (the global-table kernels read them), never mixed ones
template <int NL, bool CP>
__device__ __forceinline__ void l1_overflow(const Geom &gm, const BucketBufs &bb, uint64_t *cb, const uint64_t (&r)[NL]) {
  const uint64_t o = atomicAdd((unsigned long long *)&cb[CB_OVF1], 1ULL);
  if (o < bb.ovf1_cap) {
#pragma unroll
    for (int w = 0; w < NL; w++) bb.ovf1[o * NL + w] = (CP && w == 0) ? cp_unmix_rec(r[w], gm) : r[w];
  } else {
    atomicOr((unsigned long long *)&cb[CB_FATAL], (unsigned long long)FATAL_OVF1);
  }
}

// ---- the run of a thread when k is known at compile time (compact records) -------------------------------------------
// A thread's RPOS consecutive k-mers and their neighbours are RPOS + K + 1 <= 32 bases: one 64-bit window Wn of the
// staged tile (first base highest), its reverse complement Rn, one 32-bit window each of the "usable as extension" bits
// and of the read-boundary bits.  K-mer j with its two neighbours is then a (K+2)-base slice of Wn, and the same k-mer
// read from the other strand -- neighbours complemented and swapped, exactly what S4/S5 ask for when the reverse
// complement is the canonical form -- is the mirrored slice of Rn: every shift is a constant, nothing is rolled.
// Outputs, in the short register form of split_stage: lo[j] = the low 32 bits of the mixed k-mer (with the bucket, its
// top la >= 2K - 32 bits, that is all of it), bk[j] = bucket | extension codes << 10, or ~0 where there is no k-mer
// (a code >= 4 means "none"; here a missing extension keeps the base in its low bits, which every consumer ignores).
// (KT = 0: k is the launch's, a.k <= 23 -- the same instructions with their shift counts and masks in scalar registers; the
// instantiation for MHM2's k = 21 has them as constants)
template <int KT, int RPOS, bool SH, class TL>
__device__ __forceinline__ void cp_run_fixed(const TL &L, int lp0, bool active, const Geom &gm, const ExtractArgs &a, uint32_t (&lo)[RPOS],
                                             uint32_t (&bk)[RPOS]) {
  const int K = KT ? KT : (int)a.k;
  const int NB = RPOS + K + 1;
  static_assert(RPOS + KT + 1 <= 32 && RPOS % 8 == 0 && PRE % 8 == 0, "the window must fit one word and never start word-aligned");
  const uint64_t *W = reinterpret_cast<const uint64_t *>(L.codes);
  const uint32_t *OK = reinterpret_cast<const uint32_t *>(L.ok);
  const int p = lp0 - 1, q = p >> 5, s = p & 31;  // lp0 is a multiple of 8: s is 7, 15, 23 or 31, never 0
  const uint64_t w0 = W[q], w1 = W[q + 1];
  const uint32_t ok0 = OK[q], ok1 = OK[q + 1], g0 = L.gap[q], g1 = L.gap[q + 1];
  const uint64_t Wn = (w0 << (2 * s)) | (w1 >> (64 - 2 * s));
  const uint32_t nok = ~__builtin_amdgcn_alignbit(ok1, ok0, (uint32_t)s);  // bit i: base i of the window is no extension
  const uint32_t gapw = __builtin_amdgcn_alignbit(g1, g0, (uint32_t)s);    // bit i: read boundary before base i
  const uint64_t Rn = kc_rc_word(Wn) << (64 - 2 * NB);
  const uint64_t MID = ((1ULL << (2 * K)) - 1ULL) << 2;  // the k-mer inside a slice
  const uint32_t MK = (1u << K) - 1u;
#pragma unroll
  for (int j = 0; j < RPOS; j++) {
    const uint64_t x = Wn >> (64 - 2 * (j + K + 2)), y = Rn >> (64 - 2 * (RPOS - 1 - j + K + 2));
    // strict: a palindrome keeps the forward extensions.  An odd k has no palindromes: the k-mers always differ, the right
    // neighbours below them never decide, and only the bases above the slice need masking (one AND per side, not two)
    const uint64_t CMP = (K & 1) ? (MID | 3ULL) : MID;
    const bool swap = (y & CMP) < (x & CMP);
    const uint64_t sel = swap ? y : x;
    // extension codes: the slice's outer bases, bit 2 set where the base may not serve as one
    const uint32_t na = (nok >> j) & 1u, nb = (nok >> (j + K + 1)) & 1u;
    const uint32_t le = ((uint32_t)(sel >> (2 * K + 2)) & 3u) | ((swap ? nb : na) << 2);
    const uint32_t re = ((uint32_t)sel & 3u) | ((swap ? na : nb) << 2);
    // the invertible mix of the k-mer (kc_feistel_fwd with constant k)
    uint32_t Lh = (uint32_t)(sel >> (K + 2)) & MK, Rh = (uint32_t)(sel >> 2) & MK;
    bool valid = active && ((gapw >> (j + 1)) & ((1u << (K + 1)) - 1u)) == 0u;
    if (SH) {
      uint64_t key[1] = {(((uint64_t)Lh << K) | Rh) << (64 - 2 * K)};
      const uint32_t owner = a.reference_owner ? 0u : kc_owner_of_hash(kc_hash<1>(key), a.rank_n);
      if (a.reference_owner) {
        uint64_t rk[1];
        kc_revcomp<1>(key, K, rk);
        valid = valid && kc_reference_owner<1>(key, rk, K, a.rank_n) == a.rank_me;
      } else {
        valid = valid && owner == a.rank_me;
      }
    }
    Lh ^= kc_feistel_f(Rh, 0, K);
    Rh ^= kc_feistel_f(Lh, 1, K);
    Lh ^= kc_feistel_f(Rh, 2, K);
    const uint64_t mix = ((uint64_t)Lh << K) | Rh;
    lo[j] = (uint32_t)mix;
    bk[j] = valid ? ((uint32_t)(mix >> (2 * K - gm.la)) | (le << 10) | (re << 13)) : ~0u;
  }
}

// The same cut for a consumer that wants the k-mer records themselves (the sender side of the records flow): rec[j] = the
// canonical k-mer, first base highest, with the extension codes in its low six bits (KC_EXT_MASK), h[j] = kc_hash of the
// bare k-mer, ok bit j = the window of k-mer j lies inside one read.
template <int K, int RPOS, class TL>
__device__ __forceinline__ uint32_t cut_run_fixed(const TL &L, int lp0, bool active, uint64_t (&rec)[RPOS], uint64_t (&h)[RPOS]) {
  constexpr int NB = RPOS + K + 1;
  static_assert(NB <= 32 && RPOS % 8 == 0 && PRE % 8 == 0, "the window must fit one word and never start word-aligned");
  const uint64_t *W = reinterpret_cast<const uint64_t *>(L.codes);
  const uint32_t *OK = reinterpret_cast<const uint32_t *>(L.ok);
  const int p = lp0 - 1, q = p >> 5, s = p & 31;
  const uint64_t w0 = W[q], w1 = W[q + 1];
  const uint32_t ok0 = OK[q], ok1 = OK[q + 1], g0 = L.gap[q], g1 = L.gap[q + 1];
  const uint64_t Wn = (w0 << (2 * s)) | (w1 >> (64 - 2 * s));
  const uint32_t nok = ~__builtin_amdgcn_alignbit(ok1, ok0, (uint32_t)s);
  const uint32_t gapw = __builtin_amdgcn_alignbit(g1, g0, (uint32_t)s);
  const uint64_t Rn = kc_rc_word(Wn) << (64 - 2 * NB);
  constexpr uint64_t MID = ((1ULL << (2 * K)) - 1ULL) << 2;
  constexpr uint64_t CMP = (K & 1) ? (MID | 3ULL) : MID;
  uint32_t okm = 0;
#pragma unroll
  for (int j = 0; j < RPOS; j++) {
    const uint64_t x = Wn >> (64 - 2 * (j + K + 2)), y = Rn >> (64 - 2 * (RPOS - 1 - j + K + 2));
    const bool swap = (y & CMP) < (x & CMP);
    const uint64_t sel = swap ? y : x;
    const uint32_t na = (nok >> j) & 1u, nb = (nok >> (j + K + 1)) & 1u;
    const uint32_t le = ((uint32_t)(sel >> (2 * K + 2)) & 3u) | ((swap ? nb : na) << 2);
    const uint32_t re = ((uint32_t)sel & 3u) | ((swap ? na : nb) << 2);
    const uint64_t key = (sel >> 2) << (64 - 2 * K);  // (the left neighbour above the k-mer leaves at the top)
    const uint64_t kk[1] = {key};
    h[j] = kc_hash<1>(kk);
    // (a code >= 4 means "none": the records that travel carry exactly 4 then, like the general cut's)
    rec[j] = key | (uint64_t)((le & 4u) ? 4u : le) | ((uint64_t)((re & 4u) ? 4u : re) << 3);
    const bool valid = active && ((gapw >> (j + 1)) & ((1u << (K + 1)) - 1u)) == 0u;
    okm |= valid ? (1u << j) : 0u;
  }
  return okm;
}

// SH: the context is one shard of several and keeps only the k-mers it owns (kc_submit_reads with rank_n > 1); the
// single-shard instantiation carries none of the ownership code.
// KK: k when the instantiation is made for one k (compact records only: cp_run_fixed), 0 = any k
template <int NL, int FMT, bool CP, bool SH, int KK>
__global__ __launch_bounds__(WGB) void kc_l1_reads_kernel(ExtractArgs a, Geom gm, BucketBufs bb, uint64_t nsuper, uint32_t rot,
                                                          uint64_t *ctrs, uint64_t *cb) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1LDS &L = *reinterpret_cast<L1LDS *>(smem);
  uint64_t *sorted = reinterpret_cast<uint64_t *>(smem + ((sizeof(L1LDS) + 15) & ~size_t(15)));
  // wide records: the bucket of a staged record is remembered beside it (compact ones carry it in their top bits)
  uint16_t *sbucket = CP ? nullptr : reinterpret_cast<uint16_t *>(smem + ((sizeof(L1LDS) + 15) & ~size_t(15)) + Rnd<NL>::STAGE_READS);
  constexpr int RPOS = Rnd<NL>::RPOS_READS;
  const int tid = threadIdx.x;
  // writer id: launches rotate their first writer (rot) so that many small submits still spread evenly
  const uint32_t g = (blockIdx.x + rot) % gm.G, P1 = gm.P1;
  ChainDest D = l1_dest<NL>(gm, bb, g);
#ifdef KC_STAMPS
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
  D.stamps = cb + 8;
  D.tprev = &tprev_;
#endif
  ChainState cst = split_load_state(L.sp, P1, bb.cnt1 + (size_t)g * P1, D.chain, D.LMAX, D.log2CH, bb.used1[g], bb.used1[gm.G + g]);
  __syncthreads();
  uint32_t n_ins = 0;
  int buf = 0;
  // the bytes of the next super-tile are on their way while this one is processed; the first read of a super-tile is
  // looked up one step further ahead than the bytes, which need it for their addresses
  TileRaw<TileSuper> raw;
  auto first_of = [&](uint64_t st) -> uint64_t { return (FMT != FMT_SEQBLOCK && st < nsuper) ? a.tile_first[st] : 0; };
  tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, tid, first_of(blockIdx.x), blockIdx.x < nsuper);
  uint64_t next_first = first_of((uint64_t)blockIdx.x + gridDim.x);
  constexpr int RUNS = SUPER_SPAN / RPOS;  // runs of RPOS consecutive positions in a super-tile
  static_assert(SUPER_SPAN % RPOS == 0, "runs must tile the span");
  constexpr int NROUND = (RUNS + WGB - 1) / WGB;
  // Iteration -1 only stages the first super-tile; every further one is staged in the middle of its predecessor's last
  // round (one place in the code for the staging, one for the rounds: inlined twice they cost registers)
  for (int64_t it = -1;; it++) {
    const uint64_t st = (uint64_t)blockIdx.x + (uint64_t)(it < 0 ? 0 : it) * gridDim.x;
    const bool work = it >= 0;  // the same for every thread of the workgroup
    if (work && st >= nsuper) break;
#pragma unroll 1
    for (int round = 0; round < NROUND; round++) {
      constexpr bool SHORT = KK != 0;  // records in the short register form
      uint64_t rec[SHORT ? 1 : RPOS][NL];
      uint32_t lo[SHORT ? RPOS : 1];
      uint32_t br[RPOS];
      uint32_t total = 0;
      if (work) {
      // each thread walks RPOS consecutive positions
      const int run_id = round * WGB + tid;
      const bool active = run_id < RUNS;
      const int lp0 = PRE + (active ? run_id : 0) * RPOS;
      if constexpr (KK != 0) {
        static_assert(NL == 1 && CP, "a fixed k goes with compact records");
        cp_run_fixed<KK, RPOS, SH>(L.tile, lp0, active, gm, a, lo, br);
      } else {
        KmerRun<NL> run;
        run_begin<NL>(run, L.tile, lp0, a.k);
#pragma unroll
        for (int j = 0; j < RPOS; j++) {
          uint64_t h = 0;
          uint32_t owner = 0;
          bool valid = run_kmer<NL, !CP>(run, j, lp0, a.k, rec[j], h, SH ? a.rank_n : 1u, SH ? a.reference_owner : 0u, SH ? &owner : nullptr) && active;
          if (SH) {
            if (CP && !a.reference_owner) {
              uint64_t key[NL];
#pragma unroll
              for (int w = 0; w < NL; w++) key[w] = rec[j][w];
              key[NL - 1] &= ~KC_EXT_MASK;
              owner = kc_owner_of_hash(kc_hash<NL>(key), a.rank_n);
            }
            valid = valid && owner == a.rank_me;
          }
          if (CP) rec[j][0] = cp_mix_rec(rec[j][0], gm);
          br[j] = valid ? (CP ? cp_b1(rec[j][0], gm) : hash_b1(h, gm)) : ~0u;
          if (j + 1 < RPOS) run_advance<NL>(run, j, lp0, a.k);
        }
      }
      // the ranks: one LDS add per record, all of a thread's adds in flight together (no branch around them: a
      // position without a k-mer bumps a word of its own lane's instead)
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        const bool valid = br[j] != ~0u;
        const uint32_t rank = hist_rank(L.sp, buf, br[j] & (PMAX - 1), valid);
        br[j] = valid ? (br[j] | (rank << (SHORT ? 16 : 10))) : ~0u;
        n_ins += valid ? 1u : 0u;
      }
      KC_SPLIT_STAMP(0)  // cut the k-mers out of the super-tile, histogram
      lds_barrier();
      total = split_stage<NL, RPOS, SHORT ? 16 : 10>(
          L.sp, sorted, sbucket, buf, P1,
          [&](int j, uint64_t (&o)[NL]) {
            if constexpr (SHORT) {
              o[0] = ((uint64_t)(br[j] & (PMAX - 1)) << (64u - gm.la)) | ((uint64_t)lo[j] << (64u - gm.k2)) | (uint64_t)((br[j] >> 10) & 63u);
            } else {
#pragma unroll
              for (int w = 0; w < NL; w++) o[w] = rec[j][w];
            }
          },
          br, D, cst);
      }
      if (round == NROUND - 1) {
        // Every k-mer of this super-tile has left it (the barrier after the histogram): stage the next one now, between
        // scatter and copy-out (split_stage's comment)
        const uint64_t nst = work ? st + gridDim.x : st;
        tile_encode<FMT, TileSuper>(L.tile, raw, a, a.pos0 + (int64_t)nst * SUPER_SPAN, ctrs, tid, nst < nsuper);
        tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)(nst + gridDim.x) * SUPER_SPAN, tid, next_first, nst + gridDim.x < nsuper);
        next_first = first_of(nst + 2 * (uint64_t)gridDim.x);
        KC_SPLIT_STAMP(5)  // stage the next super-tile
      }
      if (work) {
        split_copy_out<NL>(
            L.sp, sorted, sbucket, total, D, [&](const uint64_t (&r)[NL]) { return cp_b1(r[0], gm); },
            [&](uint32_t, const uint64_t (&r)[NL]) { l1_overflow<NL, CP>(gm, bb, cb, r); },
            [&](size_t i, const uint64_t (&r)[NL]) { store_words<NL>(D.arena, i, r); });
        buf ^= 1;
      }
    }
  }
  if ((uint32_t)tid < P1) bb.cnt1[(size_t)g * P1 + tid] = cst.cur;
  if (tid == 0) {
    bb.used1[g] = min(L.sp.arena_used, gm.A1);
    bb.used1[gm.G + g] = min(L.sp.arena_top, gm.A1);
  }
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (lane_id() == 0 && n_ins) atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
}

// ---- level 1 from reads, sixteen k-mers per thread and round (compact records of a fixed k) ---------------------------
// The rounds of kc_l1_reads_kernel stage 8192 records, eight per destination: runs of 64 bytes that touch 1.44 lines of
// 128 bytes each, and a scan + reserve per eight k-mers of a thread.  Here a thread cuts the SIXTEEN k-mers that start in
// its own sixteen-base group of the staged super-tile (two windows of eight, cp_run_fixed), so a super-tile is ONE round:
// half the barriers, scans and reservations per k-mer, runs of sixteen records.  What makes 16384 staged records fit the
// LDS beside the tile: a staged record is six bytes -- the 32 bits of the mix below the bucket in one array, bucket and
// extension codes (16 bits) in another -- instead of the eight it has in memory; the copy-out puts the eight together
// again.  Every destination's run starts on an EVEN position of the staging (one pad slot per destination at most), so
// that the copy-out can take the records two at a time: a pair never straddles two destinations, one 16-byte store per
// pair and lane (an odd run's last record, and a pair that straddles two chunks of its chain, go out as single records).
constexpr int R16 = 16;
constexpr uint32_t ST16_MAIN = (uint32_t)WGB * R16 + PMAX;  // a round's records + one pad per destination
constexpr uint32_t ST16_SLOTS = ST16_MAIN + 64;             // + one slot per lane for the positions that hold no k-mer
static_assert(SUPER_SPAN % R16 == 0 && SUPER_SPAN / R16 <= WGB, "one run of sixteen positions per thread covers the super-tile");
static_assert(ST16_MAIN < 65536, "staging positions fit 16 bits");
constexpr size_t l1x16_lds_bytes() { return ((sizeof(L1LDS) + 15) & ~size_t(15)) + (size_t)ST16_SLOTS * 6; }

// scan (of the run lengths rounded up to even), reserve, scatter.  br[j] = bucket | extension codes << 10 | rank << 16,
// or ~0.  dst[b] = {x, y as in split_stage, start | run length << 16, records that found room | of which in the old
// last chunk << 16}.  Returns the staged total (pads included).
// mid(): what the kernel has to do between the scatter and the barrier behind it (the next super-tile's staging)
template <class MidFn>
__device__ __forceinline__ uint32_t split_stage_pairs(SplitLDS &L, uint32_t *slo, uint16_t *sbk, int buf, uint32_t P, const uint32_t (&lo)[R16],
                                                      const uint32_t (&br)[R16], const ChainDest &D, ChainState &st, MidFn mid) {
  const int tid = fresh_tid();
  KC_SPLIT_STAMP(1)  // barrier after the histogram
  const uint32_t v = ((uint32_t)tid < P) ? L.hist[buf][tid] : 0u;
  const uint32_t excl = block_excl_scan((v + 1u) & ~1u, L.scan);
  const uint32_t CHm = (1u << D.log2CH) - 1u;
  if ((uint32_t)tid < P) {
    const uint32_t base = st.cur;
    uint32_t fit = v;
    const uint64_t room = ((uint64_t)D.LMAX << D.log2CH) - base;  // the chain holds at most LMAX chunks
    if ((uint64_t)fit > room) fit = (uint32_t)room;
    const uint32_t have = (base + CHm) >> D.log2CH;
    uint32_t k = ((base + fit + CHm) >> D.log2CH) - have, a = 0;
    if (k) {  // (the same reservation as split_stage's)
      const bool own = (uint32_t)tid >= D.own_lo && (uint32_t)tid < D.own_hi;
      if (own) {
        a = atomicAdd(&L.arena_used, k);
        const uint32_t left = D.arena_cap - min(D.arena_cap, L.arena_top);
        if (a + k > left) {
          k = a < left ? left - a : 0;
          const uint64_t cap = ((uint64_t)(have + k) << D.log2CH) - base;
          if ((uint64_t)fit > cap) fit = (uint32_t)cap;
        }
      } else {
        const uint32_t t = atomicAdd(&L.arena_top, k);
        const uint32_t left = D.arena_cap - min(D.arena_cap, L.arena_used);
        if (t + k > left) {
          k = 0;
          const uint64_t cap = ((uint64_t)have << D.log2CH) - base;
          if ((uint64_t)fit > cap) fit = (uint32_t)cap;
        } else {
          a = D.arena_cap - t - k;
        }
      }
      uint32_t *ch = D.chain + (size_t)tid * D.LMAX + have;
      for (uint32_t i = 0; i < k; i++) ch[i] = D.arena_base + a + i;
    }
    uint4 d;
    d.x = ((st.last - D.arena_base - (base >> D.log2CH)) << D.log2CH) + base - excl;
    d.y = ((a - have) << D.log2CH) + base - excl;
    d.z = excl | (v << 16);
    d.w = fit | (((base & CHm) ? min(fit, (CHm + 1u) - (base & CHm)) : 0u) << 16);
    L.dst[tid] = d;
    if (k) st.last = D.arena_base + a + k - 1;
    st.cur = base + fit;
    // the scatter takes a run's start from the histogram's word, which has served its purpose: a dense array of words
    // spreads over all the banks, the .z of 16-byte entries over a quarter of them
    L.hist[buf][tid] = excl;
    L.hist[buf ^ 1][tid] = 0;  // next round's histogram
  }
  lds_barrier();
  KC_SPLIT_STAMP(2)  // scan + reserve
  const uint32_t total = L.scan.total;
#pragma unroll
  for (int j0 = 0; j0 < R16; j0 += 8) {
    uint32_t pos[8];
#pragma unroll
    for (int j = 0; j < 8; j++) pos[j] = L.hist[buf][br[j0 + j] & (PMAX - 1)];
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint32_t bj = br[j0 + j];
      const uint32_t p = bj != ~0u ? pos[j] + (bj >> 16) : ST16_MAIN + lane_id();
      slo[p] = lo[j0 + j];
      sbk[p] = (uint16_t)bj;
    }
  }
  mid();
  lds_barrier();
  KC_SPLIT_STAMP(3)  // scatter to LDS (+ the next super-tile's staging)
  return total;
}

// two staged records (one destination) as one 12-byte store; the address is a multiple of six, i.e. 2-byte aligned
// (three words put together by hand: of a struct with 16-bit members the compiler makes four or five narrow stores)
struct __attribute__((packed, aligned(2))) Rec6Pair {
  uint32_t w0, w1, w2;  // lo0 | bk0, lo1's low half << 16 | lo1's high half, bk1 << 16
};
struct __attribute__((packed, aligned(2))) Rec6 {
  uint32_t lo;
  uint16_t bk;
};
static_assert(sizeof(Rec6Pair) == 12 && sizeof(Rec6) == 6, "six bytes a record");

template <class OvfFn>
__device__ __forceinline__ void split_copy_out_pairs(SplitLDS &L, const uint32_t *slo, const uint16_t *sbk, uint32_t total, const ChainDest &D,
                                                     const Geom &gm, OvfFn overflow) {
  const int tid = fresh_tid();
#ifndef KC_COPY_U
#define KC_COPY_U 2
#endif
  constexpr int U = KC_COPY_U;  // pairs per thread and trip: first all their records, then all their destinations, then the stores
  // (D.arena: the owner's part of the level-1 arena in six-byte records, l1_dest6)
  uint8_t *const arena0 = reinterpret_cast<uint8_t *>(D.arena) + ((size_t)D.arena_base << D.log2CH) * 6;
  const uint32_t shb = 64u - gm.la, shl = 64u - gm.k2;  // (the record as a 64-bit mixed one, for the overflow list only)
  for (uint32_t i0 = 2u * (uint32_t)tid; i0 < total; i0 += 2u * U * WGB) {
    uint64_t lo2[U];
    uint32_t bk2[U];
    uint4 d[U];
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = i0 + 2u * (uint32_t)u * WGB;
      const uint32_t ii = i < total ? i : i0;  // in range: i0 < total
      lo2[u] = *reinterpret_cast<const uint64_t *>(slo + ii);
      bk2[u] = *reinterpret_cast<const uint32_t *>(sbk + ii);
    }
#pragma unroll
    for (int u = 0; u < U; u++) d[u] = L.dst[bk2[u] & (PMAX - 1)];
    bool odd = false;
#pragma unroll
    for (int u = 0; u < U; u++) {
      const uint32_t i = i0 + 2u * (uint32_t)u * WGB;
      const uint32_t j = i - (d[u].z & 0xFFFFu), v = d[u].z >> 16, fit = d[u].w & 0xFFFFu, wold = d[u].w >> 16;
      const bool live = i < total;  // (a pair's first position always holds a record: runs start on even positions)
      const bool f0 = live && j < fit, real1 = live && j + 1u < v, f1 = real1 && j + 1u < fit;
      // x and y are "index minus staging position" modulo 2^32: the sums must wrap in 32 bits before they are widened
      const uint32_t at0 = (j < wold ? d[u].x : d[u].y) + i, at1 = (j + 1u < wold ? d[u].x : d[u].y) + i + 1u;
      const bool pair = f0 && f1 && at1 == at0 + 1u;
      if (pair) {
        const uint32_t l1 = (uint32_t)(lo2[u] >> 32);
        Rec6Pair r;
        r.w0 = (uint32_t)lo2[u];
        r.w1 = (bk2[u] & 0xFFFFu) | (l1 << 16);
        r.w2 = (l1 >> 16) | (bk2[u] & 0xFFFF0000u);
        *reinterpret_cast<Rec6Pair *>(arena0 + (size_t)at0 * 6) = r;
      } else if (f0) {
        Rec6 r;
        r.lo = (uint32_t)lo2[u];
        r.bk = (uint16_t)bk2[u];
        *reinterpret_cast<Rec6 *>(arena0 + (size_t)at0 * 6) = r;
      }
      odd |= (real1 && !pair) || (live && !f0);
    }
    if (__any(odd)) {  // seldom: a pair across two chunks of its chain; a full chain or arena
#pragma unroll
      for (int u = 0; u < U; u++) {
        const uint32_t i = i0 + 2u * (uint32_t)u * WGB;
        if (i >= total) continue;
        const uint32_t j = i - (d[u].z & 0xFFFFu), v = d[u].z >> 16, fit = d[u].w & 0xFFFFu, wold = d[u].w >> 16;
        const uint32_t at0 = (j < wold ? d[u].x : d[u].y) + i, at1 = (j + 1u < wold ? d[u].x : d[u].y) + i + 1u;
        const uint32_t l0 = (uint32_t)lo2[u], l1 = (uint32_t)(lo2[u] >> 32);
        // (where the 32 bits hold some of the bucket's as well -- k < 16 + la / 2 -- the two ORs put the same bits in the same place)
        const uint64_t bh = (uint64_t)(bk2[u] & (PMAX - 1)) << shb;
        const uint64_t ra = bh | ((uint64_t)l0 << shl) | (uint64_t)((bk2[u] >> 10) & 63u);
        const uint64_t rb = bh | ((uint64_t)l1 << shl) | (uint64_t)(bk2[u] >> 26);
        const bool f0 = j < fit, real1 = j + 1u < v, f1 = real1 && j + 1u < fit;
        if (!f0) overflow(ra);
        if (real1 && !(f0 && f1 && at1 == at0 + 1u)) {
          if (f1) {
            Rec6 r;
            r.lo = l1;
            r.bk = (uint16_t)(bk2[u] >> 16);
            *reinterpret_cast<Rec6 *>(arena0 + (size_t)at1 * 6) = r;
          } else {
            overflow(rb);
          }
        }
      }
    }
  }
  KC_SPLIT_STAMP(4)  // copy-out
}

template <int FMT, bool SH, int KK>
__global__ __launch_bounds__(WGB) void kc_l1_reads16_kernel(ExtractArgs a, Geom gm, BucketBufs bb, uint64_t nsuper, uint32_t rot, uint64_t *ctrs,
                                                            uint64_t *cb) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1LDS &L = *reinterpret_cast<L1LDS *>(smem);
  uint32_t *slo = reinterpret_cast<uint32_t *>(smem + ((sizeof(L1LDS) + 15) & ~size_t(15)));
  uint16_t *sbk = reinterpret_cast<uint16_t *>(slo + ST16_SLOTS);
  const int tid = threadIdx.x;
  const uint32_t g = (blockIdx.x + rot) % gm.G, P1 = gm.P1;
  ChainDest D = l1_dest6(gm, bb, g);
#ifdef KC_STAMPS
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
  D.stamps = cb + 8;
  D.tprev = &tprev_;
#endif
  ChainState cst = split_load_state(L.sp, P1, bb.cnt1 + (size_t)g * P1, D.chain, D.LMAX, D.log2CH, bb.used1[g], bb.used1[gm.G + g]);
  __syncthreads();
  uint32_t n_ins = 0;
  int buf = 0;
  TileRaw<TileSuper> raw;  // the next super-tile's bytes, on their way while this one is split (kc_l1_reads_kernel)
  auto first_of = [&](uint64_t st) -> uint64_t { return (FMT != FMT_SEQBLOCK && st < nsuper) ? a.tile_first[st] : 0; };
  tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, tid, first_of(blockIdx.x), blockIdx.x < nsuper);
  uint64_t next_first = first_of((uint64_t)blockIdx.x + gridDim.x);
  constexpr int RUNS = SUPER_SPAN / R16;
  // The first super-tile is staged up front.  Every further one is staged INSIDE its predecessor's round, between barriers
  // the round has anyway: its gap words are cleared after the barrier behind the histogram (every k-mer has left the old
  // tile by then), its codes are written beside the scatter, and the barrier behind the scatter completes it -- the two
  // barriers of a staging by itself, and the skew the waves collect in front of them, are gone.
  {
    const int ft = fresh_tid();
    tile_encode<FMT, TileSuper>(L.tile, raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, ctrs, ft, blockIdx.x < nsuper);
    tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)((uint64_t)blockIdx.x + gridDim.x) * SUPER_SPAN, ft, next_first,
                                  (uint64_t)blockIdx.x + gridDim.x < nsuper);
    next_first = first_of((uint64_t)blockIdx.x + 2 * (uint64_t)gridDim.x);
  }
  for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
    uint32_t lo[R16], br[R16];
    const bool active = tid < RUNS;
    const int lp0 = PRE + (active ? tid : 0) * R16;
    {
      uint32_t l8[8], b8[8];
      cp_run_fixed<KK, 8, SH>(L.tile, lp0, active, gm, a, l8, b8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        lo[j] = l8[j];
        br[j] = b8[j];
      }
      cp_run_fixed<KK, 8, SH>(L.tile, lp0 + 8, active, gm, a, l8, b8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        lo[8 + j] = l8[j];
        br[8 + j] = b8[j];
      }
    }
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const bool valid = br[j] != ~0u;
      const uint32_t rank = hist_rank(L.sp, buf, br[j] & (PMAX - 1), valid);
      br[j] = valid ? (br[j] | (rank << 16)) : ~0u;
      n_ins += valid ? 1u : 0u;
    }
    KC_SPLIT_STAMP(0)  // cut the k-mers out of the super-tile, histogram
    lds_barrier();
    // (the thread id as a fresh value: addresses that depend on it only are otherwise kept in registers across the whole
    // loop, and in this kernel spilled)
    const int ft = fresh_tid();
    const uint64_t nst = st + gridDim.x;
    tile_encode_clear<TileSuper>(L.tile, raw, ft);
    const uint32_t total = split_stage_pairs(L.sp, slo, sbk, buf, P1, lo, br, D, cst, [&]() {
      tile_encode_fill<FMT, TileSuper>(L.tile, raw, a, a.pos0 + (int64_t)nst * SUPER_SPAN, ctrs, ft, nst < nsuper);
      tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)(nst + gridDim.x) * SUPER_SPAN, ft, next_first, nst + gridDim.x < nsuper);
      next_first = first_of(nst + 2 * (uint64_t)gridDim.x);
    });
    split_copy_out_pairs(L.sp, slo, sbk, total, D, gm, [&](uint64_t r) {
      const uint64_t rr[1] = {r};
      l1_overflow<1, true>(gm, bb, cb, rr);
    });
    buf ^= 1;
  }
  if ((uint32_t)tid < P1) bb.cnt1[(size_t)g * P1 + tid] = cst.cur;
  if (tid == 0) {
    bb.used1[g] = min(L.sp.arena_used, gm.A1);
    bb.used1[gm.G + g] = min(L.sp.arena_top, gm.A1);
  }
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (lane_id() == 0 && n_ins) atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
}

// ---- sender side of the shard exchange: bin a block's records by owner shard ------------------------------
// Same tile extraction and ranking as level 1, with the owner shard as the bucket and the caller's per-shard segments
// as destinations: one global cursor bump per shard and round (a handful of atomics per 8 Ki records) instead of one
// per wave and shard, and the records go from the registers straight to their segment.
template <int NL, int R>
__device__ __forceinline__ void split_round_flat(SplitLDS &L, int buf, uint32_t P, const uint64_t (&rec)[R][NL], const uint32_t (&br)[R],
                                                 uint64_t *records, uint64_t seg_cap, uint64_t *cursors, uint64_t *overflow_flag) {
  const int tid = threadIdx.x;
  if ((uint32_t)tid < P) {
    const uint32_t v = L.hist[buf][tid];
    uint64_t base = 0;
    if (v) base = atomicAdd((unsigned long long *)&cursors[tid], (unsigned long long)v);
    const uint64_t room = base < seg_cap ? seg_cap - base : 0;
    uint4 d;
    d.x = (uint32_t)base;
    d.y = (uint32_t)(base >> 32);
    d.z = (uint64_t)v <= room ? v : (uint32_t)room;
    d.w = 0;
    L.dst[tid] = d;
    L.hist[buf ^ 1][tid] = 0;
  }
  lds_barrier();
#pragma unroll
  for (int j = 0; j < R; j++) {
    if (br[j] != ~0u) {
      const uint32_t b = br[j] & (PMAX - 1), rank = br[j] >> 10;
      const uint4 d = L.dst[b];
      if (rank < d.z) {
        uint64_t *o = records + ((uint64_t)b * seg_cap + (((uint64_t)d.y << 32) | d.x) + rank) * NL;
#pragma unroll
        for (int w = 0; w < NL; w++) o[w] = rec[j][w];
      } else {
        *overflow_flag = 1;
      }
    }
  }
}

template <int NL, int FMT>
__global__ __launch_bounds__(WGB) void kc_bin_reads_kernel(ExtractArgs a, uint64_t nsuper, uint64_t *ctrs) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1LDS &L = *reinterpret_cast<L1LDS *>(smem);
  constexpr int RPOS = Rnd<NL>::RPOS_READS;
  const int tid = threadIdx.x;
  const uint32_t P = a.rank_n;
  if ((uint32_t)tid < PMAX) {
    L.sp.hist[0][tid] = 0;
    L.sp.hist[1][tid] = 0;
  }
  __syncthreads();
  int buf = 0;
  TileRaw<TileSuper> raw;  // see kc_l1_reads_kernel
  auto first_of = [&](uint64_t st) -> uint64_t { return (FMT != FMT_SEQBLOCK && st < nsuper) ? a.tile_first[st] : 0; };
  tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, tid, first_of(blockIdx.x), blockIdx.x < nsuper);
  uint64_t next_first = first_of((uint64_t)blockIdx.x + gridDim.x);
  constexpr int RUNS = SUPER_SPAN / RPOS;
  static_assert(SUPER_SPAN % RPOS == 0, "runs must tile the span");
  for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
    const int64_t T0 = a.pos0 + (int64_t)st * SUPER_SPAN;
    tile_encode<FMT, TileSuper>(L.tile, raw, a, T0, ctrs, tid, true);
    {
      const uint64_t nst = st + gridDim.x;
      tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)nst * SUPER_SPAN, tid, next_first, nst < nsuper);
      next_first = first_of(nst + gridDim.x);
    }
#pragma unroll 1
    for (int round = 0; round < (RUNS + WGB - 1) / WGB; round++) {
      uint64_t rec[RPOS][NL];
      uint32_t br[RPOS];
      const int run_id = round * WGB + tid;
      const bool active = run_id < RUNS;
      const int lp0 = PRE + (active ? run_id : 0) * RPOS;
      // k = 21 (MHM2's one-word k) with the library's own partition: the run cut out of one window by constant shifts
      bool fixed21 = false;
      uint32_t okm = 0;
      uint64_t hh[RPOS];
      if constexpr (NL == 1 && RPOS == 8) {
        fixed21 = a.k == 21 && !a.reference_owner;  // the same for every thread
        if (fixed21) okm = cut_run_fixed<21, 8>(L.tile, lp0, active, reinterpret_cast<uint64_t (&)[RPOS]>(rec), hh);
      }
      KmerRun<NL> run;
      if (!fixed21) run_begin<NL>(run, L.tile, lp0, a.k);
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        uint64_t h = 0;
        uint32_t owner = 0;
        bool valid;
        if (fixed21) {
          valid = (okm >> j) & 1u;
          owner = kc_owner_of_hash(hh[j], P);
        } else {
          valid = run_kmer<NL>(run, j, lp0, a.k, rec[j], h, P, a.reference_owner, &owner) && active;
          if (j + 1 < RPOS) run_advance<NL>(run, j, lp0, a.k);
        }
        br[j] = ~0u;
        const uint32_t b = P > 1 ? owner : 0u;
        if (P <= 16) {
          // few shards: the lanes of a wave that go to the same one take their ranks from one LDS add (64 separate
          // adds to a handful of words would be serialised)
          for (uint32_t d = 0; d < P; d++) {
            const bool mine = valid && b == d;
            const uint64_t m = __ballot(mine);
            if (m) {  // wave-uniform
              const int leader = __ffsll((long long)m) - 1;
              uint32_t base = 0;
              if ((int)lane_id() == leader) base = atomicAdd(&L.sp.hist[buf][d], (uint32_t)__popcll(m));
              base = __shfl(base, leader);
              if (mine) br[j] = d | ((base + (uint32_t)__popcll(m & ((1ULL << lane_id()) - 1ULL))) << 10);
            }
          }
        } else if (valid) {
          const uint32_t rank = atomicAdd(&L.sp.hist[buf][b], 1u);
          br[j] = b | (rank << 10);
        }
      }
      lds_barrier();
      split_round_flat<NL, RPOS>(L.sp, buf, P, rec, br, a.records, a.seg_capacity, ctrs + CTR_BIN0, ctrs + CTR_OVERFLOW);
      buf ^= 1;
    }
  }
}

// ---- level 1 from records (receiver side of the shard exchange) -------------------------------------
struct L1RLDS {
  SplitLDS sp;
};

template <int NL, bool CP>
__global__ __launch_bounds__(WGB) void kc_l1_records_kernel(const uint64_t *recs, uint64_t n, Geom gm, BucketBufs bb, uint32_t rot,
                                                            uint64_t *ctrs, uint64_t *cb) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1RLDS &L = *reinterpret_cast<L1RLDS *>(smem);
  uint64_t *sorted = reinterpret_cast<uint64_t *>(smem + ((sizeof(L1RLDS) + 15) & ~size_t(15)));
  constexpr int RPOS = Rnd<NL>::RPOS;
  const int tid = threadIdx.x;
  const uint32_t g = (blockIdx.x + rot) % gm.G, P1 = gm.P1;
  const ChainDest D = l1_dest<NL>(gm, bb, g);
  ChainState cst = split_load_state(L.sp, P1, bb.cnt1 + (size_t)g * P1, D.chain, D.LMAX, D.log2CH, bb.used1[g], bb.used1[gm.G + g]);
  __syncthreads();
  const uint64_t per_round = (uint64_t)WGB * RPOS;
  const uint64_t nrounds = (n + per_round - 1) / per_round;
  uint32_t n_ins = 0;
  int buf = 0;
  // the records of this workgroup's next round are already on their way while this round is split
  uint64_t nxt[RPOS][NL];
  auto load_round = [&](uint64_t rd) {  // n > 0; no branches around the loads (a lane past the end re-reads the last record)
#pragma unroll
    for (int j = 0; j < RPOS; j++) {
      uint64_t i = rd * per_round + (uint64_t)j * WGB + tid;
      i = i < n ? i : n - 1;
#pragma unroll
      for (int w = 0; w < NL; w++) nxt[j][w] = recs[i * NL + w];
    }
  };
  // rec: this round's records; nxt: the next round's, taken over in the middle of the round (split_stage's comment)
  uint64_t rec[RPOS][NL];
  if (n) load_round(blockIdx.x);
#pragma unroll
  for (int j = 0; j < RPOS; j++)
#pragma unroll
    for (int w = 0; w < NL; w++) rec[j][w] = nxt[j][w];
  if (n) load_round((uint64_t)blockIdx.x + gridDim.x);
  for (uint64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
    uint32_t br[RPOS];
    // all the ranks of a thread in flight together (no branch around the LDS adds: a lane past the end, which holds a
    // copy of the last record, bumps a word of its own lane's instead)
#pragma unroll
    for (int j = 0; j < RPOS; j++) {
      const bool valid = rd * per_round + (uint64_t)j * WGB + tid < n;
      if (CP) rec[j][0] = cp_mix_rec(rec[j][0], gm);
      const uint32_t b = CP ? cp_b1(rec[j][0], gm) : hash_b1(rec_hash<NL>(rec[j]), gm);
      const uint32_t rank = hist_rank(L.sp, buf, b, valid);
      br[j] = valid ? (b | (rank << 10)) : ~0u;
      n_ins += valid ? 1u : 0u;
    }
    lds_barrier();
    const uint32_t total = split_stage<NL, RPOS>(
        L.sp, sorted, nullptr, buf, P1,
        [&](int j, uint64_t (&o)[NL]) {
#pragma unroll
          for (int w = 0; w < NL; w++) o[w] = rec[j][w];
        },
        br, D, cst);
#pragma unroll
    for (int j = 0; j < RPOS; j++)
#pragma unroll
      for (int w = 0; w < NL; w++) rec[j][w] = nxt[j][w];
    load_round(rd + 2 * (uint64_t)gridDim.x);
    split_copy_out<NL>(
        L.sp, sorted, nullptr, total, D, [&](const uint64_t (&r)[NL]) { return CP ? cp_b1(r[0], gm) : hash_b1(rec_hash<NL>(r), gm); },
        [&](uint32_t, const uint64_t (&r)[NL]) { l1_overflow<NL, CP>(gm, bb, cb, r); },
        [&](size_t i, const uint64_t (&r)[NL]) { store_words<NL>(D.arena, i, r); });
    buf ^= 1;
  }
  if ((uint32_t)tid < P1) bb.cnt1[(size_t)g * P1 + tid] = cst.cur;
  if (tid == 0) {
    bb.used1[g] = min(L.sp.arena_used, gm.A1);
    bb.used1[gm.G + g] = min(L.sp.arena_top, gm.A1);
  }
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (lane_id() == 0 && n_ins) {
    atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
    atomicAdd((unsigned long long *)&ctrs[CTR_EXPECT], (unsigned long long)n_ins);
  }
}

// The same for compact records in the short form (Geom::cp with 2k - la <= 32): the round's records are held as the 32 bits
// of the mix below the bucket + bucket and extension codes, staged in six bytes and copied out in pairs, like
// kc_l1_reads16_kernel's (split_stage_pairs / split_copy_out_pairs).
constexpr size_t l1r16_lds_bytes() { return ((sizeof(L1RLDS) + 15) & ~size_t(15)) + (size_t)ST16_SLOTS * 6; }
__global__ __launch_bounds__(WGB) void kc_l1_records16_kernel(const uint64_t *recs, uint64_t n, Geom gm, BucketBufs bb, uint32_t rot, uint64_t *ctrs,
                                                              uint64_t *cb) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1RLDS &L = *reinterpret_cast<L1RLDS *>(smem);
  uint32_t *slo = reinterpret_cast<uint32_t *>(smem + ((sizeof(L1RLDS) + 15) & ~size_t(15)));
  uint16_t *sbk = reinterpret_cast<uint16_t *>(slo + ST16_SLOTS);
  const int tid = threadIdx.x;
  const uint32_t g = (blockIdx.x + rot) % gm.G, P1 = gm.P1;
  ChainDest D = l1_dest6(gm, bb, g);
#ifdef KC_STAMPS
  unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
  D.stamps = nullptr;
  D.tprev = &tprev_;
#endif
  ChainState cst = split_load_state(L.sp, P1, bb.cnt1 + (size_t)g * P1, D.chain, D.LMAX, D.log2CH, bb.used1[g], bb.used1[gm.G + g]);
  __syncthreads();
  const uint64_t per_round = (uint64_t)WGB * R16;
  const uint64_t nrounds = (n + per_round - 1) / per_round;
  uint32_t n_ins = 0;
  int buf = 0;
  uint64_t nxt[R16];  // the next round's records, on their way while this round is split
  auto load_round = [&](uint64_t rd) {  // n > 0; no branches around the loads (a lane past the end re-reads the last record)
#pragma unroll
    for (int j = 0; j < R16; j++) {
      uint64_t i = rd * per_round + (uint64_t)j * WGB + tid;
      i = i < n ? i : n - 1;
      nxt[j] = recs[i];
    }
  };
  uint32_t lo[R16], br[R16];
  auto take_over = [&]() {  // k-mer record -> mixed record -> short form
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const uint64_t m = cp_mix_rec(nxt[j], gm);
      lo[j] = (uint32_t)(m >> (64u - gm.k2));
      br[j] = cp_b1(m, gm) | (((uint32_t)m & 63u) << 10);
    }
  };
  if (n) load_round(blockIdx.x);
  take_over();
  if (n) load_round((uint64_t)blockIdx.x + gridDim.x);
  for (uint64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const bool valid = rd * per_round + (uint64_t)j * WGB + tid < n;
      const uint32_t rank = hist_rank(L.sp, buf, br[j] & (PMAX - 1), valid);
      br[j] = valid ? (br[j] | (rank << 16)) : ~0u;
      n_ins += valid ? 1u : 0u;
    }
    lds_barrier();
    const uint32_t total = split_stage_pairs(L.sp, slo, sbk, buf, P1, lo, br, D, cst, [&]() {});
    take_over();
    load_round(rd + 2 * (uint64_t)gridDim.x);
    split_copy_out_pairs(L.sp, slo, sbk, total, D, gm, [&](uint64_t r) {
      const uint64_t rr[1] = {r};
      l1_overflow<1, true>(gm, bb, cb, rr);
    });
    buf ^= 1;
  }
  if ((uint32_t)tid < P1) bb.cnt1[(size_t)g * P1 + tid] = cst.cur;
  if (tid == 0) {
    bb.used1[g] = min(L.sp.arena_used, gm.A1);
    bb.used1[gm.G + g] = min(L.sp.arena_top, gm.A1);
  }
  for (int o = 32; o > 0; o >>= 1) n_ins += __shfl_down(n_ins, o);
  if (lane_id() == 0 && n_ins) {
    atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n_ins);
    atomicAdd((unsigned long long *)&ctrs[CTR_EXPECT], (unsigned long long)n_ins);
  }
}

// record i of the (writer g, bucket b) chain
template <int NL>
__device__ __forceinline__ const uint64_t *l1_record(const Geom &gm, const BucketBufs &bb, uint32_t g, uint32_t b, uint32_t i) {
  const uint32_t chunk = bb.chain1[((size_t)g * gm.P1 + b) * gm.L1MAX + (i >> gm.log2CH1)];
  return bb.rec1 + (((((size_t)g * gm.A1) + chunk) << gm.log2CH1) + (i & ((1u << gm.log2CH1) - 1u))) * NL;
}

// the same in a geometry of six-byte records, as the 64-bit mixed record it stands for
__device__ __forceinline__ uint64_t l1_record6(const Geom &gm, const BucketBufs &bb, uint32_t g, uint32_t b, uint32_t i) {
  const uint32_t chunk = bb.chain1[((size_t)g * gm.P1 + b) * gm.L1MAX + (i >> gm.log2CH1)];
  const uint8_t *p = reinterpret_cast<const uint8_t *>(bb.rec1) + (((((size_t)g * gm.A1) + chunk) << gm.log2CH1) + (i & ((1u << gm.log2CH1) - 1u))) * 6;
  const Rec6 r = *reinterpret_cast<const Rec6 *>(p);
  return ((uint64_t)b << (64u - gm.la)) | ((uint64_t)r.lo << (64u - gm.k2)) | (uint64_t)((r.bk >> 10) & 63u);
}

// record i of region r's chain
template <int NL>
__device__ __forceinline__ const uint64_t *l2_record(const Geom &gm, const BucketBufs &bb, size_t r, uint32_t i) {
  const uint32_t chunk = bb.chain2[r * gm.L2MAX + (i >> gm.log2CH2)];
  return bb.rec2 + (((size_t)chunk << gm.log2CH2) + (i & ((1u << gm.log2CH2) - 1u))) * NL;
}

// compact records: 32 bits each
__device__ __forceinline__ const uint32_t *l2_record32(const Geom &gm, const BucketBufs &bb, size_t r, uint32_t i) {
  const uint32_t chunk = bb.chain2[r * gm.L2MAX + (i >> gm.log2CH2)];
  return reinterpret_cast<const uint32_t *>(bb.rec2) + (((size_t)chunk << gm.log2CH2) + (i & ((1u << gm.log2CH2) - 1u)));
}

// ---- between the levels: every bucket gets a private, exactly sized part of the level-2 arena ---------
// chunks(b) = ceil(records(b) / CH2) + P2: each of its P2 regions wastes less than one chunk
// per_bucket != 0 (level 2 in instalments: the parts must be fixed before the records are all there): every bucket gets
// room for that many records -- the buffer's capacity over the fan-out; what a bucket has beyond it overflows (exact)
__global__ __launch_bounds__(WGB) void kc_bucket_prefix_kernel(Geom gm, BucketBufs bb, FlatSrc fs, uint64_t *cb, uint64_t per_bucket) {
  __shared__ ScanLDS S;
  const uint32_t b = threadIdx.x;
  uint32_t chunks = 0;
  if (b < gm.P1) {
    uint64_t n = per_bucket;
    if (!per_bucket) {
      for (uint32_t g = 0; g < gm.G; g++) n += bb.cnt1[(size_t)g * gm.P1 + b];
      if (b >= fs.b_lo && b < fs.b_hi)
        for (uint32_t f = 0; f < fs.F; f++) n += fs.cnt[(size_t)f * fs.nbo + (b - fs.b_lo)];
    }
    chunks = (uint32_t)((n + (1u << gm.log2CH2) - 1) >> gm.log2CH2) + gm.P2;
  }
  const uint32_t e = block_excl_scan(chunks, S);
  if (b < gm.P1) bb.base2[b] = e;
  if (b == 0) {
    bb.base2[gm.P1] = S.total;
    if (S.total > gm.A2 && !per_bucket) atomicOr((unsigned long long *)&cb[CB_FATAL], (unsigned long long)FATAL_ARENA2);  // the host keeps the buffered records within its capacity
  }
}

// ---- the level-2 arena moves into a larger one (compact records; bk_l2_reserve) ---------------------------------------------
// Every bucket's part of the arena grows: its used chunks are copied to where the part now starts, and the chunk ids of
// its regions' chains move by the same amount.  One workgroup per bucket.
__global__ __launch_bounds__(WGB) void kc_l2_grow_kernel(Geom gm, const uint32_t *old_rec, uint32_t *new_rec, const uint32_t *old_base,
                                                        const uint32_t *new_base, const uint32_t *used2, uint32_t *chain2, const uint32_t *cnt2) {
  const uint32_t b = blockIdx.x, tid = threadIdx.x;
  const size_t words = (size_t)used2[b] << gm.log2CH2;  // 32-bit records
  const uint32_t *src = old_rec + ((size_t)old_base[b] << gm.log2CH2);
  uint32_t *dst = new_rec + ((size_t)new_base[b] << gm.log2CH2);
  for (size_t i = tid; i < words; i += WGB) dst[i] = src[i];
  const uint32_t delta = new_base[b] - old_base[b], CHm = (1u << gm.log2CH2) - 1u;
  for (uint32_t r = tid; r < gm.P2; r += WGB) {
    const size_t reg = (size_t)b * gm.P2 + r;
    const uint32_t nch = min((cnt2[reg] + CHm) >> gm.log2CH2, gm.L2MAX);
    for (uint32_t i = 0; i < nch; i++) chain2[reg * gm.L2MAX + i] += delta;
  }
}

// ---- level 2 ------------------------------------------------------------------------------------------
typedef uint64_t __attribute__((aligned(1))) U64Unaligned;  // an 8-byte load from any byte address (one instruction on gfx950)
struct L2LDS {
  SplitLDS sp;
  uint32_t pre[GMAX + 1];  // prefix of the bucket's G segment lengths (+ those of its flat sources)
  union {                  // (never both: instalments are not for the shard flow; the four-word kernel has no LDS to spare)
    uint64_t flo[FLAT_MAX];  // shard flow: where the bucket's run starts in each flat source
    uint32_t skip[GMAX];     // INC: records at the head of each chain that an earlier instalment has taken
  };
  static_assert(sizeof(uint64_t) * FLAT_MAX >= sizeof(uint32_t) * GMAX, "skip fits in flo's place");
};

// CP: the records are mixed ones; what leaves for the regions is their 32-bit remainder (cp_pack32)
// CR: compact records whose mix fits 32 bits below the bucket (2k - la <= 32) are kept in the short register form of
// split_stage between the rounds (half the registers of the full records)
// FL: the shard flow -- only the buckets [fs.b_lo, fs.b_hi) are this shard's, and a bucket's records are its G chains
// followed by fs.F flat sources (G + fs.F <= GMAX)
// INC: level 2 in instalments (the host pipe: the records that have arrived are split while the rest of the input is
// still on its way over PCIe, so that only the last block's records and the count kernel are left when the last byte is
// in).  An instalment takes, of every chain, what came after bb.done1 and goes on where the last one stopped: the
// regions' chains (cnt2, chain2) and the bucket's share of the arena (used2) persist; the buckets' parts of the arena
// were fixed up front (kc_bucket_prefix_kernel with per_bucket).
template <int NL, bool CP, bool CR, bool FL, bool INC = false>
__global__ __launch_bounds__(WGB) void kc_l2_split_kernel(Geom gm, BucketBufs bb, FlatSrc fs, uint64_t *cb) {
  static_assert(!CR || (CP && NL == 1), "the short form is one of compact records");
  static_assert(!(INC && FL), "instalments are for a context that is not in the shard flow");
  extern __shared__ __align__(16) uint8_t smem[];
  L2LDS &L = *reinterpret_cast<L2LDS *>(smem);
  uint64_t *sorted = reinterpret_cast<uint64_t *>(smem + ((sizeof(L2LDS) + 15) & ~size_t(15)));
  constexpr int RPOS = Rnd<NL>::RPOS;
  const int tid = threadIdx.x;
  const uint32_t P1 = gm.P1, P2 = gm.P2, G = gm.G;
  const uint32_t GT = FL ? G + fs.F : G;  // segments of a bucket
  for (uint32_t b1 = (FL ? fs.b_lo : 0u) + blockIdx.x; b1 < (FL ? fs.b_hi : P1); b1 += gridDim.x) {
    // prefix over the segments of this bucket
    {
      const int tid = fresh_tid();  // (hoisted out of the bucket loop, this address and the one of cnt2 below were spilled)
      uint32_t v = ((uint32_t)tid < G) ? bb.cnt1[(size_t)tid * P1 + b1] : 0u;
      if constexpr (INC) {
        const uint32_t d = ((uint32_t)tid < G) ? bb.done1[(size_t)tid * P1 + b1] : 0u;
        if ((uint32_t)tid < G) L.skip[tid] = d;
        v -= d;
      }
      if constexpr (FL) {
        if ((uint32_t)tid >= G && (uint32_t)tid < GT) {
          v = fs.cnt[(size_t)(tid - G) * fs.nbo + (b1 - fs.b_lo)];
          L.flo[tid - G] = fs.at[(size_t)(tid - G) * fs.nbo + (b1 - fs.b_lo)];
        }
      }
      const uint32_t e = block_excl_scan(v, L.sp.scan);
      if ((uint32_t)tid < GT) L.pre[tid] = e;
      if (tid == 0) L.pre[GT] = L.sp.scan.total;
    }
    ChainDest D;
    D.arena = bb.rec2;
    D.chain = bb.chain2 + (size_t)b1 * P2 * gm.L2MAX;
    D.log2CH = gm.log2CH2;
    D.LMAX = gm.L2MAX;
    D.arena_base = bb.base2[b1];
    D.arena_cap = bb.base2[b1 + 1] - bb.base2[b1];
    D.own_lo = 0;
    D.own_hi = PMAX;
#ifdef KC_ABLATE
    D.abl = gm.abl;
    D.abl_a = CP ? 16 : 8 / NL;
#endif
#ifdef KC_STAMPS
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    D.stamps = cb + 8;
    D.tprev = &tprev_;
#endif
    ChainState cst = split_load_state(L.sp, P2, INC ? bb.cnt2 + (size_t)b1 * P2 : nullptr, D.chain, D.LMAX, D.log2CH, INC ? bb.used2[b1] : 0u);
    __syncthreads();
    const uint32_t n = L.pre[GT];
    const uint32_t per_round = WGB * RPOS;
    // The records of the next round travel while this round is split.  A record's address needs its chunk id, which
    // is itself in memory (chain1): all ids of a round are requested first and all records after them, two memory
    // round trips per round (fetched pairwise, every record load would wait for its own id load: sixteen in a row).
    uint32_t p_ids = 0, p_rec = 0;  // segment cursors of this thread, one per pass (their indices only grow)
    uint64_t nxt[RPOS][NL];
    uint32_t nxt_flat = 0;  // FL && CR: bit j = nxt[j] came from a flat source (wire form)
    const uint32_t CH1m = (1u << gm.log2CH1) - 1u;
    // no branches around the loads (a lane past the end re-reads the bucket's last record): a load inside a
    // conditional block is waited for at the end of that block
    auto load_round = [&](uint32_t v0) {  // n > 0
      uint32_t ids[RPOS];
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        const uint32_t e = min(v0 + (uint32_t)j * WGB + tid, n - 1u);
        while (e >= L.pre[p_ids + 1]) p_ids++;
        // (a record of a flat source needs no chunk id: it re-reads the table's first word, no branch)
        const size_t ci = ((size_t)p_ids * gm.P1 + b1) * gm.L1MAX + ((e - L.pre[p_ids] + (INC ? L.skip[p_ids] : 0u)) >> gm.log2CH1);
        ids[j] = bb.chain1[FL && p_ids >= G ? 0 : ci];
      }
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        const uint32_t e = min(v0 + (uint32_t)j * WGB + tid, n - 1u);
        while (e >= L.pre[p_rec + 1]) p_rec++;
        const uint64_t *src = bb.rec1 + (((((size_t)p_rec * gm.A1) + ids[j]) << gm.log2CH1) + ((e - L.pre[p_rec] + (INC ? L.skip[p_rec] : 0u)) & CH1m)) * NL;
        if constexpr (FL) {
          const uint64_t f0 = L.flo[p_rec >= G ? p_rec - G : 0u];
          if constexpr (CR) {
            // a flat source of short-form records is in the five-byte wire form (kc_shard.hpp): the record's bytes and
            // up to three behind it in ONE unaligned 8-byte load (a bucket's block ends with that much to spare); which
            // of its records a thread took from flat sources it remembers in a bit each
            if (p_rec >= G) src = reinterpret_cast<const uint64_t *>((uintptr_t)f0 + 5u * (size_t)(e - L.pre[p_rec]));
            nxt_flat = p_rec >= G ? (nxt_flat | (1u << j)) : (nxt_flat & ~(1u << j));
            nxt[j][0] = *reinterpret_cast<const U64Unaligned *>(src);
            continue;
          }
          if (p_rec >= G) src = reinterpret_cast<const uint64_t *>((uintptr_t)f0) + (size_t)(e - L.pre[p_rec]) * NL;
        }
#pragma unroll
        for (int w = 0; w < NL; w++) nxt[j][w] = src[w];
      }
    };
    // rec (or lo + br in the short form): this round's records; nxt: the next round's, taken over in the middle of the
    // round (split_stage's comment)
    uint64_t rec[CR ? 1 : RPOS][NL];
    uint32_t lo[CR ? RPOS : 1];
    uint32_t br[RPOS];
    const uint32_t sh_b2 = gm.k2 - gm.la - gm.lb;  // CR: where the region's index starts in the mix
    auto take_over = [&]() {
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        if constexpr (CR) {
          uint32_t ext;
          if (FL && ((nxt_flat >> j) & 1u)) {  // wire form: the 32 bits below the bucket, then the extension bits in a byte
            lo[j] = (uint32_t)nxt[j][0];
            ext = (uint32_t)(nxt[j][0] >> 32) & 63u;
          } else {
            lo[j] = (uint32_t)(nxt[j][0] >> (64u - gm.k2));
            ext = (uint32_t)nxt[j][0] & 63u;
          }
          br[j] = ((lo[j] >> sh_b2) & (P2 - 1u)) | (ext << 10);
        } else {
#pragma unroll
          for (int w = 0; w < NL; w++) rec[j][w] = nxt[j][w];
        }
      }
    };
    if (n) load_round(0);
    take_over();
    if (per_round < n) load_round(per_round);
    int buf = 0;
    for (uint32_t v0 = 0; v0 < n; v0 += per_round) {
      // all the ranks of a thread in flight together (no branch around the LDS adds: a lane past the end, which holds
      // a copy of the bucket's last record, bumps a word of its own lane's instead)
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        const bool valid = v0 + (uint32_t)j * WGB + tid < n;
        uint32_t b;
        if constexpr (CR) b = br[j] & (PMAX - 1);
        else b = CP ? cp_b2(rec[j][0], gm) : hash_b2(rec_hash<NL>(rec[j]), gm);
        const uint32_t rank = hist_rank(L.sp, buf, b, valid);
        if constexpr (CR) br[j] = valid ? (br[j] | (rank << 16)) : ~0u;
        else br[j] = valid ? (b | (rank << 10)) : ~0u;
      }
      KC_SPLIT_STAMP(0)  // histogram
      lds_barrier();
      const uint32_t total = split_stage<NL, RPOS, CR ? 16 : 10>(
          L.sp, sorted, nullptr, buf, P2,
          [&](int j, uint64_t (&o)[NL]) {
            if constexpr (CR) {
              o[0] = ((uint64_t)b1 << (64u - gm.la)) | ((uint64_t)lo[j] << (64u - gm.k2)) | (uint64_t)((br[j] >> 10) & 63u);
            } else {
#pragma unroll
              for (int w = 0; w < NL; w++) o[w] = rec[j][w];
            }
          },
          br, D, cst);
      take_over();
      if (v0 + 2 * (uint64_t)per_round < n) load_round(v0 + 2 * per_round);
      KC_SPLIT_STAMP(5)  // take over the next round's records, request the one after
      split_copy_out<NL>(
          L.sp, sorted, nullptr, total, D, [&](const uint64_t (&r)[NL]) { return CP ? cp_b2(r[0], gm) : hash_b2(rec_hash<NL>(r), gm); },
          [&](uint32_t b, const uint64_t (&r)[NL]) {
            bb.flag[(size_t)b1 * P2 + b] = 1;
            const uint64_t o = atomicAdd((unsigned long long *)&cb[CB_OVF2], 1ULL);
            if (o < bb.ovf2_cap) {
#pragma unroll
              for (int w = 0; w < NL; w++) bb.ovf2[o * NL + w] = (CP && w == 0) ? cp_unmix_rec(r[w], gm) : r[w];
            } else {
              atomicOr((unsigned long long *)&cb[CB_FATAL], (unsigned long long)FATAL_OVF2);
            }
          },
          [&](size_t i, const uint64_t (&r)[NL]) {
            if (CP) reinterpret_cast<uint32_t *>(D.arena)[i] = cp_pack32(r[0], gm);
            else store_words<NL>(D.arena, i, r);
          });
      buf ^= 1;
    }
    {
      const int tid = fresh_tid();
      if ((uint32_t)tid < P2) bb.cnt2[(size_t)b1 * P2 + tid] = cst.cur;
      if constexpr (INC) {
        if ((uint32_t)tid < G) bb.done1[(size_t)tid * P1 + b1] = L.skip[tid] + (L.pre[tid + 1] - L.pre[tid]);
        if (tid == 0) bb.used2[b1] = min(L.sp.arena_used, D.arena_cap);
      }
    }
    __syncthreads();
  }
}

// ---- level 2 over six-byte level-1 records (Geom::rec6) -----------------------------------------------------------------
// The walk, the rounds, the staging and the copy-out of kc_l2_split_kernel<1, true, true, ...>; what differs is how a round's
// records arrive.  A thread takes them two at a time: a bucket's segments (its G chains, then its flat sources) are laid
// end to end with every segment starting at an EVEN in-chain index and padded to an even length, so that a pair of slots
// never straddles two segments or two chunks and a pair of a chain is ONE aligned 12-byte load (a third more records per
// second than 8-byte records one per lane, scripts/ubench_rec6.hip); a slot that holds no record of this pass -- the pad
// behind an odd segment; with instalments the record in front of an odd start, which an earlier instalment took -- is a
// bit in the thread's mask.  A pair of a flat source is the 10 bytes of two wire records (kc_shard.hpp) in one unaligned
// 12-byte load.
struct L2R6LDS {
  SplitLDS sp;
  uint32_t pre[GMAX + 1];  // prefix of the segments' padded lengths, in slots
  uint32_t end[GMAX];      // a segment's true end: records in the chain (in the flat source)
  union {
    uint64_t flo[FLAT_MAX];  // shard flow: where the bucket's run starts in each flat source
    uint32_t skip[GMAX];     // INC: records at the head of each chain that an earlier instalment has taken
  };
};
constexpr size_t l2r6_lds_bytes() { return ((sizeof(L2R6LDS) + 15) & ~size_t(15)) + Rnd<1>::STAGE; }
static_assert(l2r6_lds_bytes() <= 160 * 1024, "level 2's working set fits the LDS");
struct __attribute__((packed, aligned(2))) Load12 {
  uint32_t a, b, c;
};

template <bool FL, bool INC>
__global__ __launch_bounds__(WGB) void kc_l2_rec6_kernel(Geom gm, BucketBufs bb, FlatSrc fs, uint64_t *cb) {
  static_assert(!(INC && FL), "instalments are for a context that is not in the shard flow");
  extern __shared__ __align__(16) uint8_t smem[];
  L2R6LDS &L = *reinterpret_cast<L2R6LDS *>(smem);
  uint64_t *sorted = reinterpret_cast<uint64_t *>(smem + ((sizeof(L2R6LDS) + 15) & ~size_t(15)));
  constexpr int RPOS = 16, NPAIR = 8;
  const int tid = threadIdx.x;
  const uint32_t P1 = gm.P1, P2 = gm.P2, G = gm.G;
  const uint32_t GT = FL ? G + fs.F : G;  // segments of a bucket
  const uint32_t CH1m = (1u << gm.log2CH1) - 1u;
  const uint8_t *const rec1 = reinterpret_cast<const uint8_t *>(bb.rec1);
  for (uint32_t b1 = (FL ? fs.b_lo : 0u) + blockIdx.x; b1 < (FL ? fs.b_hi : P1); b1 += gridDim.x) {
    {
      const int tid = fresh_tid();
      uint32_t v = 0;
      if ((uint32_t)tid < G) {
        const uint32_t cnt = bb.cnt1[(size_t)tid * P1 + b1];
        uint32_t d = 0;
        if constexpr (INC) {
          d = bb.done1[(size_t)tid * P1 + b1];
          L.skip[tid] = d;
        }
        L.end[tid] = cnt;
        v = cnt > d ? ((cnt - (d & ~1u) + 1u) & ~1u) : 0u;  // slots: from the even index at or below the first record to an even end
      }
      if constexpr (FL) {
        if ((uint32_t)tid >= G && (uint32_t)tid < GT) {
          const uint32_t c = fs.cnt[(size_t)(tid - G) * fs.nbo + (b1 - fs.b_lo)];
          L.flo[tid - G] = fs.at[(size_t)(tid - G) * fs.nbo + (b1 - fs.b_lo)];
          L.end[tid] = c;
          v = (c + 1u) & ~1u;
        }
      }
      const uint32_t e = block_excl_scan(v, L.sp.scan);
      if ((uint32_t)tid < GT) L.pre[tid] = e;
      if (tid == 0) L.pre[GT] = L.sp.scan.total;
    }
    ChainDest D;
    D.arena = bb.rec2;
    D.chain = bb.chain2 + (size_t)b1 * P2 * gm.L2MAX;
    D.log2CH = gm.log2CH2;
    D.LMAX = gm.L2MAX;
    D.arena_base = bb.base2[b1];
    D.arena_cap = bb.base2[b1 + 1] - bb.base2[b1];
    D.own_lo = 0;
    D.own_hi = PMAX;
#ifdef KC_ABLATE
    D.abl = gm.abl;
    D.abl_a = 16;
#endif
#ifdef KC_STAMPS
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    D.stamps = cb + 8;
    D.tprev = &tprev_;
#endif
    ChainState cst = split_load_state(L.sp, P2, INC ? bb.cnt2 + (size_t)b1 * P2 : nullptr, D.chain, D.LMAX, D.log2CH, INC ? bb.used2[b1] : 0u);
    __syncthreads();
    const uint32_t n = L.pre[GT];  // slots of the bucket (even)
    const uint32_t per_round = WGB * RPOS;
    uint32_t p_ids = 0, p_rec = 0;  // segment cursors of this thread, one per pass (their indices only grow)
    Load12 nxt[NPAIR];
    uint32_t nxt_valid = 0, nxt_flat = 0;  // bit s: slot s of the thread holds a record; bit j: pair j came from a flat source
    // no branches around the loads (a lane past the end re-reads the bucket's last pair)
    auto load_round = [&](uint32_t v0) {  // n > 0
      uint32_t ids[NPAIR];
#pragma unroll
      for (int j = 0; j < NPAIR; j++) {
        const uint32_t e = min(v0 + 2u * ((uint32_t)j * WGB + tid), n - 2u);
        while (e >= L.pre[p_ids + 1]) p_ids++;
        // (a pair of a flat source needs no chunk id: it re-reads the table's first word, no branch)
        const uint32_t i0 = e - L.pre[p_ids] + (INC ? (L.skip[p_ids] & ~1u) : 0u);
        const size_t ci = ((size_t)p_ids * gm.P1 + b1) * gm.L1MAX + (i0 >> gm.log2CH1);
        ids[j] = bb.chain1[FL && p_ids >= G ? 0 : ci];
      }
      nxt_valid = 0;
#pragma unroll
      for (int j = 0; j < NPAIR; j++) {
        const uint32_t ev = v0 + 2u * ((uint32_t)j * WGB + tid), e = min(ev, n - 2u);
        while (e >= L.pre[p_rec + 1]) p_rec++;
        const bool flat = FL && p_rec >= G;
        const uint32_t first = (INC && !flat) ? L.skip[p_rec] : 0u;
        const uint32_t i0 = e - L.pre[p_rec] + (first & ~1u), end = L.end[p_rec];
        const uint8_t *src = rec1 + ((((size_t)p_rec * gm.A1 + ids[j]) << gm.log2CH1) + (i0 & CH1m)) * 6;
        if constexpr (FL) {
          const uint64_t f0 = L.flo[flat ? p_rec - G : 0u];
          if (flat) src = reinterpret_cast<const uint8_t *>((uintptr_t)f0) + 5u * (size_t)i0;
          nxt_flat = flat ? (nxt_flat | (1u << j)) : (nxt_flat & ~(1u << j));
        }
        const bool in = ev < n;
        nxt_valid |= (in && i0 >= first && i0 < end ? 1u : 0u) << (2 * j);
        nxt_valid |= (in && i0 + 1u < end ? 2u : 0u) << (2 * j);  // (i0 + 1 >= first always: first <= i0 + 1)
        nxt[j] = *reinterpret_cast<const Load12 *>(src);
      }
    };
    // lo + br: this round's records in the short form; nxt: the next round's, taken over in the middle of the round
    uint32_t lo[RPOS], br[RPOS];
    uint32_t cur_valid = 0;
    const uint32_t sh_b2 = gm.k2 - gm.la - gm.lb;  // where the region's index starts in the mix
    auto take_over = [&]() {
#pragma unroll
      for (int j = 0; j < NPAIR; j++) {
        const Load12 w = nxt[j];
        uint32_t e0, e1;
        if (FL && ((nxt_flat >> j) & 1u)) {  // two wire records: 4 + 1 bytes each
          lo[2 * j] = w.a;
          e0 = w.b & 63u;
          lo[2 * j + 1] = (w.b >> 8) | (w.c << 24);
          e1 = (w.c >> 8) & 63u;
        } else {  // two level-1 records: 4 + 2 bytes each, the extension codes above the bucket's ten bits
          lo[2 * j] = w.a;
          e0 = (w.b >> 10) & 63u;
          lo[2 * j + 1] = (w.b >> 16) | (w.c << 16);
          e1 = w.c >> 26;
        }
        br[2 * j] = ((lo[2 * j] >> sh_b2) & (P2 - 1u)) | (e0 << 10);
        br[2 * j + 1] = ((lo[2 * j + 1] >> sh_b2) & (P2 - 1u)) | (e1 << 10);
      }
      cur_valid = nxt_valid;
    };
    if (n) load_round(0);
    take_over();
    if (per_round < n) load_round(per_round);
    int buf = 0;
    for (uint32_t v0 = 0; v0 < n; v0 += per_round) {
#pragma unroll
      for (int j = 0; j < RPOS; j++) {
        const bool valid = (cur_valid >> j) & 1u;
        const uint32_t rank = hist_rank(L.sp, buf, br[j] & (PMAX - 1), valid);
        br[j] = valid ? (br[j] | (rank << 16)) : ~0u;
      }
      KC_SPLIT_STAMP(0)  // histogram
      lds_barrier();
      const uint32_t total = split_stage<1, RPOS, 16>(
          L.sp, sorted, nullptr, buf, P2,
          [&](int j, uint64_t (&o)[1]) {
            o[0] = ((uint64_t)b1 << (64u - gm.la)) | ((uint64_t)lo[j] << (64u - gm.k2)) | (uint64_t)((br[j] >> 10) & 63u);
          },
          br, D, cst);
      take_over();
      if (v0 + 2 * (uint64_t)per_round < n) load_round(v0 + 2 * per_round);
      KC_SPLIT_STAMP(5)  // take over the next round's records, request the one after
      split_copy_out<1>(
          L.sp, sorted, nullptr, total, D, [&](const uint64_t (&r)[1]) { return cp_b2(r[0], gm); },
          [&](uint32_t b, const uint64_t (&r)[1]) {
            bb.flag[(size_t)b1 * P2 + b] = 1;
            const uint64_t o = atomicAdd((unsigned long long *)&cb[CB_OVF2], 1ULL);
            if (o < bb.ovf2_cap) bb.ovf2[o] = cp_unmix_rec(r[0], gm);
            else atomicOr((unsigned long long *)&cb[CB_FATAL], (unsigned long long)FATAL_OVF2);
          },
          [&](size_t i, const uint64_t (&r)[1]) { reinterpret_cast<uint32_t *>(D.arena)[i] = cp_pack32(r[0], gm); });
      buf ^= 1;
    }
    {
      const int tid = fresh_tid();
      if ((uint32_t)tid < P2) bb.cnt2[(size_t)b1 * P2 + tid] = cst.cur;
      if constexpr (INC) {
        if ((uint32_t)tid < G) bb.done1[(size_t)tid * P1 + b1] = L.end[tid];  // every record of the chain has been to level 2 now
        if (tid == 0) bb.used2[b1] = min(L.sp.arena_used, D.arena_cap);
      }
    }
    __syncthreads();
  }
}

// level-1 overflow records have no region yet (rare path): they join the flagged regions' overflow list and
// their region is flagged, so the region is handled whole by the global table
template <int NL, bool CP>
__global__ void kc_ovf1_to_regions_kernel(Geom gm, BucketBufs bb, uint64_t n, uint64_t *cb) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t r[NL];
  for (int w = 0; w < NL; w++) r[w] = bb.ovf1[i * NL + w];
  size_t reg;
  if (CP) {
    const uint64_t m = cp_mix_rec(r[0], gm);
    reg = (size_t)cp_b1(m, gm) * gm.P2 + cp_b2(m, gm);
  } else {
    const uint64_t h = rec_hash<NL>(r);
    reg = (size_t)hash_b1(h, gm) * gm.P2 + hash_b2(h, gm);
  }
  bb.flag[reg] = 1;
  const uint64_t o = atomicAdd((unsigned long long *)&cb[CB_OVF2], 1ULL);
  if (o < bb.ovf2_cap) {
    for (int w = 0; w < NL; w++) bb.ovf2[o * NL + w] = r[w];
  } else {
    atomicOr((unsigned long long *)&cb[CB_FATAL], (unsigned long long)FATAL_OVF2);
  }
}

// The level-1 overflow list is nearly full while reads are still coming in: its records go to the global table now (their
// regions are flagged, so that the table gets the regions' other records too when the stage is finalized) and the list
// starts again empty.
template <int NL, bool CP>
__global__ __launch_bounds__(TPB) void kc_ovf1_drain_kernel(Geom gm, BucketBufs bb, uint64_t n, Table t, uint64_t *ctrs) {
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
    uint64_t r[NL];
    for (int w = 0; w < NL; w++) r[w] = bb.ovf1[i * NL + w];
    size_t reg;
    if (CP) {
      const uint64_t m = cp_mix_rec(r[0], gm);
      reg = (size_t)cp_b1(m, gm) * gm.P2 + cp_b2(m, gm);
    } else {
      const uint64_t h = rec_hash<NL>(r);
      reg = (size_t)hash_b1(h, gm) * gm.P2 + hash_b2(h, gm);
    }
    bb.flag[reg] = 1;
    table_insert<NL>(t, r, ctrs);
  }
}

// ---- count: the LDS probe window ---------------------------------------------------------------------
// longest region chain (L2MAX) the count kernel can stage.  (480, not 512: with 2048 slots of two-word keys -- 38 bytes a
// slot with the candidate list -- two workgroups still share a CU's 160 KiB, 81744 bytes each; at 512 they missed it by
// 150 bytes and k=33..63 counted in 38 ms instead of 28.)
constexpr uint32_t CHAIN_LDS = 480;

// LDS image of one region, structure-of-arrays so that neighbouring slots sit in neighbouring banks.
// The table size is a power of two (cheap wrap, low load: a lane's probe sequence is as long as the run of occupied
// slots it starts in).  Ten 16-bit counters per entry: the four extensions of either side and, per side,
// the occurrences that came WITHOUT a usable extension -- so that the k-mer's own count is the sum of one side's five
// and need not be counted separately (two LDS adds per occurrence instead of three).  No half can overflow in a region
// of at most 65535 records; a larger one (a k-mer seen more often than that, S6) is handed whole to the global table
// (flag 2), whose counters are 32 bits wide.
// One-word k-mers spend a sixth word on them so that each side has three words of its own: the word and the half an
// extension code e (0-3 = ACGT, 4 = none) bumps are then e >> 1 and e & 1, no selects (the instruction count of an
// occurrence is what bounds the kernel).  Longer keys keep the five-word layout, which is what fits beside them.
template <int NL>
struct CountLDS {  // header at the start of the dynamic LDS; the arrays follow, strided by the region's S
  static constexpr uint32_t SMAX = NL <= 2 ? 4096 : 2048;
  // extension words per slot.  (Compact records have the room for eight, a side's words then picked by the raw 3-bit
  // code with no clamping: four vector instructions less per record and 0.5 ms MORE, profiles/r04_ab_count_eight_words.txt
  // -- more words to sum in the table pass and to clear per slot, a larger table to spread the adds over)
  static constexpr int ew(bool) { return NL == 1 ? 6 : 5; }
  // the region's chunk ids; two buffers: the next region's ids are fetched while this one is counted
  uint32_t chain[2][CHAIN_LDS];
  uint32_t hdr[2][2];  // ... and its length and flag
  uint32_t nout;       // survivors of the region so far (their ranks)
  uint32_t ncand;      // slots of the region whose k-mer was seen twice or more (the list behind the table)
  uint32_t fail[2];    // the region does not fit the table (alternating with the chain buffers)
  uint32_t gbase_lo, gbase_hi, gbase2_lo, gbase2_hi, split;  // ranks < split sit at gbase + rank, the others at gbase2 + (rank - split)
  static constexpr size_t header_bytes() { return (sizeof(CountLDS<NL>) + 15) & ~size_t(15); }
  // cp: compact records, 32-bit keys
  static constexpr size_t bytes(uint32_t S, bool cp) { return header_bytes() + (size_t)S * ((cp ? 4 : 8 * NL) + 4 * ew(cp) + 2) + 16; }
};
static_assert(2 * CountLDS<2>::bytes(2048, false) <= 160 * 1024, "two workgroups of the two-word count kernel per CU");

// the arrays of one region table: word w of slot s at keys[w*S + s] (the LAST word is the claim word; compact
// records: one 32-bit key per slot);
// ext[q*S + s], five words (EW = 5): q = 0 left A|C<<16, 1 left G|T<<16, 2 right A|C<<16, 3 right G|T<<16,
// 4 left none | right none<<16; six words: q = 0 left A|C<<16, 1 left G|T<<16, 2 left none, 3-5 the same of the right
struct CountTab {
  uint64_t *keys;
  uint32_t *ext;
  uint16_t *cand;  // S entries: the slots the vote has to look at
  uint32_t S, lgS;
};

template <int NL, bool CP>
__device__ __forceinline__ CountTab count_tab(uint8_t *smem, uint32_t S) {
  CountTab t;
  uint8_t *p = smem + CountLDS<NL>::header_bytes();
  t.keys = reinterpret_cast<uint64_t *>(p);
  p += (size_t)S * (CP ? 4 : 8 * NL);
  t.ext = reinterpret_cast<uint32_t *>(p);
  p += (size_t)S * 4 * CountLDS<NL>::ew(CP);
  t.cand = reinterpret_cast<uint16_t *>(p);
  t.S = S;
  t.lgS = 31u - (uint32_t)__clz(S);
  return t;
}

// one occurrence with extension codes le, re (0-3 = ACGT, >= 4 = none) at slot s: S5/S6 in two LDS adds
template <int EW>
__device__ __forceinline__ void ext_count(const CountTab &tb, uint32_t s, uint32_t le, uint32_t re) {
  if constexpr (EW == 6) {
    le = min(le, 4u);  // any code >= 4 means "none"
    re = min(re, 4u);
    atomicAdd(&tb.ext[((le >> 1) << tb.lgS) + s], 1u << ((le & 1u) << 4));
    atomicAdd(&tb.ext[((3u + (re >> 1)) << tb.lgS) + s], 1u << ((re & 1u) << 4));
  } else {
    const bool nl = (le & 4u) != 0, nr = (re & 4u) != 0;
    const uint32_t wl = nl ? 4u : (le >> 1), wr = nr ? 4u : 2u + (re >> 1);
    const uint32_t il = nl ? 1u : 1u << ((le & 1u) << 4), ir = nr ? 0x10000u : 1u << ((re & 1u) << 4);
    atomicAdd(&tb.ext[(wl << tb.lgS) + s], il);
    atomicAdd(&tb.ext[(wr << tb.lgS) + s], ir);
  }
}
template <int EW>
__device__ __forceinline__ uint32_t ext_get(const uint32_t *ext, uint32_t SM, uint32_t s, uint32_t side, uint32_t e) {
  return (ext[((EW == 6 ? 3 : 2) * side + (e >> 1)) * SM + s] >> (16 * (e & 1u))) & 0xFFFFu;
}

struct OutBufs {
  uint64_t *keys;
  uint16_t *counts;
  uint8_t *left, *right;
  uint16_t *exts;    // DUMP only: 8 per entry
  uint64_t cap;
  uint64_t *cursor;  // global append position (results: &ctrs[CTR_OUT])
  // block > 0: a workgroup takes its output positions from private blocks of `block` entries and bumps the cursor
  // only when a block is used up (a bump per region makes every workgroup of the chip queue on one address: measured
  // 16 of the count kernel's 42 ms).  A region's survivors are ranked while they are written (one pass over the table,
  // no count-then-reserve), so before each region the workgroup makes sure it holds room for as many as the region can
  // have at most: its current block and, when that has less left, a spare one.  Each workgroup leaves the unused tails
  // of the (at most two) blocks it holds at the end in tails[4*wg .. 4*wg+3], and kc_out_plan_kernel /
  // kc_out_move_kernel then close those holes with entries from the end of the arrays.
  uint32_t block;
  uint64_t *tails;   // [2 * 2 * workgroups]: start and length of every hole
};

// One-word keys: the probe written for a low instruction count.  The region kernels are bound by instruction
// issue (PMC: as many scalar as vector instructions, almost no idle LDS or HBM), and the lanes of a wave probe in
// lock step, so every trip of this loop is paid by all 64 lanes.  Predicates are kept as 0/1 integers in vector
// registers (compares produce lane masks in scalar registers, and combining those is scalar work); each trip is one
// LDS compare-and-swap EMPTY -> key whose return value says "claimed" (EMPTY), "already there" (key) or "someone
// else"; lanes that are done keep issuing a CAS that can never match (expected value KEY_BUSY never occurs in a
// one-word table), i.e. a plain read.  Returns the slot; counts nothing (the caller bumps the counters, and the
// number of entries is counted when the table is scanned).
__device__ __forceinline__ uint32_t lds_probe1(unsigned long long *claim, uint32_t Sm1, uint64_t key, uint32_t slot, uint32_t valid,
                                               uint32_t &failed) {
  uint32_t act = valid;
  const uint32_t klo = (uint32_t)key, khi = (uint32_t)(key >> 32);
  uint32_t trips = 0;
  do {
    unsigned long long old = key;  // a lane that is done looks like a hit below
    if (act) old = atomicCAS(&claim[slot], (unsigned long long)KEY_EMPTY, (unsigned long long)key);  // only live lanes touch the LDS
    const uint32_t olo = (uint32_t)old, ohi = (uint32_t)(old >> 32);
    const uint32_t differs = (olo ^ klo) | (ohi ^ khi);  // 0 iff the slot holds this key
    const uint32_t notempty = ~(olo & ohi);              // 0 iff the slot was EMPTY (and is now ours)
    const uint32_t miss = min(differs, notempty) ? act : 0u;
    act = miss;
    slot = (slot + miss) & Sm1;
    if (++trips > Sm1 + 1) {  // every slot holds some other k-mer (wave-uniform exit)
      failed |= act;
      break;
    }
  } while (__any(act));
  return slot;
}

// The same probe on 32-bit keys (compact records): a 32-bit LDS compare-and-swap runs at about three times the rate
// of a 64-bit one.  EMPTY is all ones, which no key is (a key has at most 26 bits).  (Reading the slot first and
// claiming only an empty one -- reads of one address share an access, compare-and-swaps of one address queue -- was
// measured 6 % slower: the extra dependent LDS trip of the new keys costs more than the queueing.)
// The probe sequence steps by an odd stride taken from the key's bits above the start slot (every slot is visited once
// in S steps, and keys that collide on a slot part ways at once): the lanes of a wave probe in lock step, a wave pays
// for its longest probe, and without the clusters of a unit stride the longest of 64 is shorter.
__device__ __forceinline__ uint32_t lds_probe32(uint32_t *claim, uint32_t Sm1, uint32_t key, uint32_t slot, uint32_t stride, bool valid,
                                                uint32_t &failed) {
  // slot and step as byte offsets.  The loop is written out: every instruction of a trip is paid by the whole wave, the
  // count kernel is bound by instruction issue (203 instructions per 64 records at 2.75 cycles each, §5), and what the
  // compiler makes of the same loop in C++ is 28 instructions per trip (the lanes that are done kept out of the LDS by
  // saving and restoring the execution mask around the compare-and-swap, the "anybody left?" test through a vector
  // register and back).  Here the execution mask itself is the set of lanes still probing: it only shrinks, a lane's
  // offset stops moving the moment the lane drops out, and a trip is 13 instructions.
#ifndef KC_PROBE_CXX
  uint32_t at = slot << 2, old, addr, trips = Sm1 + 2;
  const uint32_t m4 = Sm1 << 2, step = stride << 2;
  const uint32_t base = (uint32_t)(uintptr_t)claim;  // the array's LDS address (low half of the flat one)
  const uint64_t live = __builtin_amdgcn_ballot_w64(valid);  // the lanes that hold a record (a compare's result: already a lane mask)
  uint64_t sv, t;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_and_b64 exec, exec, %[live]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_add_u32_e32 %[addr], %[base], %[at]\n\t"
      "ds_cmpst_rtn_b32 %[old], %[addr], %[empty], %[key]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_ne_u32_e32 vcc, %[old], %[key]\n\t"    // not this key
      "v_cmp_ne_u32_e64 %[t], -1, %[old]\n\t"       // and not (until now) empty
      "s_and_b64 vcc, vcc, %[t]\n\t"
      "s_and_b64 exec, exec, vcc\n\t"               // the lanes that go on
      "s_cbranch_execz 2f\n\t"
      "v_add_u32_e32 %[at], %[step], %[at]\n\t"
      "v_and_b32_e32 %[at], %[m4], %[at]\n\t"
      "s_sub_u32 %[trips], %[trips], 1\n\t"
      "s_cmp_lg_u32 %[trips], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "v_mov_b32_e32 %[fl], 1\n"                    // every slot holds some other k-mer
      "2:\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [old] "=&v"(old), [addr] "=&v"(addr), [at] "+v"(at), [fl] "+v"(failed), [trips] "+s"(trips), [sv] "=&s"(sv), [t] "=&s"(t)
      : [live] "s"(live), [base] "s"(base), [empty] "v"(0xFFFFFFFFu), [key] "v"(key), [step] "v"(step), [m4] "s"(m4)
      : "vcc", "scc", "memory");
  return at >> 2;
#else
  uint32_t trips = 0, at = slot << 2;
  const uint32_t m4 = Sm1 << 2, step = stride << 2;
  bool go = valid;  // this lane is still probing (a lane mask in scalar registers)
  do {
    uint32_t old = key;  // a lane that is done looks like a hit below
    if (go) old = atomicCAS(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(claim) + at), 0xFFFFFFFFu, key);
    go = go & (old != key) & (old != 0xFFFFFFFFu);  // neither this key nor (until now) empty
    at = (at + (go ? step : 0u)) & m4;
    if (++trips > Sm1 + 1) {  // every slot holds some other k-mer (wave-uniform exit)
      failed |= go ? 1u : 0u;
      break;
    }
  } while (__builtin_amdgcn_ballot_w64(go) != 0);
  return at >> 2;
#endif
}

template <int NL>
__device__ __forceinline__ uint32_t lds_probeN(const CountTab &tb, const uint64_t (&key)[NL], uint32_t slot, uint32_t valid, uint32_t &failed);

// Two-word keys, the probe written out like lds_probe32: the execution mask is the set of lanes still probing.  A trip:
// the compare-and-swap of the claim word (EMPTY -> BUSY) and, right behind it, the read of the other word (the LDS serves a
// wave's requests in order: a lane that finds the key's last word published reads a first word written before the
// publication); one wait; the winners write their first word and publish; "same" = both words equal; a lane that found
// BUSY looks at the same slot again; everybody else moves on by one slot.  24 instructions a trip against the 60-odd the
// compiler makes of lds_probeN<2> (exec saved and restored around every conditional LDS access, every predicate through
// a vector register).  Returns the slot; failed |= 1 when every slot holds some other k-mer.
__device__ __forceinline__ uint32_t lds_probe2(const CountTab &tb, uint64_t key0, uint64_t klast, uint32_t slot, bool valid, uint32_t &failed) {
#ifndef KC_PROBE2_CXX
  const uint32_t m8 = (tb.S - 1u) << 3;
  uint32_t at = slot << 3, a0, a1, trips = 4u * tb.S + 2u;
  const uint32_t base0 = (uint32_t)(uintptr_t)tb.keys, base1 = base0 + (tb.S << 3);  // word 0 of every slot, then the claim words
  const uint64_t live = __builtin_amdgcn_ballot_w64(valid);
  const uint64_t empty = KEY_EMPTY, busy = KEY_BUSY;
  uint64_t old, w0, sv, act, won, t1, t2;
  asm volatile(
      "s_mov_b64 %[sv], exec\n\t"
      "s_and_b64 %[act], exec, %[live]\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "s_cbranch_execz 2f\n"
      "1:\n\t"
      "v_add_u32_e32 %[a1], %[base1], %[at]\n\t"
      "v_add_u32_e32 %[a0], %[base0], %[at]\n\t"
      "ds_cmpst_rtn_b64 %[old], %[a1], %[empty], %[busy]\n\t"
      "ds_read_b64 %[w0], %[a0]\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "v_cmp_eq_u64_e64 %[won], %[old], %[empty]\n\t"
      "s_and_b64 exec, %[act], %[won]\n\t"          // the lanes that claimed a slot: first word, then the publication
      "ds_write_b64 %[a0], %[key0]\n\t"
      "ds_write_b64 %[a1], %[klast]\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "v_cmp_eq_u64_e64 %[t1], %[old], %[klast]\n\t"
      "v_cmp_eq_u64_e64 %[t2], %[w0], %[key0]\n\t"
      "s_and_b64 %[t1], %[t1], %[t2]\n\t"            // the slot holds this key
      "s_or_b64 %[t1], %[t1], %[won]\n\t"
      "s_andn2_b64 %[act], %[act], %[t1]\n\t"        // the lanes that go on
      "s_mov_b64 exec, %[act]\n\t"
      "s_cbranch_execz 2f\n\t"
      "v_cmp_ne_u64_e64 %[t2], %[old], %[busy]\n\t"  // (a slot being written: look again; any other: next slot)
      "s_and_b64 exec, %[act], %[t2]\n\t"
      "v_add_u32_e32 %[at], 8, %[at]\n\t"
      "v_and_b32_e32 %[at], %[m8], %[at]\n\t"
      "s_mov_b64 exec, %[act]\n\t"
      "s_sub_u32 %[trips], %[trips], 1\n\t"
      "s_cmp_lg_u32 %[trips], 0\n\t"
      "s_cbranch_scc1 1b\n\t"
      "v_mov_b32_e32 %[fl], 1\n"                      // every slot holds some other k-mer
      "2:\n\t"
      "s_mov_b64 exec, %[sv]\n\t"
      : [old] "=&v"(old), [w0] "=&v"(w0), [a0] "=&v"(a0), [a1] "=&v"(a1), [at] "+v"(at), [fl] "+v"(failed), [trips] "+s"(trips), [sv] "=&s"(sv),
        [act] "=&s"(act), [won] "=&s"(won), [t1] "=&s"(t1), [t2] "=&s"(t2)
      : [live] "s"(live), [base0] "s"(base0), [base1] "s"(base1), [empty] "v"(empty), [busy] "v"(busy), [key0] "v"(key0), [klast] "v"(klast),
        [m8] "s"(m8)
      : "vcc", "scc", "memory");
  return at >> 3;
#else
  const uint64_t key[2] = {key0, klast};
  return lds_probeN<2>(tb, key, slot, valid ? 1u : 0u, failed);
#endif
}

// Several-word keys, same style: the last word is the claim word (EMPTY -> BUSY -> the key's last word), the others
// are written between the claim and its publication.  A lane that finds BUSY tries the same slot again (its owner
// publishes within this same trip of its own wave); a lane that finds its last word compares the other words.
template <int NL>
__device__ __forceinline__ uint32_t lds_probeN(const CountTab &tb, const uint64_t (&key)[NL], uint32_t slot, uint32_t valid, uint32_t &failed) {
  const uint32_t SM = tb.S, Sm1 = tb.S - 1u;
  unsigned long long *claim = (unsigned long long *)&tb.keys[(NL - 1) * SM];
  const unsigned long long klast = key[NL - 1];
  uint32_t act = valid, trips = 0;
  do {
    unsigned long long old;
    uint32_t won, same;
    if constexpr (NL >= 3) {
      // The claim and the other words in one LDS round trip: the compare-and-swap goes out for every lane (a lane that is
      // done expects a value no claim word ever holds, which makes it a plain read) and the reads of the other words
      // right behind it -- the LDS serves a wave's requests in order, so a lane that finds the key published reads words
      // that were written before the publication.  (With the reads behind the winners' writes, a trip was two trips to
      // the LDS.)
      old = atomicCAS(&claim[slot], act ? (unsigned long long)KEY_EMPTY : (unsigned long long)KEY_NEVER, (unsigned long long)KEY_BUSY);
      uint64_t other[NL > 1 ? NL - 1 : 1];
#pragma unroll
      for (int j = 0; j < NL - 1; j++) other[j] = tb.keys[j * SM + slot];
      won = (act && old == KEY_EMPTY) ? 1u : 0u;
      if (won) {
#pragma unroll
        for (int j = 0; j < NL - 1; j++) tb.keys[j * SM + slot] = key[j];
        __threadfence_block();  // the words are in place before the claim word says so
        atomicExch(&claim[slot], klast);
      }
      same = (old == klast || !act) ? 1u : 0u;
#pragma unroll
      for (int j = 0; j < NL - 1; j++) same &= (other[j] == key[j] || !act) ? 1u : 0u;
    } else {
      // (two-word keys: the reads every lane then issues on every trip cost more than the round trip they save: k=51
      // count 25.4 -> 27.4 ms; three words: k=77 41.4 -> 39.7 ms, profiles/r03_ab_probe_words_with_claim.txt)
      old = klast;  // a lane that is done looks like a hit on its own key below
      if (act) old = atomicCAS(&claim[slot], (unsigned long long)KEY_EMPTY, (unsigned long long)KEY_BUSY);
      won = (act && old == KEY_EMPTY) ? 1u : 0u;
      if (won) {
#pragma unroll
        for (int j = 0; j < NL - 1; j++) tb.keys[j * SM + slot] = key[j];
        __threadfence_block();  // the words are in place before the claim word says so
        atomicExch(&claim[slot], klast);
      }
      same = old == klast ? 1u : 0u;
#pragma unroll
      for (int j = 0; j < NL - 1; j++) same &= (tb.keys[j * SM + slot] == key[j]) ? 1u : 0u;
    }
    const uint32_t busy = old == KEY_BUSY ? 1u : 0u;
    const uint32_t hit = won | same;
    const uint32_t miss = act & ~hit & ~busy & 1u;
    act &= ~hit & 1u;
    slot = (slot + miss) & Sm1;
    if (++trips > 4u * (Sm1 + 1u)) {  // every slot holds some other k-mer (wave-uniform exit)
      failed |= act;
      break;
    }
  } while (__any(act));
  return slot;
}

// DUMP = false: S7 vote + S8 purge, survivors to the result arrays.  DUMP = true: every entry with its raw
// (clipped) counters, for tests of S5/S6; no statistics are touched.
// __launch_bounds__(WGB, 8): two 1024-thread workgroups per CU need at most 64 registers per lane.
// CP: compact 32-bit records (cp_pack32); the table keys are their upper 26 bits, the k-mer is rebuilt from the
// region and the key when an entry is written out.
//
// A region costs the workgroup three barriers: [inserts] barrier [one pass over the table: purge the k-mers seen once,
// list the others] barrier [the list: vote, rank, write out, clean the slot] barrier.  Everything a region needs from memory before its first record -- its length, its flag, its
// chunk ids -- is fetched one region ahead (registers, then the other half of T.chain), and one word of every 128-byte
// line of the next region's records is touched before the table pass, so that the record loads after the barrier
// find their lines in the L2 instead of waiting for HBM with nothing else to do.  (Round 2 listed the occupied slots,
// voted over the list, reserved the output in a third step and wrote in a fourth: seven barriers, 40 % of the kernel's
// time in those short dependent phases.)
constexpr uint32_t CHAIN_PRE = 64;  // chunk ids fetched ahead (a region of up to 65535 records has at most 64 chunks of 1024)
template <int NL, bool DUMP, bool CP>
__global__ __launch_bounds__(WGB, 8) void kc_count_kernel(Geom gm, BucketBufs bb, OutBufs out, int dmin_thres, uint64_t *ctrs,
                                                       uint64_t *cb) {
  extern __shared__ __align__(16) uint8_t smem[];
  CountLDS<NL> &T = *reinterpret_cast<CountLDS<NL> *>(smem);
  const int tid = threadIdx.x;
  const uint32_t S = gm.S, SM = gm.S;
  constexpr int EW = CountLDS<NL>::ew(CP);
  const CountTab tb = count_tab<NL, CP>(smem, S);
  const size_t R = (size_t)gm.P1 * gm.P2;
  // diagnostic builds only (-DKC_STAMPS): thread 0 accumulates the cycles between the phase boundaries of every region
  // into cb[8..]; the shipped build has no stamp code at all
#ifdef KC_STAMPS
  unsigned long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memtime();
  const bool stamp = tid == 0;
#define KC_STAMP(k)                                             \
  if (stamp) {                                                  \
    const unsigned long long tn = __builtin_amdgcn_s_memtime(); \
    tacc[k] += tn - tprev;                                      \
    tprev = tn;                                                 \
  }
#else
#define KC_STAMP(k)
#endif
  // the table is cleared once; every region leaves it clean again by resetting exactly the slots it claimed
  uint32_t *const keys32 = reinterpret_cast<uint32_t *>(tb.keys);  // CP: the key array holds 32-bit keys
  auto reset_slot = [&](uint32_t s) {
    if (CP) {
      keys32[s] = 0xFFFFFFFFu;
    } else {
#pragma unroll
      for (int w = 0; w < NL; w++) tb.keys[w * SM + s] = KEY_EMPTY;
    }
#pragma unroll
    for (int e = 0; e < EW; e++) tb.ext[e * SM + s] = 0;
  };
  for (uint32_t s = tid; s < S; s += WGB) reset_slot(s);
  if (tid == 0) {
    T.nout = 0;
    T.ncand = 0;
    T.fail[0] = T.fail[1] = 0;
  }
  // thread 0: the output blocks this workgroup holds -- the one it is filling and a spare
  uint64_t cur_base = 0, cur_left = 0, sp_base = 0, sp_left = 0;
  bool pending = false;  // thread 0: the last counted region's survivors are not yet taken off the blocks
  auto settle = [&]() {  // thread 0, after the barrier that ends a region's table pass
    if (!pending) return;
    pending = false;
    uint64_t need = T.nout;
    T.nout = 0;
    if (need <= cur_left) {
      cur_base += need;
      cur_left -= need;
    } else {  // the current block is used up: the spare becomes the current one
      need -= cur_left;
      cur_base = sp_base + need;
      cur_left = sp_left - need;
      sp_base = sp_left = 0;
    }
  };
  // every lane's share of the statistics (flushed once, at the end)
  uint32_t acc_entries = 0, acc_kept = 0;
  unsigned long long acc_sum = 0;
  uint32_t sink = 0;  // what the look-ahead touches end up in (never zero-tested before the end)
  // ---- look-ahead state: region r's header is in registers / T.chain[buf] when its iteration starts ----
  // (vector loads by a few lanes, parked in LDS: a scalar load's wait would also be a wait for every LDS operation in
  // flight, and the first one of the insert loop would then sit out the scalar load's latency)
  size_t r = blockIdx.x;
  int buf = 0;
  const uint32_t npre = min(CHAIN_PRE, gm.L2MAX);
  auto header_word = [&](size_t reg) -> uint32_t {  // lane t < npre: chunk id t; lane npre: the length; lane npre + 1: the flag
    const uint32_t *src = (uint32_t)tid < npre ? bb.chain2 + reg * gm.L2MAX + tid : (uint32_t)tid == npre ? bb.cnt2 + reg : bb.flag + reg;
    return *src;
  };
  auto header_park = [&](int b, uint32_t v) {
    if ((uint32_t)tid < npre) T.chain[b][tid] = v;
    else T.hdr[b][tid - npre] = v;
  };
  if (r < R && (uint32_t)tid < npre + 2u) header_park(0, header_word(r));
  __syncthreads();
  const uint32_t CHm = (1u << gm.log2CH2) - 1u;
  while (r < R) {
    const size_t rn = r + gridDim.x;
    const bool has_next = rn < R;
    // the next region's header: requested now, parked before this iteration's barrier
    uint32_t c_nxt = 0;
    if (has_next && (uint32_t)tid < npre + 2u) c_nxt = header_word(rn);
    const uint32_t n = T.hdr[buf][0], f_cur = T.hdr[buf][1];
    const bool process = n != 0 && f_cur == 0 && n <= KC_COUNT_MAX;  // uniform across the workgroup
    if (n > KC_COUNT_MAX && f_cur == 0 && tid == 0) bb.flag[r] = 2;  // a 16-bit counter could overflow: the global table takes the region
    if (process) {
      if (tid == 0) {
        settle();
        T.ncand = 0;  // (the last region's list was read out before the barrier that ended it)
        // room for as many survivors as this region can have: count >= 2 each, and no more than slots
        const uint64_t most = DUMP ? min(n, S) : min(n >> 1, S);
        if (out.block != 0) {
          if (cur_left + sp_left < most) {  // (then there is no spare: a spare, S entries or more and untouched, always suffices)
            const uint64_t take = ((uint64_t)S + out.block - 1) / out.block * out.block;
            sp_base = atomicAdd((unsigned long long *)out.cursor, (unsigned long long)take);
            sp_left = take;
          }
          T.gbase_lo = (uint32_t)cur_base;
          T.gbase_hi = (uint32_t)(cur_base >> 32);
          T.gbase2_lo = (uint32_t)sp_base;
          T.gbase2_hi = (uint32_t)(sp_base >> 32);
          T.split = (uint32_t)min(cur_left, (uint64_t)0xFFFFFFFFu);
        }
        pending = true;
      }
      const uint32_t nch = (n + CHm) >> gm.log2CH2;
      if (nch > npre) {  // (chunks far smaller than a region: tiny geometries only)
        for (uint32_t i = npre + tid; i < nch; i += WGB) T.chain[buf][i] = bb.chain2[r * gm.L2MAX + i];
        __syncthreads();
      }
      KC_STAMP(0)  // header
      // several independent loads in flight per thread before the dependent LDS work starts
      // (two: with the next region's lines touched into the L2 a region ahead the loads need no depth, and eight records
      // in registers cost spilled registers -- 19.1 -> 18.6 ms, profiles/r03_ab_count_batch.txt)
#ifndef KC_BATCH
#define KC_BATCH 2
#endif
      constexpr int BATCH = NL == 1 ? KC_BATCH : NL == 2 ? 4 : 2;
#ifndef KC_COUNT_NO_PAIRS
      if constexpr (NL == 1 && CP) {
        // Compact records: a lane takes TWO neighbouring records with one 8-byte load (the order in which a region's
        // records are counted is free): one chunk id, one address, one load instruction for two records.  A chunk holds an
        // even number of records, so the pair lies in one chunk.
        uint32_t failed = 0;
        for (uint32_t i0 = 0; i0 < n; i0 += 2u * WGB) {
          uint32_t i = i0 + 2u * (uint32_t)tid;
          const bool v0 = i < n, v1 = i + 1u < n;
          i = v0 ? i : 0u;
          const size_t at = ((size_t)T.chain[buf][i >> gm.log2CH2] << gm.log2CH2) + (i & CHm);
          const uint2 rr = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint32_t *>(bb.rec2) + at);
          if (__builtin_amdgcn_ballot_w64(v0) == 0 || KC_ABL(gm, 2)) continue;  // past the end of the region for the whole wave
#pragma unroll
          for (int j = 0; j < 2; j++) {
            const bool v = j ? v1 : v0;
            const uint32_t r0 = j ? rr.y : rr.x, key = r0 >> 6;
            const uint32_t s = lds_probe32(keys32, S - 1u, key, key & (S - 1u), ((key >> tb.lgS) << 1) | 1u, v, failed);
            const uint32_t le = r0 & 7u, re = (r0 >> 3) & 7u;
            // only lanes that hold a record touch the counters: 64 atomic adds of zero to one LDS word are serialised
            if (v && !KC_ABL(gm, 1)) ext_count<EW>(tb, s, le, re);
          }
        }
        if (failed) T.fail[buf] = 1;
      } else
#endif
      for (uint32_t i0 = 0; i0 < n; i0 += WGB * BATCH) {
        uint64_t rec[BATCH][NL];
        uint32_t rec32[BATCH];
        // no branches around the loads (an out-of-range lane re-reads record 0): the compiler can then issue
        // all of them before the first wait instead of fencing each one off in its own basic block
        size_t at[BATCH];
#pragma unroll
        // (taking a wave's chunk id into a scalar register -- its 64 records lie in one chunk -- and adding the lanes'
        // offsets to a scalar address saves five vector instructions per record and cost 4 ms: profiles/r03_ab_count_*)
        for (int j = 0; j < BATCH; j++) {
          uint32_t i = i0 + (uint32_t)j * WGB + tid;
          i = i < n ? i : 0u;
          at[j] = ((size_t)T.chain[buf][i >> gm.log2CH2] << gm.log2CH2) + (i & CHm);
        }
#pragma unroll
        for (int j = 0; j < BATCH; j++) {
          if (CP) {
            rec32[j] = reinterpret_cast<const uint32_t *>(bb.rec2)[at[j]];
          } else {
#pragma unroll
            for (int w = 0; w < NL; w++) rec[j][w] = bb.rec2[at[j] * NL + w];
          }
        }
        if constexpr (NL == 1 && CP) {
          uint32_t failed = 0;
#pragma unroll
          for (int j = 0; j < BATCH; j++) {
            const bool v = (i0 + (uint32_t)j * WGB + tid) < n;
            const uint32_t r0 = rec32[j], key = r0 >> 6;
            if (__builtin_amdgcn_ballot_w64(v) == 0 || KC_ABL(gm, 2)) continue;  // past the end of the region for the whole wave
            const uint32_t s = lds_probe32(keys32, S - 1u, key, key & (S - 1u), ((key >> tb.lgS) << 1) | 1u, v, failed);
            const uint32_t le = r0 & 7u, re = (r0 >> 3) & 7u;
            // only lanes that hold a record touch the counters: the idle lanes of a wave all re-read record 0, and 64
            // atomic adds of zero to one LDS word are serialised
            if (v && !KC_ABL(gm, 1)) ext_count<EW>(tb, s, le, re);
          }
          if (failed) T.fail[buf] = 1;
        } else if constexpr (NL == 1) {
          uint32_t failed = 0;
#pragma unroll
          for (int j = 0; j < BATCH; j++) {
            const uint32_t v = (i0 + (uint32_t)j * WGB + tid) < n ? 1u : 0u;
            const uint64_t r0 = rec[j][0], key = r0 & ~KC_EXT_MASK;
            uint64_t kk[1] = {key};
            if (!__any(v)) continue;  // past the end of the region for the whole wave
            const uint32_t s = lds_probe1((unsigned long long *)tb.keys, S - 1u, key, hash_slot(kc_hash<1>(kk), S), v, failed);
            // S5/S6: the k-mer itself, and each extension that is an ACGT base (codes 0-3; 4 = none adds 0).  Only
            // lanes that hold a record touch the counters (see the compact path above)
            const uint32_t le = (uint32_t)r0 & 7u, re = ((uint32_t)r0 >> 3) & 7u;
            if (v) ext_count<EW>(tb, s, le, re);
          }
          if (failed) T.fail[buf] = 1;
        } else {
          uint32_t failed = 0;
#pragma unroll
          for (int j = 0; j < BATCH; j++) {
            const uint32_t v = (i0 + (uint32_t)j * WGB + tid) < n ? 1u : 0u;
            if (!__any(v)) continue;  // past the end of the region for the whole wave
            uint64_t key[NL];
#pragma unroll
            for (int w = 0; w < NL; w++) key[w] = rec[j][w];
            const uint32_t le = (uint32_t)(rec[j][NL - 1] & 7u), re = (uint32_t)((rec[j][NL - 1] >> 3) & 7u);
            key[NL - 1] &= ~KC_EXT_MASK;
            uint32_t s;
            if constexpr (NL == 2) s = lds_probe2(tb, key[0], key[1], hash_slot(kc_hash<NL>(key), S), v != 0u, failed);
            else s = lds_probeN<NL>(tb, key, hash_slot(kc_hash<NL>(key), S), v, failed);
            if (v && !(failed & 1u)) ext_count<EW>(tb, s, le, re);
          }
          if (failed) T.fail[buf] = 1;
        }
      }
    }
    // the next region's header moves to the other half of T.chain / T.hdr (nobody reads that half before the barrier)
    if (has_next && (uint32_t)tid < npre + 2u) header_park(buf ^ 1, c_nxt);
    __syncthreads();
    KC_STAMP(1)  // loads + inserts + barrier
    if (process) {
      // touch the next region's records, one word per 128-byte line, so that its loads after the next barrier hit the L2
      uint32_t touch = 0;
      if (has_next) {
        const uint32_t n_nxt = T.hdr[buf ^ 1][0], f_nxt = T.hdr[buf ^ 1][1];
        constexpr uint32_t PER_LINE = CP ? 32u : 16u / NL;  // records per 128 bytes
        const uint32_t i = (uint32_t)fresh_tid() * PER_LINE;  // (recomputed: hoisted out of the region loop the address was spilled)
        if (f_nxt == 0 && n_nxt <= KC_COUNT_MAX && i < n_nxt && (i >> gm.log2CH2) < npre) {
          const size_t a = ((size_t)T.chain[buf ^ 1][i >> gm.log2CH2] << gm.log2CH2) + (i & CHm);
          touch = CP ? reinterpret_cast<const uint32_t *>(bb.rec2)[a] : (uint32_t)bb.rec2[a * NL];
        }
      }
      const bool failed = T.fail[buf] != 0;  // more distinct k-mers than slots: the whole region goes to the global table instead
      const uint64_t gbase = ((uint64_t)T.gbase_hi << 32) | T.gbase_lo, gbase2 = ((uint64_t)T.gbase2_hi << 32) | T.gbase2_lo;
      const uint32_t split = T.split;
      // First the table, slot by slot: a k-mer seen once is purged on the spot (S8) -- five slots of six on reads with
      // half a percent of errors -- and the slots seen twice or more are listed.  Then S7's vote over the LIST: a wave
      // pays for the vote as soon as one of its lanes needs it, and with every sixth slot needing it that was every wave
      // of the table (the vote and the write were 6 of the kernel's 23 ms); the list keeps the vote to the first few
      // waves.  The survivors of a wave take their ranks from ONE add to the region's counter and are written straight
      // to their places; every visited slot is left clean.
      // (two slots per thread and trip, every LDS read of both requested before the first is used and no branch around
      // them -- an empty slot's counters are zero anyway --, one bump of the list's counter per wave and trip: the pass is a
      // chain of LDS round trips, KC_STAMPS puts it at 37 % of the kernel's time for a dozen instructions a slot)
      for (uint32_t s0 = 0; s0 < S; s0 += 2 * WGB) {
        uint32_t sx[2];
        bool in[2], taken[2], cand[2];
        uint32_t w[2][3];
#pragma unroll
        for (int u = 0; u < 2; u++) {
          sx[u] = s0 + (uint32_t)u * WGB + tid;
          in[u] = sx[u] < S;
          const uint32_t si = in[u] ? sx[u] : 0u;
          taken[u] = in[u] && (CP ? keys32[si] != 0xFFFFFFFFu : tb.keys[(NL - 1) * SM + si] != KEY_EMPTY);
          w[u][0] = tb.ext[si];
          w[u][1] = tb.ext[SM + si];
          w[u][2] = tb.ext[(EW == 5 ? 4 : 2) * SM + si];
        }
#pragma unroll
        for (int u = 0; u < 2; u++) {
          cand[u] = false;
          if (taken[u] && !failed && !KC_ABL(gm, 3)) {
            // the k-mer's count: every occurrence bumped exactly one of the left side's counters (<= 65535: n is)
            const uint32_t count = (w[u][0] & 0xFFFFu) + (w[u][0] >> 16) + (w[u][1] & 0xFFFFu) + (w[u][1] >> 16) + (w[u][2] & 0xFFFFu);
            cand[u] = DUMP || count >= 2;
            acc_entries++;
          }
        }
        const uint64_t m0 = __ballot(cand[0]), m1 = __ballot(cand[1]);
        if (m0 | m1) {  // wave-uniform
          const uint32_t n0 = (uint32_t)__popcll(m0);
          uint32_t base = 0;
          if (lane_id() == 0) base = atomicAdd(&T.ncand, n0 + (uint32_t)__popcll(m1));
          base = __shfl(base, 0);
          const uint64_t below = (1ULL << lane_id()) - 1ULL;
          if (cand[0]) tb.cand[base + (uint32_t)__popcll(m0 & below)] = (uint16_t)sx[0];
          if (cand[1]) tb.cand[base + n0 + (uint32_t)__popcll(m1 & below)] = (uint16_t)sx[1];
        }
#pragma unroll
        for (int u = 0; u < 2; u++)
          if (taken[u] && !cand[u]) reset_slot(sx[u]);  // leave the table clean for the next region
      }
      __syncthreads();
      const uint32_t ncand = T.ncand;
      for (uint32_t i0 = 0; i0 < ncand; i0 += WGB) {
        const uint32_t i = i0 + tid;
        const bool live = i < ncand;
        const uint32_t s = live ? tb.cand[i] : 0u;
        bool keep = false;
        uint32_t count = 0, l = 0, rr = 0;
        uint32_t w[EW];
        if (live) {
#pragma unroll
          for (int x = 0; x < EW; x++) w[x] = tb.ext[x * SM + s];
          count = (w[0] & 0xFFFFu) + (w[0] >> 16) + (w[1] & 0xFFFFu) + (w[1] >> 16) + (w[EW == 5 ? 4 : 2] & 0xFFFFu);
          if (DUMP) {
            keep = true;
          } else {
            const uint32_t lc[4] = {w[0] & 0xFFFFu, w[0] >> 16, w[1] & 0xFFFFu, w[1] >> 16};
            constexpr int R0 = EW == 6 ? 3 : 2;
            const uint32_t rc[4] = {w[R0] & 0xFFFFu, w[R0] >> 16, w[R0 + 1] & 0xFFFFu, w[R0 + 1] >> 16};
            l = vote_ext(lc, count, dmin_thres);
            rr = vote_ext(rc, count, dmin_thres);
            keep = l < 4u && rr < 4u;
          }
        }
        const uint64_t m = __ballot(keep);
        if (m) {  // wave-uniform
          const int leader = __ffsll((long long)m) - 1;
          const uint32_t cnt = (uint32_t)__popcll(m);
          uint64_t o = 0;
          if (out.block != 0) {
            uint32_t base = 0;
            if ((int)lane_id() == leader) base = atomicAdd(&T.nout, cnt);
            base = __shfl(base, leader);
            const uint32_t rank = base + (uint32_t)__popcll(m & ((1ULL << lane_id()) - 1ULL));
            o = rank < split ? gbase + rank : gbase2 + (rank - split);
          } else {  // no blocks (dumps, tiny result sets): positions straight from the global cursor
            unsigned long long base = 0;
            if ((int)lane_id() == leader) base = atomicAdd((unsigned long long *)out.cursor, (unsigned long long)cnt);
            base = __shfl(base, leader);
            o = base + (uint64_t)__popcll(m & ((1ULL << lane_id()) - 1ULL));
          }
          if (keep && o < out.cap) {  // beyond the arrays: the host sees the cursor past cap and re-runs with more room
            if (CP) {
              out.keys[o] = cp_unpack_rec(keys32[s] << 6, r, gm);
            } else {
#pragma unroll
              for (int x = 0; x < NL; x++) out.keys[o * NL + x] = tb.keys[x * SM + s];
            }
            out.counts[o] = (uint16_t)count;
            if (DUMP) {
              constexpr int R0 = EW == 6 ? 3 : 2;
#pragma unroll
              for (int x = 0; x < 4; x++) {
                out.exts[o * 8 + x] = (uint16_t)((w[x >> 1] >> (16 * (x & 1))) & 0xFFFFu);
                out.exts[o * 8 + 4 + x] = (uint16_t)((w[R0 + (x >> 1)] >> (16 * (x & 1))) & 0xFFFFu);
              }
            } else {
              out.left[o] = (uint8_t)("ACGT"[l & 3u]);
              out.right[o] = (uint8_t)("ACGT"[rr & 3u]);
            }
          }
          if (keep) {
            acc_kept++;
            acc_sum += count;
          }
        }
        if (live) reset_slot(s);
      }
      if (failed && tid == 0) {
        bb.flag[r] = 2;
        T.fail[buf] = 0;  // (this half is next used two regions on)
      }
      sink ^= touch;  // (first use of the touched word: by now it has arrived)
      __syncthreads();
      KC_STAMP(2)  // table pass + barrier
    }
    r = rn;
    buf ^= 1;
  }
  if (tid == 0) {
    settle();
    if (out.block) {
      out.tails[4 * blockIdx.x] = cur_base;
      out.tails[4 * blockIdx.x + 1] = cur_left;
      out.tails[4 * blockIdx.x + 2] = sp_base;
      out.tails[4 * blockIdx.x + 3] = sp_left;
    }
  }
  if (!DUMP) {
    // (a lane's entries and its survivors are counted in different passes over different slots: the difference is only
    // meaningful summed over the lanes, modulo 2^64)
    unsigned long long e = acc_entries, p = (unsigned long long)acc_entries - (unsigned long long)acc_kept, sm = acc_sum;
    for (int o = 32; o > 0; o >>= 1) {
      e += __shfl_down(e, o);
      p += __shfl_down(p, o);
      sm += __shfl_down(sm, o);
    }
    if (lane_id() == 0) {
      if (e) atomicAdd((unsigned long long *)&cb[CB_ENTRIES], e);
      if (p) atomicAdd((unsigned long long *)&ctrs[CTR_PURGED], p);  // (may be "negative": the sum over the waves is not)
      if (sm) atomicAdd((unsigned long long *)&ctrs[CTR_SUM_COUNTS], sm);
    }
  }
  if (sink == 0x9E3779B9u && R == 0) cb[CB_DUMP] = sink;  // keeps the look-ahead loads alive; never true
#ifdef KC_STAMPS
  if (stamp)
    for (int k = 0; k < 5; k++) atomicAdd((unsigned long long *)&cb[8 + k], tacc[k]);
#endif
#undef KC_STAMP
}

// ---- closing the holes that block-wise output leaves (OutBufs::block) ------------------------------------------------
// plan[0] = entries to move, plan[1] = holes below the final size, plan[2] = source runs; then the hole runs
// (start, entries before it) and the source runs, PLAN_RUNS pairs each.
constexpr uint32_t PLAN_RUNS = 1024;  // at most two holes per workgroup (<= 2 workgroups per CU) and one source run between two holes
constexpr size_t PLAN_WORDS = 4 + 4 * (size_t)(PLAN_RUNS + 1);

// One workgroup.  T = the cursor (every block ever taken), holes = the unused tails; the final size is n = T - sum of
// the holes.  Holes below n are filled, in order, with the entries that sit at or above n.  Leaves n in *cursor and T
// in *reserved.
__global__ __launch_bounds__(WGB) void kc_out_plan_kernel(const uint64_t *tails, uint32_t nwg, uint64_t *cursor, uint64_t *reserved,
                                                        uint64_t *plan) {
  __shared__ uint64_t hs[PLAN_RUNS], hl[PLAN_RUNS];  // holes sorted by position
  __shared__ uint32_t nh;
  const uint32_t tid = threadIdx.x;
  if (tid == 0) nh = 0;
  __syncthreads();
  // rank the non-empty holes by position (positions are distinct: blocks do not overlap)
  uint64_t pos = 0, len = 0;
  if (tid < nwg) {
    pos = tails[2 * tid];
    len = tails[2 * tid + 1];
  }
  if (len) {
    uint32_t rank = 0;
    for (uint32_t j = 0; j < nwg; j++)
      if (tails[2 * j + 1] && tails[2 * j] < pos) rank++;
    hs[rank] = pos;
    hl[rank] = len;
    atomicAdd(&nh, 1u);
  }
  __syncthreads();
  if (tid == 0) {
    const uint64_t T = *cursor;
    uint64_t holes = 0;
    for (uint32_t i = 0; i < nh; i++) holes += hl[i];
    const uint64_t n = T - holes;
    uint64_t *hrun = plan + 4, *srun = plan + 4 + 2 * (PLAN_RUNS + 1);
    uint64_t nhole = 0, nsrc = 0, hcount = 0, scount = 0;
    uint64_t at = n;  // walks [n, T): what is not a hole there is a source
    for (uint32_t i = 0; i < nh; i++) {
      const uint64_t b = hs[i], e = hs[i] + hl[i];
      if (b < n) {  // (part of) a hole to fill
        const uint64_t ee = e < n ? e : n;
        hrun[2 * nhole] = b;
        hrun[2 * nhole + 1] = hcount;
        hcount += ee - b;
        nhole++;
      }
      if (e > n) {  // (part of) a hole in the tail: entries between `at` and it are sources
        const uint64_t bb = b > n ? b : n;
        if (bb > at) {
          srun[2 * nsrc] = at;
          srun[2 * nsrc + 1] = scount;
          scount += bb - at;
          nsrc++;
        }
        at = e;
      }
    }
    if (T > at) {
      srun[2 * nsrc] = at;
      srun[2 * nsrc + 1] = scount;
      scount += T - at;
      nsrc++;
    }
    hrun[2 * nhole + 1] = hcount;  // sentinels for the searches
    srun[2 * nsrc + 1] = scount;
    plan[0] = hcount < scount ? hcount : scount;  // equal by construction
    plan[1] = nhole;
    plan[2] = nsrc;
    *reserved = T;
    *cursor = n;
  }
}

// entry m of the move: from the m-th source position to the m-th hole position
template <int NL>
__global__ void kc_out_move_kernel(const uint64_t *plan, OutBufs out) {
  const uint64_t nmove = plan[0];
  const uint32_t nhole = (uint32_t)plan[1], nsrc = (uint32_t)plan[2];
  const uint64_t *hrun = plan + 4, *srun = plan + 4 + 2 * (PLAN_RUNS + 1);
  auto locate = [](const uint64_t *run, uint32_t nrun, uint64_t m) -> uint64_t {  // last run whose prefix is <= m
    uint32_t lo = 0, hi = nrun;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (run[2 * mid + 1] <= m) lo = mid; else hi = mid;
    }
    return run[2 * lo] + (m - run[2 * lo + 1]);
  };
  for (uint64_t m = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; m < nmove; m += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t d = locate(hrun, nhole, m), sidx = locate(srun, nsrc, m);
    if (d >= out.cap || sidx >= out.cap) continue;  // the arrays were too small: the host runs the pass again
#pragma unroll
    for (int w = 0; w < NL; w++) out.keys[d * NL + w] = out.keys[sidx * NL + w];
    out.counts[d] = out.counts[sidx];
    out.left[d] = out.left[sidx];
    out.right[d] = out.right[sidx];
  }
}

// records of flagged regions go to the global table (kc_kernels.hpp): one workgroup per flagged region at a time
template <int NL, bool CP>
__global__ __launch_bounds__(TPB) void kc_flagged_to_table_kernel(Geom gm, BucketBufs bb, Table t, uint64_t *ctrs) {
  const size_t R = (size_t)gm.P1 * gm.P2;
  for (size_t r = blockIdx.x; r < R; r += gridDim.x) {
    if (!bb.flag[r]) continue;
    const uint32_t n = bb.cnt2[r];
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      uint64_t rec[NL];
      if (CP) {
        rec[0] = cp_unpack_rec(*l2_record32(gm, bb, r, i), r, gm);
      } else {
        const uint64_t *src = l2_record<NL>(gm, bb, r, i);
        for (int w = 0; w < NL; w++) rec[w] = src[w];
      }
      table_insert<NL>(t, rec, ctrs);
    }
  }
}

// every buffered level-1 record goes to the global table (the context ran out of buffer room and
// switches to the table path for good)
template <int NL, bool CP>
__global__ __launch_bounds__(TPB) void kc_l1_to_table_kernel(Geom gm, BucketBufs bb, Table t, uint64_t *ctrs) {
  const size_t nseg = (size_t)gm.G * gm.P1;
  for (size_t sgi = blockIdx.x; sgi < nseg; sgi += gridDim.x) {
    const uint32_t n = bb.cnt1[sgi];
    const uint32_t g = (uint32_t)(sgi / gm.P1), b = (uint32_t)(sgi % gm.P1);
    for (uint32_t i = threadIdx.x; i < n; i += TPB) {
      uint64_t rec[NL];
      if (NL == 1 && CP && gm.rec6) {
        rec[0] = l1_record6(gm, bb, g, b, i);
      } else {
        const uint64_t *src = l1_record<NL>(gm, bb, g, b, i);
        for (int w = 0; w < NL; w++) rec[w] = src[w];
      }
      if (CP) rec[0] = cp_unmix_rec(rec[0], gm);
      table_insert<NL>(t, rec, ctrs);
    }
  }
}

__global__ void kc_sum_flagged_kernel(Geom gm, BucketBufs bb, uint64_t *cb) {
  const size_t R = (size_t)gm.P1 * gm.P2;
  unsigned long long acc = 0;
  for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < R; r += (size_t)gridDim.x * blockDim.x)
    if (bb.flag[r]) acc += bb.cnt2[r];
  for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o);
  if (lane_id() == 0 && acc) atomicAdd((unsigned long long *)&cb[CB_FLAGGED_RECS], acc);
}

}  // namespace kc
