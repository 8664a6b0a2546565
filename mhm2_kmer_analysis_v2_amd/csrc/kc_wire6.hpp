// kc_wire6.hpp -- the records flow with the library's own wire record (KC_FLAG_WIRE_UNITS).
//
// The records flow (kc_extract_partition on the sender, kc_insert_records on the receiver; the reference's
// ThreeTierAggrStore<Supermer>::update / flush_updates + insert_supermer_block, src/kcount/kmer_dht.cpp:143-151,247-258,
// src/kcount/kcount-gpu/gpu_hash_table.cpp:655-695) lets every shard use the whole geometry -- a full-size shard at any
// number of shards -- at the price of two more passes over every record.  With plain k-mer records those two passes cost
// what two level 1s cost (27 + 29 ms per 50 M reads): the sender hashes every k-mer for its owner and stores 8-byte
// records from the registers, the receiver mixes every k-mer again.  For the short-form compact records of k = 21
// (Geom::rec6) this file gives both passes what level 1 got:
//
//   sender    kc_bin16_kernel      level 1's super-tile staging and cut (cp_run_fixed: sixteen k-mers per thread, already
//                                  MIXED), the owner shard taken from eight bits of the mix (wire6_owner: bits that only
//                                  the probe stride of the region tables uses -- buckets, regions and start slots keep
//                                  all of theirs, so every shard uses the whole geometry), the PIECE from the top bits of
//                                  the level-1 bucket (below), ranked with LDS adds over (owner, piece) x lane-cell
//                                  counters (a handful of cells alone would serialise the adds), staged sorted by (owner,
//                                  piece) in six bytes, copied out as 12-byte pairs in runs of hundreds of records
//   wire      SIX bytes a record   the 32 bits of the mix below the bucket + bucket | extension codes << 10: what level 1
//                                  stages.  The unit of the exchange is FOUR records = three words; a run a sender appends
//                                  is padded to an even length and a piece is closed to whole units with marker slots
//                                  (bk = 0xFFFF; 0.1 % of the slots), so that pieces of whole words can be shipped and
//                                  laid end to end
//   receiver  kc_l1_wire6_kernel   level 1 from received records: nothing to unpack, mix or hash -- a 12-byte load per
//                                  pair, the bucket is ten bits of the record; staging and copy-out are level 1's own.
//                                  It reads up to sixteen pieces per launch, a round of a workgroup out of one piece
//
// A record is written once more than in the unsharded pass (the wire segment) and read once more.
#pragma once
#include "kc_bucketed.hpp"

namespace kc {

constexpr uint32_t WIRE6_UNIT_RECORDS = 4, WIRE6_UNIT_WORDS = 3;
constexpr uint32_t WIRE6_MARK = 0xFFFFu;  // bucket | extension codes of a slot that holds no record (kc_bin16_kernel keeps records off this value)
constexpr uint32_t WIRE6_MAX_SHARDS = 64;

// owner shard of a mixed k-mer: bits 13..20 of the mix, scaled to [0, n)
__host__ __device__ __forceinline__ uint32_t wire6_owner(uint32_t lo, uint32_t n) { return (((lo >> 13) & 255u) * n) >> 8; }

constexpr size_t bin16_lds_bytes() { return l1x16_lds_bytes(); }

// Pieces per destination: the sender keeps a destination's records apart by the top bits of their level-1 BUCKET, one piece
// per value.  The receiver's level 1 is bound by the store requests of its 1024-way split -- sixteen records per bucket and
// round are runs of 96 bytes that touch 1.75 lines each -- and it reads what it receives front to back: a round that
// comes out of ONE piece of eight meets 128 buckets instead of 1024, its runs are 128 records long, and the requests
// per record drop to little more than the lines the records fill: 26 -> 18 ms per 50 M reads (profiles/r04_ab_wire6_pieces.txt;
// two pieces 20.9, four 18.5, eight 17.8, sixteen 17.9).  The sender sorts by (owner, piece) instead of by owner at the
// price of a few instructions per record (21.1 -> 22.1 ms).  128 (owner, piece) cells at most, so that the sender's own
// runs stay long.
constexpr uint32_t WIRE6_MAX_LG_PIECES = 3;  // (sixteen measured no better than eight)
__host__ __device__ __forceinline__ uint32_t wire6_ceil_log2(uint32_t n) {
  uint32_t l = 0;
  while ((1u << l) < n) l++;
  return l;
}
__host__ __device__ __forceinline__ uint32_t wire6_lg_pieces(uint32_t nshards) {
  const uint32_t lg = wire6_ceil_log2(nshards);
  return lg >= 7u ? 0u : min(WIRE6_MAX_LG_PIECES, 7u - lg);
}

// cur[d << lgQ | q]: slots appended to piece q of destination d so far (even); the piece starts at
// records + (d << lgQ | q) * seg_units * 3 words and holds seg_units * 4 slots
template <int FMT, int KK>
__global__ __launch_bounds__(WGB) void kc_bin16_kernel(ExtractArgs a, Geom gm, uint64_t nsuper, uint64_t *ctrs, uint64_t *cur, uint32_t lgQ) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1LDS &L = *reinterpret_cast<L1LDS *>(smem);
  uint32_t *slo = reinterpret_cast<uint32_t *>(smem + ((sizeof(L1LDS) + 15) & ~size_t(15)));
  uint16_t *sbk = reinterpret_cast<uint16_t *>(slo + ST16_SLOTS);
  const int tid = threadIdx.x;
  const uint32_t P = a.rank_n;  // <= WIRE6_MAX_SHARDS
  const uint32_t qsh = gm.la - lgQ, V = P << lgQ;  // V (owner, piece) cells, <= 128 (lgQ = wire6_lg_pieces(P))
  // lane cells per (owner, piece): LDS adds of a wave to a handful of words would be serialised
  const uint32_t lgV = wire6_ceil_log2(V), lgS = lgV >= 7u ? 0u : 7u - lgV, SUBm = (1u << lgS) - 1u, NV = V << lgS;
  for (uint32_t i = (uint32_t)tid; i < PMAX + 64; i += WGB) {
    L.sp.hist[0][i] = 0;
    L.sp.hist[1][i] = 0;
  }
  __syncthreads();
  int buf = 0;
  TileRaw<TileSuper> raw;  // the next super-tile's bytes, on their way while this one is split (kc_l1_reads16_kernel)
  auto first_of = [&](uint64_t st) -> uint64_t { return (FMT != FMT_SEQBLOCK && st < nsuper) ? a.tile_first[st] : 0; };
  tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, tid, first_of(blockIdx.x), blockIdx.x < nsuper);
  uint64_t next_first = first_of((uint64_t)blockIdx.x + gridDim.x);
  constexpr int RUNS = SUPER_SPAN / R16;
  {
    const int ft = fresh_tid();
    tile_encode<FMT, TileSuper>(L.tile, raw, a, a.pos0 + (int64_t)blockIdx.x * SUPER_SPAN, ctrs, ft, blockIdx.x < nsuper);
    tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)((uint64_t)blockIdx.x + gridDim.x) * SUPER_SPAN, ft, next_first,
                                  (uint64_t)blockIdx.x + gridDim.x < nsuper);
    next_first = first_of((uint64_t)blockIdx.x + 2 * (uint64_t)gridDim.x);
  }
  const uint64_t piece_slots = a.seg_capacity * WIRE6_UNIT_RECORDS;
  uint8_t *const out0 = reinterpret_cast<uint8_t *>(a.records);
  const uint32_t lcell = lane_id() & SUBm;
  auto cell_of = [&](uint32_t lo32, uint32_t bk16) -> uint32_t { return (wire6_owner(lo32, P) << lgQ) | ((bk16 & (PMAX - 1)) >> qsh); };
  for (uint64_t st = blockIdx.x; st < nsuper; st += gridDim.x) {
    uint32_t lo[R16], br[R16];
    const bool active = tid < RUNS;
    const int lp0 = PRE + (active ? tid : 0) * R16;
    {
      uint32_t l8[8], b8[8];
      cp_run_fixed<KK, 8, false>(L.tile, lp0, active, gm, a, l8, b8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        lo[j] = l8[j];
        br[j] = b8[j];
      }
      cp_run_fixed<KK, 8, false>(L.tile, lp0 + 8, active, gm, a, l8, b8);
#pragma unroll
      for (int j = 0; j < 8; j++) {
        lo[8 + j] = l8[j];
        br[8 + j] = b8[j];
      }
    }
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const bool valid = br[j] != ~0u;
      const uint32_t vb = (cell_of(lo[j], br[j]) << lgS) | lcell;
      const uint32_t rank = hist_rank(L.sp, buf, valid ? vb : 0u, valid);
      // (both extensions missing with all their ignored low bits set, in the last bucket, would read as a marker: a missing
      // extension is any code >= 4, so the left one's lowest bit is dropped there)
      const uint32_t b16 = br[j] & 0xFFFFu;
      br[j] = valid ? ((b16 == WIRE6_MARK ? (WIRE6_MARK & ~(1u << 10)) : b16) | (rank << 16)) : ~0u;
    }
    lds_barrier();
    const int ft = fresh_tid();
    const uint64_t nst = st + gridDim.x;
    tile_encode_clear<TileSuper>(L.tile, raw, ft);
    // scan + reserve: thread t < NV has lane cell t & SUBm of (owner, piece) cell t >> lgS; a cell's lane cells lie side by
    // side in the staging, its run starts on an even position and is padded to an even length (one marker slot), so that
    // the copy-out takes pairs and a pair never straddles two cells
    {
      const uint32_t v = ((uint32_t)ft < NV) ? L.sp.hist[buf][ft] : 0u;
      uint32_t run_total = v;  // over the cell's lane cells (consecutive lanes of one wave)
      for (uint32_t o = 1; o <= SUBm; o <<= 1) run_total += (uint32_t)__shfl_xor((int)run_total, (int)o);
      const uint32_t pad = run_total & 1u;
      const bool last_lane_cell = ((uint32_t)ft & SUBm) == SUBm;
      const uint32_t excl = block_excl_scan(v + (((uint32_t)ft < NV && last_lane_cell) ? pad : 0u), L.sp.scan);
      const uint32_t run_start = (uint32_t)__shfl((int)excl, (int)(lane_id() & ~SUBm));
      if ((uint32_t)ft < NV) {
        const uint32_t dq = (uint32_t)ft >> lgS;
        if (((uint32_t)ft & SUBm) == 0) {
          const uint32_t want = run_total + pad;
          uint64_t base = 0;
          if (want) base = atomicAdd((unsigned long long *)&cur[dq], (unsigned long long)want);
          const bool ok = base + want + 2 <= piece_slots;  // (two slots stay free for kc_wire6_seal_kernel)
          if (!ok) ctrs[CTR_OVERFLOW] = 1;
          const uint64_t delta = (uint64_t)dq * piece_slots + base - run_start;  // slot in the whole buffer = delta + staging position
          uint4 d;
          d.x = (uint32_t)delta;
          d.y = (uint32_t)(delta >> 32);
          d.z = ok ? 1u : 0u;
          d.w = 0;
          L.sp.dst[dq] = d;
        }
        if (last_lane_cell && pad) {
          slo[excl + v] = 0;
          sbk[excl + v] = (uint16_t)WIRE6_MARK;
        }
        L.sp.hist[buf][ft] = excl;   // where the lane cell's records start in the staging
        L.sp.hist[buf ^ 1][ft] = 0;  // next round's histogram
      }
    }
    lds_barrier();
    const uint32_t total = L.sp.scan.total;
#pragma unroll
    for (int j0 = 0; j0 < R16; j0 += 8) {
      uint32_t pos[8];
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t bj = br[j0 + j];
        pos[j] = L.sp.hist[buf][bj != ~0u ? ((cell_of(lo[j0 + j], bj) << lgS) | lcell) : 0u];
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t bj = br[j0 + j];
        const uint32_t p = bj != ~0u ? pos[j] + (bj >> 16) : ST16_MAIN + lane_id();
        slo[p] = lo[j0 + j];
        sbk[p] = (uint16_t)bj;
      }
    }
    tile_encode_fill<FMT, TileSuper>(L.tile, raw, a, a.pos0 + (int64_t)nst * SUPER_SPAN, ctrs, ft, nst < nsuper);
    tile_prefetch<FMT, TileSuper>(raw, a, a.pos0 + (int64_t)(nst + gridDim.x) * SUPER_SPAN, ft, next_first, nst + gridDim.x < nsuper);
    next_first = first_of(nst + 2 * (uint64_t)gridDim.x);
    lds_barrier();
    // copy-out: the staging two slots at a time; a pair's first slot is always a record (runs start on even positions, a
    // marker only ever closes a run) and names the pair's cell
    {
      const int ct = fresh_tid();
      for (uint32_t i = 2u * (uint32_t)ct; i < total; i += 2u * WGB) {
        const uint64_t lo2 = *reinterpret_cast<const uint64_t *>(slo + i);
        const uint32_t bk2 = *reinterpret_cast<const uint32_t *>(sbk + i);
        const uint4 d = L.sp.dst[cell_of((uint32_t)lo2, bk2)];
        if (d.z) {
          const uint64_t at = (((uint64_t)d.y << 32) | d.x) + i;
          const uint32_t l1 = (uint32_t)(lo2 >> 32);
          Rec6Pair r;
          r.w0 = (uint32_t)lo2;
          r.w1 = (bk2 & 0xFFFFu) | (l1 << 16);
          r.w2 = (l1 >> 16) | (bk2 & 0xFFFF0000u);
          *reinterpret_cast<Rec6Pair *>(out0 + at * 6) = r;
        }
      }
    }
    buf ^= 1;
  }
}

// behind the sender: every piece is closed to whole units (a piece's length is even: at most one pair of marker slots)
__global__ void kc_wire6_seal_kernel(uint64_t *cur, uint32_t npieces, uint8_t *out0, uint64_t piece_slots) {
  const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= npieces) return;
  const uint64_t c = cur[j];
  if ((c & 3u) == 0 || c + 2 > piece_slots) return;  // (an overflowing piece has been reported already)
  Rec6Pair r;
  r.w0 = 0;
  r.w1 = WIRE6_MARK;
  r.w2 = WIRE6_MARK << 16;
  *reinterpret_cast<Rec6Pair *>(out0 + ((uint64_t)j * piece_slots + c) * 6) = r;
  cur[j] = c + 2;
}

// ---- receiver: level 1 from wire records ---------------------------------------------------------------------------------
// What one launch reads: up to sixteen pieces, piece j = `pairs[j]` pairs of six-byte slots (marker slots among them) at
// base + j * stride, 4-byte aligned.  A round of a workgroup comes out of ONE piece (rpre[j]: rounds before piece j; a
// piece's last round may be short), so that a sender's pieces -- which lie apart in its buffer -- cost one launch, not one
// each, and nothing has to be looked up per record.
constexpr int WIRE6_SRC_PIECES = 16;
struct Wire6Src {
  const uint8_t *base;
  uint64_t stride;  // bytes
  uint32_t n;       // pieces, every one with at least one pair
  uint32_t rpre[WIRE6_SRC_PIECES + 1];
  uint32_t pairs[WIRE6_SRC_PIECES];
};

__global__ __launch_bounds__(WGB) void kc_l1_wire6_kernel(Wire6Src src, Geom gm, BucketBufs bb, uint32_t rot, uint64_t *ctrs, uint64_t *cb) {
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
  constexpr int NPAIR = R16 / 2;
  constexpr uint32_t PPR = (uint32_t)WGB * NPAIR;  // pairs per round
  const uint32_t nrounds = src.rpre[src.n];
  uint32_t n_ins = 0;
  int buf = 0;
  Load12 nxt[NPAIR];  // the next round's pairs, on their way while this round is split
  uint32_t lim_nxt = 0, lim = 0;  // pairs the round holds (the same for every thread)
  auto load_round = [&](uint64_t rd) {  // no branches around the loads (a lane past the end re-reads the piece's last pair)
    uint32_t j = 0;
    const uint32_t r = rd < nrounds ? (uint32_t)rd : nrounds - 1;
    for (uint32_t q = 1; q < src.n; q++) j += r >= src.rpre[q] ? 1u : 0u;
    const uint32_t first = (r - src.rpre[j]) * PPR, np = src.pairs[j];
    lim_nxt = rd < nrounds ? min(PPR, np - first) : 0u;
    const uint8_t *pb = src.base + (uint64_t)j * src.stride;
#pragma unroll
    for (int u = 0; u < NPAIR; u++) {
      uint32_t p = first + (uint32_t)u * WGB + (uint32_t)tid;
      p = p < np ? p : np - 1;
      nxt[u] = *reinterpret_cast<const Load12 *>(pb + (uint64_t)p * 12);
    }
  };
  uint32_t lo[R16], br[R16];
  auto take_over = [&]() {
    lim = lim_nxt;
#pragma unroll
    for (int u = 0; u < NPAIR; u++) {
      const Load12 w = nxt[u];
      lo[2 * u] = w.a;
      br[2 * u] = w.b & 0xFFFFu;
      lo[2 * u + 1] = (w.b >> 16) | (w.c << 16);
      br[2 * u + 1] = w.c >> 16;
    }
  };
  load_round(blockIdx.x);
  take_over();
  load_round((uint64_t)blockIdx.x + gridDim.x);
  for (uint64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const bool valid = (uint32_t)(j >> 1) * WGB + (uint32_t)tid < lim && br[j] != WIRE6_MARK;
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

// ---- a context that has left the bucketed path: wire records back to k-mer records for the global table ----
// out: room for nslots records; *n_out counts the records written (markers dropped)
__global__ __launch_bounds__(TPB) void kc_wire6_expand_kernel(const uint8_t *recs, uint64_t nslots, Geom gm, uint64_t *out, unsigned long long *n_out) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ((nslots + 63) & ~63ULL); i += (uint64_t)gridDim.x * blockDim.x) {
    bool valid = i < nslots;
    uint64_t rec = 0;
    if (valid) {
      const Rec6 r = *reinterpret_cast<const Rec6 *>(recs + i * 6);
      valid = r.bk != WIRE6_MARK;
      const uint64_t mixed = ((uint64_t)((uint32_t)r.bk & (PMAX - 1)) << (64u - gm.la)) | ((uint64_t)r.lo << (64u - gm.k2)) | (uint64_t)(((uint32_t)r.bk >> 10) & 63u);
      rec = cp_unmix_rec(mixed, gm);
    }
    const uint64_t m = __ballot(valid);
    uint64_t base = 0;
    if (lane_id() == 0 && m) base = atomicAdd(n_out, (unsigned long long)__popcll(m));
    base = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(base >> 32), 0) << 32) | (uint32_t)__shfl((int)(uint32_t)base, 0);
    if (valid) out[base + (uint64_t)__popcll(m & ((1ULL << lane_id()) - 1ULL))] = rec;
  }
}

}  // namespace kc
