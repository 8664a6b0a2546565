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
//                                  all of theirs, so every shard uses the whole geometry), ranked with LDS adds over
//                                  owner x sub-counter cells (a handful of owners alone would serialise the adds),
//                                  staged sorted by owner in six bytes, copied out as 12-byte pairs in long runs
//   wire      SIX bytes a record   the 32 bits of the mix below the bucket + bucket | extension codes << 10: what level 1
//                                  stages.  The unit of the exchange is FOUR records = three words, every run a sender
//                                  appends to a segment is padded to whole units with marker slots (bk = 0xFFFF), so that
//                                  segments of whole words can be shipped and laid end to end
//   receiver  kc_l1_wire6_kernel   level 1 from received records: nothing to unpack, mix or hash -- a 12-byte load per
//                                  pair, the bucket is ten bits of the record; staging and copy-out are level 1's own
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
// the smallest value of those eight bits that owner o has, in place (a marker slot must land in its owner's run)
__host__ __device__ __forceinline__ uint32_t wire6_marker_lo(uint32_t o, uint32_t n) { return ((o * 256u + n - 1u) / n) << 13; }

constexpr size_t bin16_lds_bytes() { return l1x16_lds_bytes(); }

// cursors[d]: slots appended to segment d so far (a multiple of four); segment d starts at records + d * seg_units * 3 words
template <int FMT, int KK>
__global__ __launch_bounds__(WGB) void kc_bin16_kernel(ExtractArgs a, Geom gm, uint64_t nsuper, uint64_t *ctrs) {
  extern __shared__ __align__(16) uint8_t smem[];
  L1LDS &L = *reinterpret_cast<L1LDS *>(smem);
  uint32_t *slo = reinterpret_cast<uint32_t *>(smem + ((sizeof(L1LDS) + 15) & ~size_t(15)));
  uint16_t *sbk = reinterpret_cast<uint16_t *>(slo + ST16_SLOTS);
  const int tid = threadIdx.x;
  const uint32_t P = a.rank_n;  // <= WIRE6_MAX_SHARDS
  // cells per owner: as many as keep owner x cell within the histogram and a cell's lanes within a wave
  uint32_t lgP = 0;
  while ((1u << lgP) < P) lgP++;
  const uint32_t lgS = min(6u, 10u - lgP), SUBm = (1u << lgS) - 1u, V = P << lgS;
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
  const uint64_t seg_slots = a.seg_capacity * WIRE6_UNIT_RECORDS;
  uint8_t *const out0 = reinterpret_cast<uint8_t *>(a.records);
  const uint32_t cell = lane_id() & SUBm;
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
      const uint32_t vb = (wire6_owner(lo[j], P) << lgS) | cell;
      const uint32_t rank = hist_rank(L.sp, buf, vb, valid);
      // (both extensions missing with all their ignored low bits set, in the last bucket, would read as a marker: a missing
      // extension is any code >= 4, so the left one's lowest bit is dropped there)
      const uint32_t b16 = br[j] & 0xFFFFu;
      br[j] = valid ? ((b16 == WIRE6_MARK ? (WIRE6_MARK & ~(1u << 10)) : b16) | (rank << 16)) : ~0u;
    }
    lds_barrier();
    const int ft = fresh_tid();
    const uint64_t nst = st + gridDim.x;
    tile_encode_clear<TileSuper>(L.tile, raw, ft);
    // scan + reserve: thread t < V has cell t of owner t >> lgS; an owner's cells lie side by side in the staging, its run
    // is padded to whole units (marker slots behind the last cell's records)
    {
      const uint32_t v = ((uint32_t)ft < V) ? L.sp.hist[buf][ft] : 0u;
      uint32_t own_total = v;  // over the owner's cells (consecutive lanes of one wave)
      for (uint32_t o = 1; o <= SUBm; o <<= 1) own_total += (uint32_t)__shfl_xor((int)own_total, (int)o);
      const uint32_t pad = (0u - own_total) & (WIRE6_UNIT_RECORDS - 1u);
      const bool last_cell = ((uint32_t)ft & SUBm) == SUBm;
      const uint32_t excl = block_excl_scan(v + (((uint32_t)ft < V && last_cell) ? pad : 0u), L.sp.scan);
      const uint32_t own_start = (uint32_t)__shfl((int)excl, (int)(lane_id() & ~SUBm));
      if ((uint32_t)ft < V) {
        const uint32_t o = (uint32_t)ft >> lgS;
        if (((uint32_t)ft & SUBm) == 0) {
          const uint32_t want = own_total + pad;
          uint64_t base = 0;
          if (want) base = atomicAdd((unsigned long long *)&ctrs[CTR_BIN0 + o], (unsigned long long)want);
          const bool ok = base + want <= seg_slots;
          if (!ok) ctrs[CTR_OVERFLOW] = 1;
          const uint64_t delta = (uint64_t)o * seg_slots + base - own_start;  // slot in the whole buffer = delta + staging position
          uint4 d;
          d.x = (uint32_t)delta;
          d.y = (uint32_t)(delta >> 32);
          d.z = ok ? 1u : 0u;
          d.w = 0;
          L.sp.dst[o] = d;
        }
        if (last_cell) {
          const uint32_t mlo = wire6_marker_lo(o, P);
          for (uint32_t i = 0; i < pad; i++) {
            slo[excl + v + i] = mlo;
            sbk[excl + v + i] = (uint16_t)WIRE6_MARK;
          }
        }
        L.sp.hist[buf][ft] = excl;      // where the cell's records start in the staging
        L.sp.hist[buf ^ 1][ft] = 0;     // next round's histogram
      }
    }
    lds_barrier();
    const uint32_t total = L.sp.scan.total;
#pragma unroll
    for (int j0 = 0; j0 < R16; j0 += 8) {
      uint32_t pos[8];
#pragma unroll
      for (int j = 0; j < 8; j++) pos[j] = L.sp.hist[buf][(wire6_owner(lo[j0 + j], P) << lgS) | cell];
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
    // copy-out: the staging two slots at a time; an owner's run is a whole number of units, so a pair never straddles two
    // owners and always holds two slots (records or markers)
    {
      const int ct = fresh_tid();
      for (uint32_t i = 2u * (uint32_t)ct; i < total; i += 2u * WGB) {
        const uint64_t lo2 = *reinterpret_cast<const uint64_t *>(slo + i);
        const uint32_t bk2 = *reinterpret_cast<const uint32_t *>(sbk + i);
        const uint4 d = L.sp.dst[wire6_owner((uint32_t)lo2, P)];
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

// ---- receiver: level 1 from wire records ---------------------------------------------------------------------------------
// recs: nslots slots of six bytes (a multiple of four; marker slots among them), 4-byte aligned
__global__ __launch_bounds__(WGB) void kc_l1_wire6_kernel(const uint8_t *recs, uint64_t nslots, Geom gm, BucketBufs bb, uint32_t rot, uint64_t *ctrs,
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
  constexpr int NPAIR = R16 / 2;
  const uint64_t npairs = nslots / 2, pairs_per_round = (uint64_t)WGB * NPAIR;
  const uint64_t nrounds = (npairs + pairs_per_round - 1) / pairs_per_round;
  uint32_t n_ins = 0;
  int buf = 0;
  Load12 nxt[NPAIR];  // the next round's pairs, on their way while this round is split
  auto load_round = [&](uint64_t rd) {  // npairs > 0; no branches around the loads (a lane past the end re-reads the last pair)
#pragma unroll
    for (int u = 0; u < NPAIR; u++) {
      uint64_t p = rd * pairs_per_round + (uint64_t)u * WGB + tid;
      p = p < npairs ? p : npairs - 1;
      nxt[u] = *reinterpret_cast<const Load12 *>(recs + p * 12);
    }
  };
  uint32_t lo[R16], br[R16];
  auto take_over = [&]() {
#pragma unroll
    for (int u = 0; u < NPAIR; u++) {
      const Load12 w = nxt[u];
      lo[2 * u] = w.a;
      br[2 * u] = w.b & 0xFFFFu;
      lo[2 * u + 1] = (w.b >> 16) | (w.c << 16);
      br[2 * u + 1] = w.c >> 16;
    }
  };
  if (npairs) load_round(blockIdx.x);
  take_over();
  if (npairs) load_round((uint64_t)blockIdx.x + gridDim.x);
  for (uint64_t rd = blockIdx.x; rd < nrounds; rd += gridDim.x) {
#pragma unroll
    for (int j = 0; j < R16; j++) {
      const bool valid = rd * pairs_per_round + (uint64_t)(j >> 1) * WGB + tid < npairs && br[j] != WIRE6_MARK;
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
