// kc_shard.hpp -- the single-pass shard flow: ownership by level-1 bucket.
//
// The reference sends every supermer to the rank that owns its k-mers (ThreeTierAggrStore<Supermer>::update,
// src/kcount/kmer_dht.cpp:143-151,247-258; owner = get_kmer_target_rank, kmer_dht.cpp:192-196) and the owner inserts
// them (HashTableGPUDriver::insert_supermer_block, gpu_hash_table.cpp:655-695).  Here the owner of a k-mer is the owner
// of its level-1 bucket (shard_of_bucket: each shard owns a contiguous range of the P1 buckets), so the sender's
// ordinary level-1 pass (kc_l1_reads_kernel over ALL its k-mers, no ownership test per k-mer) has already sorted its
// records by destination when it ends:
//
//   sender    kc_l1_reads_kernel           reads -> (writer, bucket) chains, exactly as without shards
//             kc_shard_plan_kernel         per foreign bucket: how many records, where in the destination's segment
//             kc_shard_pack_kernel         the chains of the foreign buckets, copied dense into one segment per
//                                          destination (header + per-bucket counts + records, bucket after bucket);
//                                          the buckets this shard owns never move
//   wire      one segment per destination  (the caller: csrc/kc_exchange.hpp over RCCL, dist.py over torch.distributed)
//   receiver  kc_shard_index_kernel        a received segment becomes one more *flat source* of its buckets:
//                                          per bucket a count and the address of its first record -- no copy
//             kc_l2_split_kernel<.., FL>   level 2 walks, per bucket, its own G chains and then the flat sources
//
// so a record is written once by level 1, copied once into the wire segment if it is foreign, and read once by level 2
// where it lands: the two extra passes of the records flow (kc_bin_reads_kernel on the sender, kc_l1_records_kernel on
// the receiver) are gone, and a shard's own share costs nothing extra at all.
//
// Records that found no room at level 1 (kc_bucketed.hpp's first overflow list) travel as "loose" k-mer records behind
// the buckets of their destination's segment and go to the receiver's global table with their region flagged, like the
// list's own records do at home (kc_ovf1_drain_kernel).
#pragma once
#include "kc_bucketed.hpp"

namespace kc {

// Two forms of a segment's record area (word 3 of the header says which, and how many words the area has):
//   SHARD_WIRE_WORDS    every record as its NL words, bucket after bucket;
//   SHARD_WIRE_COMPACT  compact records in the short form (Geom::cp with 2k - la <= 32: the metric's k=21): FIVE BYTES a
//                       record -- the 32 bits of the mix below the bucket, then the six extension bits in a byte; the
//                       bucket, which the segment is sorted by, is implied.  xGMI is what bounds an 8-shard stage at the
//                       kernels' rate (7/8 of 6.4 G records per 50 M reads and shard: 44.8 GB as words = 83 ms of a
//                       76.8 GB/s link's time, above the kernels' 72; 28 GB = 52 ms as five bytes).  A bucket's block is
//                       5 * count bytes, padded to whole words with at least three spare bytes behind the last record:
//                       level 2 takes a record with ONE unaligned 8-byte load (kc_l2_split_kernel<.., CR, FL>), like a
//                       record of a chain, and never reads outside the block.
constexpr uint32_t SHARD_WIRE_WORDS = 0, SHARD_WIRE_COMPACT = 1;
constexpr uint32_t SHARD_HDR = 4;       // words of a segment before its per-bucket counts
constexpr uint32_t SHARD_MAX = 64;      // shards of one exchange (the ABI's rank_n limit)

// owner of level-1 bucket b among n shards, and the first bucket of shard d: contiguous ranges
__host__ __device__ inline uint32_t shard_of_bucket(uint32_t b, uint32_t P1, uint32_t n) { return (uint32_t)(((uint64_t)b * n) / P1); }
__host__ __device__ inline uint32_t shard_first_bucket(uint32_t d, uint32_t P1, uint32_t n) { return (uint32_t)(((uint64_t)d * P1 + n - 1) / n); }
// words of a segment's header for nb buckets: signature, nb | loose << 32, records, words of the record area | form << 56,
// then nb u32 counts
__host__ __device__ inline uint64_t shard_header_words(uint32_t nb) { return SHARD_HDR + ((uint64_t)nb + 1) / 2; }
// words of a bucket's block of cnt records
__host__ __device__ inline uint64_t shard_bucket_words(uint64_t cnt, uint32_t nl, uint32_t wire) {
  return wire == SHARD_WIRE_COMPACT ? (5 * cnt + 3 + 7) >> 3 : cnt * nl;
}
__host__ __device__ inline uint32_t shard_wire_of(const Geom &gm) { return (gm.cp && gm.k2 - gm.la <= 32u) ? SHARD_WIRE_COMPACT : SHARD_WIRE_WORDS; }

// Per destination: where each of its buckets starts in the segment's record area, how many records it gets, the
// header.  One workgroup.  off[b]: WORD offset of bucket b's block inside its destination's record area; totals[d]:
// records for destination d (for d == me: the records that stay); wtotals[d]: words of its record area; flags[d] != 0:
// segment d is too small (nothing is written to it and the chains are kept, so nothing is lost).
__global__ __launch_bounds__(WGB) void kc_shard_plan_kernel(Geom gm, BucketBufs bb, uint32_t me, uint32_t n, uint64_t *segs, uint64_t seg_words,
                                                            uint64_t sig, uint32_t nl, uint64_t *off, uint64_t *totals, uint64_t *flags,
                                                            uint64_t *wtotals) {
  const uint32_t wire = shard_wire_of(gm);
  __shared__ uint32_t s_n[PMAX];
  __shared__ uint32_t s_bad[SHARD_MAX];
  const uint32_t b = threadIdx.x;
  uint32_t nb_rec = 0;
  if (b < gm.P1) {
    uint64_t s = 0;
    for (uint32_t g = 0; g < gm.G; g++) s += bb.cnt1[(size_t)g * gm.P1 + b];
    nb_rec = (uint32_t)s;  // a bucket holds < 2^31 records (bk_init)
    s_n[b] = nb_rec;
  }
  __syncthreads();
  if (b < n) {
    const uint32_t lo = shard_first_bucket(b, gm.P1, n), hi = shard_first_bucket(b + 1, gm.P1, n);
    uint64_t run = 0, wrun = 0;
    for (uint32_t i = lo; i < hi; i++) {
      off[i] = wrun;
      run += s_n[i];
      wrun += shard_bucket_words(s_n[i], nl, wire);
    }
    totals[b] = run;
    wtotals[b] = wrun;
    const uint64_t H = shard_header_words(hi - lo);
    const bool bad = b != me && H + wrun > seg_words;
    flags[b] = bad ? 1 : 0;
    s_bad[b] = bad ? 1u : 0u;
    if (b != me && !bad) {
      uint64_t *seg = segs + (size_t)b * seg_words;
      seg[0] = sig;
      seg[1] = (uint64_t)(hi - lo);  // loose records: filled in by the host when there are any
      seg[2] = run;
      seg[3] = wrun | ((uint64_t)wire << 56);
      if ((hi - lo) & 1u) reinterpret_cast<uint32_t *>(seg + SHARD_HDR)[hi - lo] = 0;  // padding of the counts
    }
  }
  __syncthreads();
  if (b < gm.P1) {
    const uint32_t d = shard_of_bucket(b, gm.P1, n);
    if (d != me && !s_bad[d]) reinterpret_cast<uint32_t *>(segs + (size_t)d * seg_words + SHARD_HDR)[b - shard_first_bucket(d, gm.P1, n)] = nb_rec;
  }
}

// one record of the compact wire form
struct __attribute__((packed)) WireRec5 {
  uint32_t lo;
  uint8_t ext;
};
static_assert(sizeof(WireRec5) == 5, "five bytes on the wire");

// The (writer, bucket) chains of the foreign buckets, copied dense: bucket after bucket, inside a bucket writer after
// writer.  Q workgroups share a bucket (writers q, q + Q, ...).  A plain streaming copy: 8 NL bytes read and 8 NL (five:
// WIRE5) written per foreign record.
template <int NL, bool WIRE5>
__global__ __launch_bounds__(WGB) void kc_shard_pack_kernel(Geom gm, BucketBufs bb, uint32_t me, uint32_t n, uint64_t *segs, uint64_t seg_words,
                                                            const uint64_t *off, const uint64_t *flags, uint32_t Q) {
  static_assert(!WIRE5 || NL == 1, "the compact wire form is one of one-word records");
  __shared__ ScanLDS S;
  __shared__ uint32_t pre[GMAX + 1];
  const uint32_t b = blockIdx.x / Q, q = blockIdx.x % Q, tid = threadIdx.x;
  const uint32_t d = shard_of_bucket(b, gm.P1, n);
  if (d == me || flags[d]) return;
  const uint32_t v = tid < gm.G ? bb.cnt1[(size_t)tid * gm.P1 + b] : 0u;
  const uint32_t e = block_excl_scan(v, S);
  if (tid < gm.G) pre[tid] = e;
  if (tid == 0) pre[gm.G] = S.total;
  __syncthreads();
  const uint32_t lo = shard_first_bucket(d, gm.P1, n), nb = shard_first_bucket(d + 1, gm.P1, n) - lo;
  uint64_t *dst = segs + (size_t)d * seg_words + shard_header_words(nb) + off[b];
  const uint32_t CHm = (1u << gm.log2CH1) - 1u;
  for (uint32_t g = q; g < gm.G; g += Q) {
    const uint32_t cnt = pre[g + 1] - pre[g];
    const uint32_t *chain = bb.chain1 + ((size_t)g * gm.P1 + b) * gm.L1MAX;
    const uint64_t *arena = bb.rec1 + (((size_t)g * gm.A1) << gm.log2CH1) * NL;
    uint64_t *out = dst + (size_t)pre[g] * NL;
    WireRec5 *out5 = reinterpret_cast<WireRec5 *>(dst) + pre[g];
    for (uint32_t i = tid; i < cnt; i += WGB) {
      const uint64_t *src = arena + (((size_t)chain[i >> gm.log2CH1] << gm.log2CH1) + (i & CHm)) * NL;
      if constexpr (WIRE5) {
        WireRec5 w;
        if (gm.rec6) {  // six-byte level-1 records (Geom::rec6): the same 32 bits, the extension codes above the bucket's ten
          const uint8_t *p6 = reinterpret_cast<const uint8_t *>(bb.rec1) +
                              (((((size_t)g * gm.A1) + chain[i >> gm.log2CH1]) << gm.log2CH1) + (i & CHm)) * 6;
          const Rec6 r6 = *reinterpret_cast<const Rec6 *>(p6);
          w.lo = r6.lo;
          w.ext = (uint8_t)((r6.bk >> 10) & 63u);
        } else {
          const uint64_t r = src[0];  // mix << (64 - k2) | extension codes
          w.lo = (uint32_t)(r >> (64u - gm.k2));
          w.ext = (uint8_t)(r & KC_EXT_MASK);
        }
        out5[i] = w;
      } else {
#pragma unroll
        for (int w = 0; w < NL; w++) out[(size_t)i * NL + w] = src[w];
      }
    }
  }
  if constexpr (WIRE5) {  // the spare bytes behind the last record are read (never used) by level 2: defined, not stale
    if (q == 0 && tid < 16) {
      const uint64_t cnt_b = pre[gm.G], nbytes = shard_bucket_words(cnt_b, 1, SHARD_WIRE_COMPACT) * 8;
      if (5 * cnt_b + tid < nbytes) reinterpret_cast<uint8_t *>(dst)[5 * cnt_b + tid] = 0;
    }
  }
}

// after the pack: the foreign chains start again empty and the records that left no longer count as this shard's.  The
// chains of foreign buckets take their chunks from the top of every writer's arena (ChainDest::own_lo): once every one of
// them is empty the top is given back whole, so a shard can take any number of blocks.
__global__ void kc_shard_release_kernel(Geom gm, BucketBufs bb, uint32_t me, uint32_t n, const uint64_t *totals, const uint64_t *flags,
                                        uint64_t *ctrs) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < (size_t)gm.G * gm.P1) {
    const uint32_t d = shard_of_bucket((uint32_t)(i % gm.P1), gm.P1, n);
    if (d != me && !flags[d]) bb.cnt1[i] = 0;
  }
  if (i < gm.G) {
    bool all = true;  // (a destination whose segment was too small keeps its chains: nothing is given back then)
    for (uint32_t d = 0; d < n; d++) all = all && (d == me || !flags[d]);
    if (all) bb.used1[gm.G + i] = 0;
  }
  if (i == 0) {
    uint64_t gone = 0;
    for (uint32_t d = 0; d < n; d++)
      if (d != me && !flags[d]) gone += totals[d];
    atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)(0 - gone));
  }
}

template <int NL, bool CP>
__device__ __forceinline__ void shard_region_of(const Geom &gm, const uint64_t (&r)[NL], uint32_t &b1, size_t &reg) {
  if (CP) {
    const uint64_t m = cp_mix_rec(r[0], gm);
    b1 = cp_b1(m, gm);
    reg = (size_t)b1 * gm.P2 + cp_b2(m, gm);
  } else {
    const uint64_t h = rec_hash<NL>(r);
    b1 = hash_b1(h, gm);
    reg = (size_t)b1 * gm.P2 + hash_b2(h, gm);
  }
}

// The first overflow list after a level-1 pass of the shard flow: what this shard owns goes to its global table (region
// flagged: kc_ovf1_drain_kernel), the rest behind the buckets of its destination's segment as loose k-mer records.
template <int NL, bool CP>
__global__ __launch_bounds__(TPB) void kc_shard_route_ovf1_kernel(Geom gm, BucketBufs bb, uint64_t n_ovf, uint32_t me, uint32_t n, uint64_t *segs,
                                                                  uint64_t seg_words, const uint64_t *wtotals, uint64_t *loose, uint64_t *flags,
                                                                  Table t, uint64_t *ctrs) {
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n_ovf; i += (uint64_t)gridDim.x * TPB) {
    uint64_t r[NL];
    for (int w = 0; w < NL; w++) r[w] = bb.ovf1[i * NL + w];
    uint32_t b1;
    size_t reg;
    shard_region_of<NL, CP>(gm, r, b1, reg);
    const uint32_t d = shard_of_bucket(b1, gm.P1, n);
    if (d == me) {
      bb.flag[reg] = 1;
      table_insert<NL>(t, r, ctrs);
    } else {
      const uint64_t pos = atomicAdd((unsigned long long *)&loose[d], 1ULL);
      const uint32_t nb = shard_first_bucket(d + 1, gm.P1, n) - shard_first_bucket(d, gm.P1, n);
      const uint64_t at = shard_header_words(nb) + wtotals[d] + pos * NL;  // loose records: whole k-mer records behind the record area
      if (!flags[d] && at + NL <= seg_words) {
        uint64_t *dst = segs + (size_t)d * seg_words + at;
        for (int w = 0; w < NL; w++) dst[w] = r[w];
        atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)(0 - 1ULL));
      } else {
        flags[d] = 2;  // no room for the loose records: the caller's segment is too small
      }
    }
  }
}

// ---- receiver ---------------------------------------------------------------------------------------
// (FlatSrc, what level 2 needs to know about the flat sources, is declared in kc_bucketed.hpp)
// a received segment becomes flat source f: per bucket its count and the address of its first record
__global__ __launch_bounds__(WGB) void kc_shard_index_kernel(const uint64_t *seg, uint32_t nb, uint32_t nl, uint32_t wire, uint32_t *cnt,
                                                             uint64_t *at, uint64_t *ctrs) {
  __shared__ uint32_t s_n[PMAX];
  const uint32_t *counts = reinterpret_cast<const uint32_t *>(seg + SHARD_HDR);
  for (uint32_t i = threadIdx.x; i < nb; i += WGB) s_n[i] = counts[i];
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint64_t *recs = seg + shard_header_words(nb);
    uint64_t run = 0, wrun = 0;
    for (uint32_t i = 0; i < nb; i++) {
      cnt[i] = s_n[i];
      at[i] = (uint64_t)(uintptr_t)(recs + wrun);
      run += s_n[i];
      wrun += shard_bucket_words(s_n[i], nl, wire);
    }
    // (CTR_EXPECT bounds what level 1 of THIS shard has buffered: received records never pass through it)
    atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)run);
  }
}

// the loose records of a received segment: into the global table, their regions flagged
template <int NL, bool CP>
__global__ __launch_bounds__(TPB) void kc_shard_loose_kernel(Geom gm, BucketBufs bb, const uint64_t *recs, uint64_t n, Table t, uint64_t *ctrs) {
  for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < n; i += (uint64_t)gridDim.x * TPB) {
    uint64_t r[NL];
    for (int w = 0; w < NL; w++) r[w] = recs[i * NL + w];
    uint32_t b1;
    size_t reg;
    shard_region_of<NL, CP>(gm, r, b1, reg);
    bb.flag[reg] = 1;
    table_insert<NL>(t, r, ctrs);
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    atomicAdd((unsigned long long *)&ctrs[CTR_INSERTED], (unsigned long long)n);
  }
}

}  // namespace kc
