// kc_supermer.hpp -- the reference's wire format on the device, for runs mixed with unmodified MHM2 ranks.
//
// Sender (replaces parse_and_pack / build_supermers / pack_seqs, src/kcount/kcount-gpu/parse_and_pack.cpp:127-237, with
// the CPU backend's semantics -- SeqBlockInserter::process_seq, src/kcount/kcount_cpu.cpp:73-103): every k-mer of a
// '_'-joined, case-masked block that has both neighbours gets the reference's target rank (quick_hash of the minimizer
// of its canonical form, modulo the ranks); a supermer is a maximal run of consecutive k-mers of one read with one
// target, written down as {target, offset, len} over the block: its first k-mer's left neighbour up to its last
// k-mer's right neighbour.  The block itself is packed two characters per byte with the reference's nibble codes
// (_ 0, acgt 1-4, ACGT 5-8, N 9: parse_and_pack.cpp:196-213); the host cuts each supermer's bytes out of it and masks
// the odd nibbles (src/kcount/kcount_gpu.cpp:153-161).
// Receiver (replaces gpu_unpack_supermer_block, gpu_hash_table.cpp:281-292): packed bytes back to the '_'-joined
// ASCII block that the extraction kernels read; the byte '_' that HashTableGPUDriver::insert_supermer puts between
// two supermers (gpu_hash_table.cpp:681-695) becomes two separators.
#pragma once
#include "kc_common.hpp"

namespace kc {

struct SupermerInfo {  // layout of kcount_gpu::SupermerInfo (parse_and_pack.hpp:50-54) = kc_supermer of the C ABI
  int32_t target;
  int32_t offset;
  uint16_t len;
};

__device__ __forceinline__ bool sm_is_base(uint8_t c) {
  const uint8_t u = c & 0xDFu;
  return u == 'A' || u == 'C' || u == 'G' || u == 'T' || u == 'N';
}

// ---- sender: target of every k-mer ---------------------------------------------------------------------------------
// The reference's target is quick_hash(minimizer) % ranks, the minimizer being the largest of min(m-mer, its reverse
// complement) over the k - m + 1 m-mers of the k-mer (get_minimizer_fast, kmer.cpp:349-398; kc_reference_owner).  That
// min belongs to a POSITION of the block, not to a k-mer: a workgroup stages the characters of its tile once (2-bit
// codes, 16 to a word; a running count of the characters that are no bases), every position gets its canonical m-mer
// (one 64-bit window of the codes: m <= 27) into LDS, and a k-mer's minimizer is the largest of k - m + 1 consecutive
// ones -- one byte read from memory per character and k - m + 1 LDS reads per k-mer, where the first version of this
// kernel read k + 2 bytes from memory and built both strands of the whole k-mer per position.
constexpr int SM_WG = 256;
constexpr int SM_TILE = 3840;  // k-mer start positions per workgroup (15 per thread)
constexpr int SM_LEAD = 16;    // characters staged before the tile's first position (the left neighbour is one of them)
constexpr int SM_HALO = 160;   // ... and behind its last one: k at least
constexpr int SM_NC = SM_LEAD + SM_TILE + SM_HALO;  // staged characters: 251 groups of 16, one per thread
static_assert(SM_NC % 16 == 0 && SM_NC / 16 + 3 <= SM_WG && SM_TILE % SM_WG == 0, "one group of characters per thread");

struct SmLDS {
  uint64_t least[SM_NC];             // min(m-mer starting here, its reverse complement), first base highest
  uint32_t codes[SM_NC / 16 + 3];    // 16 bases per word, first base highest
  uint16_t pre[SM_NC + 16];          // characters before this one that are no bases ('_', anything else, outside the block)
  uint32_t wsum[SM_WG / 64];
};

__global__ __launch_bounds__(SM_WG) void kc_supermer_targets_kernel(const uint8_t *seqs, uint64_t len, int k, uint32_t rank_n,
                                                                    int32_t *targets, uint64_t *bad) {
  __shared__ SmLDS L;
  constexpr int G = SM_NC / 16;
  const int t = threadIdx.x, lane = t & 63;
  const int64_t c0 = (int64_t)blockIdx.x * SM_TILE - SM_LEAD;  // block position of staged character 0
  const int m = kc_minimizer_len(k), ncand = k - m + 1;
  // stage: thread t takes characters 16t .. 16t+15
  uint32_t code = 0, stop = 0;
  if (t < G) {
    const int64_t q = c0 + 16 * t;
    uint32_t w[4];
    if (q >= 0 && q + 16 <= (int64_t)len && (((uintptr_t)(seqs + q)) & 15u) == 0) {
      const uint4 v = *reinterpret_cast<const uint4 *>(seqs + q);
      w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; j++) {
        w[j] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int64_t pos = q + 4 * j + i;
          const uint32_t c = (pos >= 0 && pos < (int64_t)len) ? seqs[pos] : (uint32_t)'_';
          w[j] |= c << (8 * i);
        }
      }
    }
    bool anybad = false;
#pragma unroll
    for (int i = 0; i < 16; i++) {
      const uint32_t c = (w[i >> 2] >> (8 * (i & 3))) & 0xFFu;
      const bool isb = sm_is_base((uint8_t)c);
      code |= kc_base_code(c) << (30 - 2 * i);  // N counts as G inside a k-mer (S3)
      stop |= (isb ? 0u : 1u) << i;
      anybad |= !isb && c != '_';
    }
    if (anybad) *bad = 1;  // the reference's kernel has no code for such a character either (parse_and_pack.cpp:196-213)
  }
  if (t < G + 3) L.codes[t] = code;
  // running count of the stops: scan over the threads' counts, then inside each group
  const uint32_t cnt = __popc(stop);
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t v = __shfl_up(incl, o);
    if (lane >= o) incl += v;
  }
  if (lane == 63) L.wsum[t >> 6] = incl;
  __syncthreads();
  uint32_t before = incl - cnt;
  for (int w = 0; w < (t >> 6); w++) before += L.wsum[w];
  if (t <= G) {  // thread G: nothing of its own, the total for the sixteen entries behind the last character
    uint32_t *pre32 = reinterpret_cast<uint32_t *>(L.pre) + 8 * t;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      const uint32_t a = before + __popc(stop & ((1u << i) - 1u)), b = before + __popc(stop & ((2u << i) - 1u));
      pre32[i >> 1] = a | (b << 16);
    }
  }
  __syncthreads();
  // the canonical m-mer of every staged position (consecutive lanes, consecutive positions)
  const uint64_t mmask = ~0ULL << (64 - 2 * m);
  for (int c = t; c < SM_NC; c += SM_WG) {
    const int g = c >> 4, s = 2 * (c & 15);
    const uint64_t hi = ((uint64_t)L.codes[g] << 32) | L.codes[g + 1];
    const uint64_t win = (s ? ((hi << s) | ((uint64_t)L.codes[g + 2] >> (32 - s))) : hi) & mmask;
    const uint64_t rc = kc_rc_word(win) << (64 - 2 * m);
    L.least[c] = win < rc ? win : rc;
  }
  __syncthreads();
  // the k-mers: the window of a k-mer and its two neighbours must hold bases only (inside one read: S1, S5)
#pragma unroll 1
  for (int j = 0; j < SM_TILE / SM_WG; j++) {
    const int c = SM_LEAD + t + SM_WG * j;
    const int64_t p = c0 + c;
    if (p >= (int64_t)len) break;
    int32_t tg = -1;
    if (L.pre[c + k + 1] == L.pre[c - 1]) {
      uint64_t best = 0;
      for (int i = 0; i < ncand; i++) {
        const uint64_t v = L.least[c + i];
        best = v > best ? v : best;
      }
      tg = (int32_t)(kc_quick_hash(best) % (uint64_t)rank_n);
    }
    targets[p] = tg;
  }
}

// A k-mer whose predecessor has another target (or none) starts a supermer and walks to its end.  A workgroup takes
// SB_WG * SB_PER consecutive positions and bumps the two global counters once: first it counts, then it hands out its
// slots.  (One bump per wave was what the kernel's time consisted of: a returning atomic on one address is served every
// ten nanoseconds or so, 95 ms for the 4 M waves of a 268 MB block.)
constexpr int SB_WG = 1024, SB_PER = 16;
__global__ __launch_bounds__(SB_WG) void kc_supermer_build_kernel(const int32_t *targets, uint64_t len, int k, SupermerInfo *out, uint32_t cap,
                                                                  uint32_t *n_out, uint32_t *n_kmers, uint32_t *too_long) {
  __shared__ uint32_t s_kmers, s_starts, s_base, s_next;
  const int tid = threadIdx.x;
  const uint32_t lane = lane_id();
  if (tid == 0) s_kmers = s_starts = s_next = 0;
  __syncthreads();
  const uint64_t p0 = (uint64_t)blockIdx.x * (SB_WG * SB_PER);
  uint32_t starts = 0, nk = 0;  // bit j: position p0 + j * SB_WG + tid starts a supermer
#pragma unroll
  for (int j = 0; j < SB_PER; j++) {
    const uint64_t p = p0 + (uint64_t)j * SB_WG + tid;
    const int32_t t = p < len ? targets[p] : -1;
    const bool start = t >= 0 && (p == 0 || targets[p - 1] != t);
    nk += t >= 0 ? 1u : 0u;
    starts |= (start ? 1u : 0u) << j;
  }
  uint32_t ns = __popc(starts);
  for (int o = 32; o > 0; o >>= 1) {
    nk += __shfl_down(nk, o);
    ns += __shfl_down(ns, o);
  }
  if (lane == 0) {
    if (nk) atomicAdd(&s_kmers, nk);
    if (ns) atomicAdd(&s_starts, ns);
  }
  __syncthreads();
  if (tid == 0) {
    if (s_kmers) atomicAdd(n_kmers, s_kmers);
    s_base = s_starts ? atomicAdd(n_out, s_starts) : 0u;
  }
  __syncthreads();
  const uint32_t base = s_base;
#pragma unroll 1
  for (int j = 0; j < SB_PER; j++) {
    const bool start = (starts >> j) & 1u;
    const uint64_t ms = __ballot(start);
    if (!ms) continue;
    uint32_t wbase = 0;
    if (lane == 0) wbase = atomicAdd(&s_next, (uint32_t)__popcll(ms));
    wbase = (uint32_t)__shfl((int)wbase, 0);
    if (start) {
      const uint64_t p = p0 + (uint64_t)j * SB_WG + tid;
      const int32_t t = targets[p];
      uint64_t e = p + 1;
      while (e < len && targets[e] == t) e++;
      const uint32_t slot = base + wbase + (uint32_t)__popcll(ms & ((1ULL << lane) - 1ULL));
      if (e - p + (uint64_t)k + 1 > 65535) {  // the length travels in 16 bits (reads are short; the reference has the same limit)
        *too_long = 1;
      } else if (slot < cap) {
        out[slot].target = t;
        out[slot].offset = (int32_t)(p - 1);
        out[slot].len = (uint16_t)(e - p + k + 1);
      }
    }
  }
}

__device__ __forceinline__ uint8_t sm_nibble(uint8_t c) {  // parse_and_pack.cpp:196-213
  switch (c) {
    case 'a': return 1;
    case 'c': return 2;
    case 'g': return 3;
    case 't': return 4;
    case 'A': return 5;
    case 'C': return 6;
    case 'G': return 7;
    case 'T': return 8;
    case 'N':
    case 'n': return 9;
    default: return 0;  // '_' and the end of the block
  }
}

// the same without a branch per letter
__device__ __forceinline__ uint32_t sm_nibble_of(uint32_t c) {
  const uint32_t u = c & 0xDFu;
  const bool acgt = u == 'A' || u == 'C' || u == 'G' || u == 'T';
  return acgt ? ((c & 0x20u) ? 1u : 5u) + kc_base_code(c) : (u == 'N' ? 9u : 0u);
}

// sixteen characters per thread -> eight bytes
__global__ void kc_pack_seqs_kernel(const uint8_t *seqs, uint64_t len, uint8_t *packed) {
  const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  if (i >= len) return;
  if (i + 16 <= len && (((uintptr_t)(seqs + i)) & 15u) == 0 && (((uintptr_t)(packed + i / 2)) & 7u) == 0) {
    const uint4 v = *reinterpret_cast<const uint4 *>(seqs + i);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[2] = {0, 0};
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
      const uint32_t hi = sm_nibble_of((w[j >> 2] >> (8 * (j & 3))) & 0xFFu), lo = sm_nibble_of((w[j >> 2] >> (8 * (j & 3) + 8)) & 0xFFu);
      o[j >> 3] |= ((hi << 4) | lo) << (8 * ((j >> 1) & 3));
    }
    *reinterpret_cast<uint2 *>(packed + i / 2) = make_uint2(o[0], o[1]);
    return;
  }
  for (uint64_t j = i; j < i + 16 && j < len; j += 2) {
    const uint8_t hi = sm_nibble(seqs[j]), lo = j + 1 < len ? sm_nibble(seqs[j + 1]) : 0;
    packed[j / 2] = (uint8_t)((hi << 4) | lo);
  }
}

// packed supermers (as cut by kcount_gpu.cpp:153-161, joined by the byte '_') -> ASCII block, two characters per byte;
// eight bytes per thread
__global__ void kc_unpack_supermers_kernel(const uint8_t *packed, uint64_t len, uint8_t *seqs, uint64_t *bad) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (i0 >= len) return;
  const bool fast = i0 + 8 <= len && (((uintptr_t)(packed + i0)) & 7u) == 0 && (((uintptr_t)(seqs + 2 * i0)) & 15u) == 0;
  uint32_t in[2] = {0, 0};
  const int n = fast ? 8 : (int)(len - i0 < 8 ? len - i0 : 8);
  if (fast) {
    const uint2 v = *reinterpret_cast<const uint2 *>(packed + i0);
    in[0] = v.x;
    in[1] = v.y;
  } else {
    for (int j = 0; j < n; j++) in[j >> 2] |= (uint32_t)packed[i0 + j] << (8 * (j & 3));
  }
  // gpu_hash_table.cpp:270: "_acgtACGTN" by nibble, as two 40-bit tables of low and high character bits would be no
  // shorter than this: 0 -> '_'; 1-4 -> "acgt"; 5-8 -> "ACGT"; 9 -> 'N'
  auto to_char = [](uint32_t x) -> uint32_t {
    const uint32_t letters = 0x54474341u;  // "ACGT", first letter lowest
    const uint32_t b = (x - 1u) & 3u;
    const uint32_t up = (letters >> (8 * b)) & 0xFFu;
    return x == 0 ? (uint32_t)'_' : x <= 4 ? (up | 0x20u) : x <= 8 ? up : (uint32_t)'N';
  };
  uint32_t out[4] = {0, 0, 0, 0};
  bool anybad = false;
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t b = (in[j >> 2] >> (8 * (j & 3))) & 0xFFu;
    uint32_t c0 = '_', c1 = '_';
    if (b != '_') {
      const uint32_t hi = b >> 4, lo = b & 15u;
      anybad |= j < n && (hi > 9 || lo > 9);
      c0 = hi <= 9 ? to_char(hi) : (uint32_t)'_';
      c1 = lo <= 9 ? to_char(lo) : (uint32_t)'_';
    }
    out[j >> 1] |= (c0 | (c1 << 8)) << (16 * (j & 1));
  }
  if (anybad) *bad = 1;
  if (fast) {
    *reinterpret_cast<uint4 *>(seqs + 2 * i0) = make_uint4(out[0], out[1], out[2], out[3]);
  } else {
    for (int j = 0; j < 2 * n; j++) seqs[2 * i0 + j] = (uint8_t)(out[j >> 2] >> (8 * (j & 3)));
  }
}

}  // namespace kc
