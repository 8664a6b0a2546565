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

// target rank of the k-mer starting at every position of the block, -1 where there is none: the window of the k-mer
// and its two neighbours must lie inside one read (S1, S5)
template <int NL>
__global__ void kc_supermer_targets_kernel(const uint8_t *seqs, uint64_t len, int k, uint32_t rank_n, int32_t *targets, uint64_t *bad) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= len) return;
  int32_t t = -1;
  if (p >= 1 && p + k < len) {
    bool ok = true;
    for (int i = -1; i <= k && ok; i++) {
      const uint8_t c = seqs[p + i];
      if (c == '_') ok = false;
      else if (!sm_is_base(c)) { ok = false; *bad = 1; }
    }
    if (ok) {
      uint64_t f[NL], r[NL];
#pragma unroll
      for (int j = 0; j < NL; j++) f[j] = 0;
      for (int i = 0; i < k; i++) {
        const uint64_t code = kc_base_code(seqs[p + i]);
#pragma unroll
        for (int j = 0; j < NL; j++)
          if (j == (i >> 5)) f[j] |= code << (62 - 2 * (i & 31));
      }
      kc_revcomp<NL>(f, k, r);
      t = (int32_t)kc_reference_owner<NL>(f, r, k, rank_n);
    }
  }
  targets[p] = t;
}

// one thread per position: a k-mer whose predecessor has another target (or none) starts a supermer and walks to its end
__global__ void kc_supermer_build_kernel(const int32_t *targets, uint64_t len, int k, SupermerInfo *out, uint32_t cap, uint32_t *n_out,
                                         uint32_t *n_kmers, uint32_t *too_long) {
  const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= len) return;
  const int32_t t = targets[p];
  if (t < 0) return;
  atomicAdd(n_kmers, 1u);
  if (p > 0 && targets[p - 1] == t) return;
  uint64_t e = p + 1;
  while (e < len && targets[e] == t) e++;
  if (e - p + (uint64_t)k + 1 > 65535) {  // the length travels in 16 bits (reads are short; the reference has the same limit)
    *too_long = 1;
    return;
  }
  const uint32_t slot = atomicAdd(n_out, 1u);
  if (slot < cap) {
    out[slot].target = t;
    out[slot].offset = (int32_t)(p - 1);
    out[slot].len = (uint16_t)(e - p + k + 1);
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

__global__ void kc_pack_seqs_kernel(const uint8_t *seqs, uint64_t len, uint8_t *packed) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (len + 1) / 2) return;
  const uint8_t hi = sm_nibble(seqs[2 * i]), lo = 2 * i + 1 < len ? sm_nibble(seqs[2 * i + 1]) : 0;
  packed[i] = (uint8_t)((hi << 4) | lo);
}

// packed supermers (as cut by kcount_gpu.cpp:153-161, joined by the byte '_') -> ASCII block, two characters per byte
__global__ void kc_unpack_supermers_kernel(const uint8_t *packed, uint64_t len, uint8_t *seqs, uint64_t *bad) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= len) return;
  const uint8_t b = packed[i];
  uint8_t c0 = '_', c1 = '_';
  if (b != '_') {
    const uint32_t hi = b >> 4, lo = b & 15u;
    if (hi > 9 || lo > 9) *bad = 1;
    const char to_base[10] = {'_', 'a', 'c', 'g', 't', 'A', 'C', 'G', 'T', 'N'};  // gpu_hash_table.cpp:270
    c0 = hi <= 9 ? (uint8_t)to_base[hi] : (uint8_t)'_';
    c1 = lo <= 9 ? (uint8_t)to_base[lo] : (uint8_t)'_';
  }
  seqs[2 * i] = c0;
  seqs[2 * i + 1] = c1;
}

}  // namespace kc
