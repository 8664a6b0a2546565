// ref_gpu_common.hip -- TEST INFRASTRUCTURE ONLY (part of the oracle; never linked by the product).
//
// Pins S3 (2-bit packing), S4 (reverse complement) and the slot hash to the REFERENCE'S OWN device code: this file
// #includes, by path and unmodified, the two headers of the reference's GPU backend that need nothing but the HIP
// runtime,
//     <reference>/src/gpu-utils/gpu_common.hpp                 (pack_seq_to_kmer :199-231, revcomp :181-197,
//                                                               comp_nucleotide :158-179)
//     <reference>/src/kcount/kcount-gpu/gpu_hash_funcs.hpp     (gpu_murmurhash3_64 :59-143, the hash kmer_hash of
//                                                               gpu_hash_table.cpp:121-124 calls)
// and wraps them in one kernel and one C entry point.  No reference source is copied and no stand-in header is written:
// `make -C oracle ref` compiles this file with
//     hipcc -DHIP_GPU --offload-arch=gfx950 -I<reference>/src
// into oracle/_ref/libref_gpu_common.so (git-ignored; travels to the GPU box with gpurun; built only where the
// reference checkout exists).  tests/test_gpu_ref_pin.py compares orc_pack_kmer / orc_revcomp / orc_kmer_hash and the
// HIP path's canonical records with what it returns.  (The reference's GPU twin rejects any non-ACGT character, F4a:
// the comparison is made on ACGT-only input; N -> G stays pinned by the survey's known answers.)
#include <stdint.h>
#include <hip/hip_runtime.h>

#include "gpu-utils/gpu_common.hpp"
#include "kcount/kcount-gpu/gpu_hash_funcs.hpp"

namespace {

constexpr int MAXL = 8;  // words per k-mer this wrapper handles (k <= 255)

// thread i: the k-mer starting at seqs[i]
__global__ void ref_kmers_kernel(char *seqs, int npos, int kmer_len, int num_longs, uint64_t *kmers, uint64_t *rcs, uint64_t *hashes,
                                 uint64_t *rc_hashes, uint8_t *ok) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npos) return;
  uint64_t kmer[MAXL], rc[MAXL];
  for (int j = 0; j < MAXL; j++) kmer[j] = rc[j] = 0;
  const bool good = gpu_common::pack_seq_to_kmer(seqs + i, kmer_len, num_longs, kmer);
  ok[i] = good ? 1 : 0;
  if (!good) return;
  gpu_common::revcomp(kmer, rc, kmer_len, num_longs);
  for (int j = 0; j < num_longs; j++) {
    kmers[(size_t)i * num_longs + j] = kmer[j];
    rcs[(size_t)i * num_longs + j] = rc[j];
  }
  hashes[i] = gpu_murmurhash3_64(reinterpret_cast<const void *>(kmer), (uint32_t)(num_longs * sizeof(uint64_t)));
  rc_hashes[i] = gpu_murmurhash3_64(reinterpret_cast<const void *>(rc), (uint32_t)(num_longs * sizeof(uint64_t)));
}

__global__ void ref_comp_kernel(const char *in, int n, char *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = gpu_common::comp_nucleotide(in[i]);
}

#define REF_TRY(x)                    \
  do {                                \
    const hipError_t e_ = (x);        \
    if (e_ != hipSuccess) {           \
      rc_ = (int)e_;                  \
      goto done;                      \
    }                                 \
  } while (0)

}  // namespace

// Host buffers in, host buffers out.  seqs: len characters; position i < len - kmer_len + 1 gives k-mer i.
// kmers / rcs: npos * num_longs words; hashes / rc_hashes: npos; ok: npos (0 = the reference rejected the window).
// Returns 0, or the HIP error code.
extern "C" int ref_gpu_kmers(const char *seqs, int len, int kmer_len, int num_longs, uint64_t *kmers, uint64_t *rcs, uint64_t *hashes,
                             uint64_t *rc_hashes, uint8_t *ok) {
  int rc_ = 0;
  const int npos = len - kmer_len + 1;
  if (npos <= 0 || num_longs > MAXL || num_longs != kmer_len / 32 + 1) return -1;
  char *d_seqs = nullptr;
  uint64_t *d_k = nullptr, *d_r = nullptr, *d_h = nullptr, *d_rh = nullptr;
  uint8_t *d_ok = nullptr;
  const size_t kb = (size_t)npos * num_longs * sizeof(uint64_t), hb = (size_t)npos * sizeof(uint64_t);
  REF_TRY(hipMalloc(&d_seqs, (size_t)len));
  REF_TRY(hipMalloc(&d_k, kb));
  REF_TRY(hipMalloc(&d_r, kb));
  REF_TRY(hipMalloc(&d_h, hb));
  REF_TRY(hipMalloc(&d_rh, hb));
  REF_TRY(hipMalloc(&d_ok, (size_t)npos));
  REF_TRY(hipMemcpy(d_seqs, seqs, (size_t)len, hipMemcpyHostToDevice));
  REF_TRY(hipMemset(d_k, 0, kb));
  REF_TRY(hipMemset(d_r, 0, kb));
  REF_TRY(hipMemset(d_h, 0, hb));
  REF_TRY(hipMemset(d_rh, 0, hb));
  hipLaunchKernelGGL(ref_kmers_kernel, dim3((npos + 255) / 256), dim3(256), 0, 0, d_seqs, npos, kmer_len, num_longs, d_k, d_r, d_h, d_rh, d_ok);
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(kmers, d_k, kb, hipMemcpyDeviceToHost));
  REF_TRY(hipMemcpy(rcs, d_r, kb, hipMemcpyDeviceToHost));
  REF_TRY(hipMemcpy(hashes, d_h, hb, hipMemcpyDeviceToHost));
  REF_TRY(hipMemcpy(rc_hashes, d_rh, hb, hipMemcpyDeviceToHost));
  REF_TRY(hipMemcpy(ok, d_ok, (size_t)npos, hipMemcpyDeviceToHost));
done:
  (void)hipFree(d_seqs);
  (void)hipFree(d_k);
  (void)hipFree(d_r);
  (void)hipFree(d_h);
  (void)hipFree(d_rh);
  (void)hipFree(d_ok);
  return rc_;
}

// comp_nucleotide over n characters
extern "C" int ref_gpu_comp(const char *in, int n, char *out) {
  int rc_ = 0;
  char *d_in = nullptr, *d_out = nullptr;
  if (n <= 0) return -1;
  REF_TRY(hipMalloc(&d_in, (size_t)n));
  REF_TRY(hipMalloc(&d_out, (size_t)n));
  REF_TRY(hipMemcpy(d_in, in, (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(ref_comp_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, d_in, n, d_out);
  REF_TRY(hipGetLastError());
  REF_TRY(hipDeviceSynchronize());
  REF_TRY(hipMemcpy(out, d_out, (size_t)n, hipMemcpyDeviceToHost));
done:
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return rc_;
}
