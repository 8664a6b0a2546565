// Would 6-byte level-1 records pay?  The two access shapes they need, against the 8-byte ones the kernels use now, per
// record (not per byte):
//   stores: every workgroup appends runs of 16 records round-robin to 1024 streams of its own, a lane stores a PAIR of
//           records -- 16 bytes at an 8-byte aligned address (now) or 12 bytes (global_store_dwordx3) at an address that
//           is only 2-byte aligned (6-byte records); runs start where the last one ended.
//   loads:  every workgroup streams records of long contiguous chains, a lane loads one record -- 8 bytes aligned (now) or
//           the 8 bytes at a 6-byte stride (one unaligned global_load_dwordx2 per record, masked to 48 bits).
// hipcc -O3 --offload-arch=gfx950 scripts/ubench_rec6.hip -o ubench_rec6 && ./ubench_rec6
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int WG = 1024, STREAMS = 1024;
struct __attribute__((packed, aligned(2))) P12 { uint32_t a, b, c; };
struct __attribute__((aligned(8))) P16 { uint64_t a, b; };
typedef uint64_t __attribute__((aligned(1))) u64u;

// RB = bytes per record (6 or 8); runs of 16 records (8 pairs) per stream and round; lane t of the workgroup: pair t & 7
// of stream t >> 3 (+128 per trip)
template <int RB>
__global__ __launch_bounds__(WG) void store_kernel(uint8_t *buf, size_t stream_bytes, int rounds) {
  uint8_t *mine = buf + (size_t)blockIdx.x * STREAMS * stream_bytes;
  const int t = threadIdx.x;
  for (int r = 0; r < rounds; r++) {
    // a run's start: where the last one ended, plus a per-stream offset so that runs start anywhere
#pragma unroll
    for (int trip = 0; trip < 8; trip++) {
      const int s = (t >> 3) + 128 * trip;
      const size_t at = (size_t)s * stream_bytes + ((size_t)r * 16 + (s * 5 & 15)) * RB + (size_t)(t & 7) * 2 * RB;
      if (RB == 8) {
        P16 v = {(uint64_t)r, (uint64_t)t};
        *reinterpret_cast<P16 *>(mine + at) = v;
      } else {
        P12 v = {(uint32_t)r, (uint32_t)t, (uint32_t)s};
        *reinterpret_cast<P12 *>(mine + at) = v;
      }
    }
  }
}

template <int RB>
__global__ __launch_bounds__(WG) void load_kernel(const uint8_t *buf, size_t bytes_per_wg, uint64_t *sink) {
  const uint8_t *mine = buf + (size_t)blockIdx.x * bytes_per_wg;
  const size_t nrec = bytes_per_wg / RB - 2;
  uint64_t acc = 0;
  for (size_t i0 = 0; i0 + (size_t)WG * 16 <= nrec; i0 += (size_t)WG * 16) {
    uint64_t v[16];
#pragma unroll
    for (int j = 0; j < 16; j++) v[j] = *reinterpret_cast<const u64u *>(mine + (i0 + (size_t)j * WG + threadIdx.x) * RB);
#pragma unroll
    for (int j = 0; j < 16; j++) acc += RB == 6 ? (v[j] & 0xFFFFFFFFFFFFULL) : v[j];
  }
  if (acc == 0x123456789ULL) sink[0] = acc;
}

// two 6-byte records per lane: one aligned 12-byte load (global_load_dwordx3)
struct __attribute__((aligned(4))) L12 { uint32_t a, b, c; };
__global__ __launch_bounds__(WG) void load_pairs_kernel(const uint8_t *buf, size_t bytes_per_wg, uint64_t *sink) {
  const uint8_t *mine = buf + (size_t)blockIdx.x * bytes_per_wg;
  const size_t npair = bytes_per_wg / 12 - 2;
  uint64_t acc = 0;
  for (size_t i0 = 0; i0 + (size_t)WG * 8 <= npair; i0 += (size_t)WG * 8) {
    L12 v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = *reinterpret_cast<const L12 *>(mine + (i0 + (size_t)j * WG + threadIdx.x) * 12);
#pragma unroll
    for (int j = 0; j < 8; j++) acc += (uint64_t)v[j].a + v[j].b + v[j].c;
  }
  if (acc == 0x123456789ULL) sink[0] = acc;
}

int main() {
  const int G = 256, rounds = 1500;
  const size_t stream_bytes = (size_t)(rounds + 2) * 16 * 8 + 256;  // room for the 8-byte runs; the 6-byte ones use 3/4 of it
  uint8_t *buf;
  uint64_t *sink;
  const size_t total = (size_t)G * STREAMS * stream_bytes;
  CK(hipMalloc(&buf, total));
  CK(hipMalloc(&sink, 8));
  CK(hipMemset(buf, 0, total));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double recs = (double)G * STREAMS * 16.0 * rounds;
  for (int rep = 0; rep < 3; rep++) {
    float ms8, ms6;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(store_kernel<8>, dim3(G), dim3(WG), 0, 0, buf, stream_bytes, rounds);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms8, e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(store_kernel<6>, dim3(G), dim3(WG), 0, 0, buf, stream_bytes, rounds);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms6, e0, e1));
    printf("stores, runs of 16 records to 1024 streams per workgroup: 8-byte records %.2f ms = %.1f G records/s (%.2f TB/s); 6-byte %.2f ms = %.1f G records/s (%.2f TB/s)\n",
           ms8, recs / ms8 / 1e6, recs * 8 / ms8 / 1e9, ms6, recs / ms6 / 1e6, recs * 6 / ms6 / 1e9);
  }
  const size_t per_wg = total / G;
  for (int rep = 0; rep < 3; rep++) {
    float ms8, ms6;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(load_kernel<8>, dim3(G), dim3(WG), 0, 0, buf, per_wg, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms8, e0, e1));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(load_kernel<6>, dim3(G), dim3(WG), 0, 0, buf, per_wg, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms6, e0, e1));
    printf("loads, one record per lane from contiguous chains: 8-byte records %.2f ms = %.1f G records/s (%.2f TB/s); 6-byte (unaligned 8-byte load) %.2f ms = %.1f G records/s (%.2f TB/s)\n",
           ms8, (double)(per_wg / 8) * G / ms8 / 1e6, (double)total / ms8 / 1e9, ms6, (double)(per_wg / 6) * G / ms6 / 1e6, (double)total / ms6 / 1e9);
  }
  for (int rep = 0; rep < 3; rep++) {
    float ms;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(load_pairs_kernel, dim3(G), dim3(WG), 0, 0, buf, per_wg, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("loads, two 6-byte records per lane (one aligned 12-byte load): %.2f ms = %.1f G records/s (%.2f TB/s)\n", ms,
           (double)(per_wg / 6) * G / ms / 1e6, (double)total / ms / 1e9);
  }
  return 0;
}
