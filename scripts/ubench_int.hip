// micro-benchmark: integer op throughput on gfx950 (which hash arithmetic is affordable per k-mer)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ void k(uint64_t *out, int iters) {
  uint64_t a = threadIdx.x * 0x9E3779B97F4A7C15ULL + blockIdx.x, b = a ^ 0x1234567, c = a + 99, d = b * 3;
  uint32_t x = (uint32_t)a, y = (uint32_t)b, z = (uint32_t)c, w = (uint32_t)d;
  for (int i = 0; i < iters; i++) {
    if (OP == 0) { x = x * 0x85ebca6bu + 1; y = y * 0xc2b2ae35u + 1; z = z * 0x85ebca6bu + 3; w = w * 0xc2b2ae35u + 7; }          // 4 mul_lo_u32 (+add)
    if (OP == 1) { x = (x ^ (x >> 15)) + y; y = (y ^ (y >> 13)) + z; z = (z ^ (z >> 16)) + w; w = (w ^ (w >> 11)) + x; }      // 4x (shift, xor, add)
    if (OP == 2) { a *= 0xff51afd7ed558ccdULL; b *= 0xc4ceb9fe1a85ec53ULL; c *= 0xff51afd7ed558ccdULL; d *= 0xc4ceb9fe1a85ec53ULL; } // 4 mul64
    if (OP == 3) { a ^= a >> 33; a *= 0xff51afd7ed558ccdULL; a ^= a >> 33; a *= 0xc4ceb9fe1a85ec53ULL; a ^= a >> 33;
                   b ^= b >> 33; b *= 0xff51afd7ed558ccdULL; b ^= b >> 33; b *= 0xc4ceb9fe1a85ec53ULL; b ^= b >> 33; }          // 2 mix64
    if (OP == 4) { x = __umulhi(x, 0x85ebca6bu) + y; y = __umulhi(y, 0xc2b2ae35u) + z; z = __umulhi(z, 0x85ebca6bu) + w; w = __umulhi(w, 0xc2b2ae35u) + x; } // 4 mul_hi
    if (OP == 5) { a = (uint64_t)x * y + a; b = (uint64_t)y * z + b; c = (uint64_t)z * w + c; d = (uint64_t)w * x + d; x += 3; y += 5; z += 7; w += 11; } // 4 mad_u64_u32
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + x + y + z + w;
}
template <int OP> void run(const char *name, double ops_per_iter) {
  uint64_t *o; hipMalloc(&o, 256 * 8 * 1024 * 8);
  int iters = 4096;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256 * 8, 1024>>>(o, 16);
  hipEventRecord(e0); k<OP><<<256 * 8, 1024>>>(o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double lanes = 256.0 * 8 * 1024 * iters * ops_per_iter;
  printf("%-28s %8.3f ms  %8.2f T lane-ops/s  (%.2f cycles per wave-instr per SIMD at 2.4GHz)\n", name, ms, lanes / ms / 1e9,
         (ms * 1e-3 * 2.4e9) / (256.0 * 8 * 1024 / 64 * iters * ops_per_iter / (256 * 4)));
  hipFree(o);
}
int main() {
  run<0>("mul_lo_u32 (+add)", 4); run<1>("shift+xor+add", 4); run<2>("mul64", 4); run<3>("mix64 (murmur fmix)", 2);
  run<4>("mul_hi_u32 (+add)", 4); run<5>("mad_u64_u32", 4);
  return 0;
}
