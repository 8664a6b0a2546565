// micro-benchmark: LDS op throughput on gfx950 with random (hashed) addresses, 16 waves per CU, one workgroup per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, int iters, uint32_t mask) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 32768; i += 1024) lds[i] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
  unsigned long long *l64 = (unsigned long long *)lds;
  for (int i = 0; i < iters; i++) {
    x = x * 1664525u + 1013904223u;
    uint32_t a = (x >> 8) & mask;
    if (OP == 0) atomicAdd(&lds[a], 1u);                       // ds_add_u32 no return
    if (OP == 1) acc += atomicAdd(&lds[a], 1u);                // ds_add_rtn_u32
    if (OP == 2) lds[a] = x;                                   // ds_write_b32
    if (OP == 3) acc += lds[a];                                // ds_read_b32
    if (OP == 4) acc += (uint32_t)atomicCAS(&l64[a >> 1], ~0ULL, (unsigned long long)x);  // ds_cmpst_rtn_b64
    if (OP == 5) acc += (uint32_t)l64[a >> 1];                 // ds_read_b64
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc + lds[threadIdx.x];
}
template <int OP> void run(const char *name, uint32_t mask) {
  uint32_t *o; hipMalloc(&o, 256 * 1024 * 4);
  int iters = 2048;
  hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<256, 1024, 131072>>>(o, 16, mask);
  hipEventRecord(e0); k<OP><<<256, 1024, 131072>>>(o, iters, mask); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double ops = 256.0 * 1024 * iters;
  printf("%-22s mask %6x  %7.3f ms  %7.2f G lane-ops/s/CU  = %.2f lanes/clk/CU @2.4GHz\n", name, mask, ms, ops / ms / 1e6 / 256, ops / 256 / (ms * 1e-3 * 2.4e9));
  hipFree(o);
}
int main() {
  for (uint32_t mask : {32767u, 4095u, 0u}) {
    run<0>("ds_add_u32", mask); run<1>("ds_add_rtn_u32", mask); run<2>("ds_write_b32", mask); run<3>("ds_read_b32", mask);
    run<4>("ds_cmpst_rtn_b64", mask); run<5>("ds_read_b64", mask);
  }
  return 0;
}
