// micro-benchmark: LDS op throughput on gfx950 with random (hashed) or lane-linear addresses, at 1, 2, 4 and 8 waves per
// SIMD (one workgroup of 256*w threads per CU; w = 8: two workgroups of 1024).  The LDS twin of ubench_issue.hip: what an
// LDS atomic / compare-and-swap / read / write costs per lane, and how much of it bank conflicts take.
//   hipcc -O2 --offload-arch=gfx950 -o ubench_lds scripts/ubench_lds.hip && ./ubench_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
enum { L_ADD, L_ADD_RTN, L_CAS32, L_CAS64, L_WRITE32, L_READ32, L_WRITE64, L_READ64, L_READ128, L_COUNT };
static const char *lname[L_COUNT] = {"ds_add_u32", "ds_add_rtn_u32", "ds_cmpst_rtn_b32", "ds_cmpst_rtn_b64", "ds_write_b32",
                                     "ds_read_b32", "ds_write_b64", "ds_read_b64", "ds_read_b128"};
// words of LDS the addresses range over: 16384 words = 64 KiB
constexpr uint32_t WORDS = 16384;
template <int OP>
__global__ __launch_bounds__(1024) void k(uint32_t *out, unsigned long long *stamps, int iters, int linear) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < (int)WORDS; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u, acc = 0;
  // eight random addresses per lane, moved by a constant every trip (two cheap instructions per address: the address
  // arithmetic must not be what is measured)
  uint32_t ra[8];
  for (int u = 0; u < 8; u++) {
    x = x * 1664525u + 1013904223u;
    ra[u] = (x >> 8) & (WORDS - 1u);
  }
  unsigned long long *l64 = (unsigned long long *)lds;
  uint4 *l128 = (uint4 *)lds;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 8; u++) {
      ra[u] = (ra[u] + 0x2F35u) & (WORDS - 1u);
      // linear: consecutive lanes, consecutive elements (conflict-free); the base moves with i so nothing is hoisted
      const uint32_t a = linear ? ((threadIdx.x + (uint32_t)(i * 8 + u) * 64u) & (WORDS - 1u)) : ra[u];
      if (OP == L_ADD) atomicAdd(&lds[a], 1u);
      if (OP == L_ADD_RTN) acc += atomicAdd(&lds[a], 1u);
      if (OP == L_CAS32) acc += atomicCAS(&lds[a], 0xFFFFFFFFu, x);
      if (OP == L_CAS64) acc += (uint32_t)atomicCAS(&l64[a >> 1], ~0ULL, (unsigned long long)x);
      if (OP == L_WRITE32) lds[a] = x;
      if (OP == L_READ32) acc += lds[a];
      if (OP == L_WRITE64) l64[a >> 1] = x;
      if (OP == L_READ64) acc += (uint32_t)l64[a >> 1];
      if (OP == L_READ128) acc += l128[a >> 2].x;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    stamps[2 * w] = t1 - t0;
    stamps[2 * w + 1] = w1 - w0;
  }
}
template <int OP>
void run(int wps, int linear) {
  const int threads = wps >= 8 ? 1024 : 256 * wps, grid = wps >= 8 ? 512 : 256;
  const size_t ldsb = wps >= 8 ? 70 * 1024 : 100 * 1024;
  const int iters = 512;
  uint32_t *o;
  unsigned long long *st;
  const size_t nw = (size_t)grid * threads / 64;
  (void)hipMalloc(&o, (size_t)grid * threads * 4);
  (void)hipMalloc(&st, nw * 16);
  (void)hipFuncSetAttribute((const void *)k<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  k<OP><<<grid, threads, ldsb>>>(o, st, 16, linear);
  k<OP><<<grid, threads, ldsb>>>(o, st, iters, linear);
  (void)hipDeviceSynchronize();
  std::vector<unsigned long long> h(2 * nw);
  (void)hipMemcpy(h.data(), st, nw * 16, hipMemcpyDeviceToHost);
  std::vector<double> cyc(nw);
  for (size_t i = 0; i < nw; i++) cyc[i] = (double)h[2 * i];
  std::sort(cyc.begin(), cyc.end());
  const double med = cyc[nw / 2];
  // the CU executed (waves per CU) * iters * 8 wave-instructions during the median wave's loop
  const double wpc = (double)(wps >= 8 ? 32 : 4 * wps);
  printf("%-18s %-7s waves/SIMD %d  %6.2f cycles per wave-instr per CU = %5.1f lanes/clk/CU\n", lname[OP], linear ? "linear" : "random", wps,
         med / (wpc * iters * 8), 64.0 * wpc * iters * 8 / med);
  (void)hipFree(o);
  (void)hipFree(st);
}
template <int OP>
void sweep() {
  for (int lin : {0, 1})
    for (int w : {1, 2, 4, 8}) run<OP>(w, lin);
}
int main() {
  sweep<L_ADD>(); sweep<L_ADD_RTN>(); sweep<L_CAS32>(); sweep<L_CAS64>(); sweep<L_WRITE32>(); sweep<L_READ32>(); sweep<L_WRITE64>(); sweep<L_READ64>();
  sweep<L_READ128>();
  return 0;
}
