// micro-benchmark: what does the split kernels' store pattern cost, and what would a better one buy?  Every workgroup
// (one per CU, 1024 threads) appends runs round-robin to 1024 streams of its own, like a level-1 writer / a level-2
// bucket: per round every stream gets one run.  Variants: bytes per lane (4, 8, 16), bytes per run (32, 64, 128, 256),
// runs aligned to their own size or starting at any multiple of the lane width (what the kernels do today: a run
// starts where the last one ended, and run lengths vary).  Reports TB/s of payload.
//   hipcc -O2 --offload-arch=gfx950 -o ubench_stores scripts/ubench_stores.hip && ./ubench_stores
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

// LB: bytes per lane; RUN: nominal bytes per run; jitter: run lengths vary by +-half (unaligned starts) or not
template <int LB>
__global__ __launch_bounds__(1024) void k(uint8_t *arena, size_t stream_bytes, uint32_t rounds, uint32_t run, uint32_t jitter, uint32_t nstreams) {
  __shared__ uint32_t cur[1024], rbase[1024], rlen[1024];  // bytes written to each stream; this round's start and length
  const uint32_t t = threadIdx.x;
  cur[t] = 0;
  __syncthreads();
  uint8_t *mine = arena + (size_t)blockIdx.x * 1024 * stream_bytes;
  const uint32_t lanes_per_run = run / LB;              // lanes that share a run (nominal)
  const uint32_t runs_per_pass = 1024 / lanes_per_run;  // runs the workgroup writes per store instruction
  uint32_t x = t * 2654435761u + blockIdx.x;
  for (uint32_t r = 0; r < rounds; r++) {
    {  // thread s: this round's run of stream s -- nominal length, or nominal -/+ up to half
      uint32_t len = lanes_per_run;
      if (jitter) len = lanes_per_run / 2 + ((t * 0x9E3779B9u + r * 0x85EBCA6Bu) >> 16) % (lanes_per_run + 1);
      // nstreams < 1024: the 1024 runs of a round go to only nstreams streams (run i to stream i % nstreams, one behind
      // the other), so that few lines are open at a time and a line is completed soon after it was begun
      const uint32_t st = t % nstreams;
      uint32_t before = 0;
      for (uint32_t i = st; i < t; i += nstreams)
        before += jitter ? lanes_per_run / 2 + ((i * 0x9E3779B9u + r * 0x85EBCA6Bu) >> 16) % (lanes_per_run + 1) : lanes_per_run;
      rbase[t] = cur[st] + before * LB;
      rlen[t] = len;
    }
    __syncthreads();
    if (t < nstreams) {
      uint32_t tot = 0;
      for (uint32_t i = t; i < 1024; i += nstreams) tot += rlen[i];
      cur[t] += tot * LB;
    }
    for (uint32_t s0 = 0; s0 < 1024; s0 += runs_per_pass) {
      const uint32_t s = s0 + t / lanes_per_run, l = t % lanes_per_run;
      const uint32_t base = rbase[s], len = rlen[s];
      if (l < len && base + (l + 1) * LB <= stream_bytes * (1024 / nstreams)) {
        uint8_t *p = mine + (size_t)(s % nstreams) * (stream_bytes * (1024 / nstreams)) + base + l * LB;
        if (LB == 4) *(uint32_t *)p = x;
        if (LB == 8) *(uint64_t *)p = x;
        if (LB == 16) *(uint4 *)p = make_uint4(x, x, x, x);
      }
    }
    __syncthreads();
  }
}

template <int LB>
void run(uint8_t *arena, size_t arena_bytes, uint32_t run_bytes, uint32_t jitter, uint32_t nstreams = 1024) {
  const uint32_t rounds = 128;
  const size_t stream_bytes = (size_t)rounds * run_bytes * 3 / 2 + 256;
  if ((size_t)256 * 1024 * stream_bytes > arena_bytes) { printf("arena too small\n"); return; }
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  k<LB><<<256, 1024>>>(arena, stream_bytes, 8, run_bytes, jitter, nstreams);
  (void)hipEventRecord(e0);
  k<LB><<<256, 1024>>>(arena, stream_bytes, rounds, run_bytes, jitter, nstreams);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double payload = 256.0 * 1024 * rounds * run_bytes;
  printf("%2d B/lane  %3u-byte runs  %4u streams per workgroup  %-28s %7.3f ms  %5.2f TB/s\n", LB, run_bytes, nstreams,
         jitter ? "varying length (unaligned)" : "fixed length (aligned)", ms,
         payload / (ms * 1e-3) / 1e12);
}

int main() {
  const size_t arena_bytes = (size_t)24 << 30;
  uint8_t *arena;
  if (hipMalloc(&arena, arena_bytes) != hipSuccess) return 1;
  (void)hipMemset(arena, 0, arena_bytes);
  for (uint32_t jitter : {0u, 1u}) {
    for (uint32_t rb : {32u, 64u, 128u, 256u}) {
      run<4>(arena, arena_bytes, rb, jitter);
      run<8>(arena, arena_bytes, rb, jitter);
      if (rb >= 64) run<16>(arena, arena_bytes, rb, jitter);
    }
  }
  // few open streams: does the L2 put the pieces of a line together before memory sees them?
  for (uint32_t ns : {512u, 256u, 128u, 32u}) {
    run<4>(arena, arena_bytes, 64, 1, ns);
    run<8>(arena, arena_bytes, 64, 1, ns);
  }
  return 0;
}
