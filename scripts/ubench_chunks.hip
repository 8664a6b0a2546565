// micro-benchmark: streaming a large buffer in scattered 4 KiB / 8 KiB chunks, 1024-thread workgroups, one per CU
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int W>  // bytes per lane per load: 8 or 16
__global__ __launch_bounds__(1024) void k(const uint8_t *buf, uint64_t nchunks, uint32_t chunk_bytes, int rounds, uint64_t *out, int scattered) {
  uint64_t acc = 0;
  const uint32_t per_round_bytes = 1024 * W * 16;  // 16 loads in flight per lane
  uint64_t x = blockIdx.x * 0x9E3779B97F4A7C15ULL + 12345;
  for (int r = 0; r < rounds; r++) {
    uint64_t v[16][W / 8];
#pragma unroll
    for (int j = 0; j < 16; j++) {
      const uint32_t off_in_round = (j * 1024 + threadIdx.x) * W;  // byte offset within this round's 128/256 KiB
      const uint32_t chunk_in_round = off_in_round / chunk_bytes;
      uint64_t cid;
      if (scattered) {
        uint64_t h = (x + chunk_in_round) * 0xD1342543DE82EF95ULL; h ^= h >> 29;
        cid = h % nchunks;
      } else {
        cid = ((uint64_t)blockIdx.x * rounds + r) * (per_round_bytes / chunk_bytes) + chunk_in_round;
        cid %= nchunks;
      }
      const uint8_t *p = buf + cid * chunk_bytes + (off_in_round % chunk_bytes);
      if (W == 8) v[j][0] = *(const uint64_t *)p;
      else { ulonglong2 t = *(const ulonglong2 *)p; v[j][0] = t.x; v[j][W / 8 - 1] = t.y; }
    }
#pragma unroll
    for (int j = 0; j < 16; j++) acc += v[j][0] + 3 * v[j][W / 8 - 1];
    x += 977;
  }
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
template <int W> void run(const uint8_t *buf, uint64_t bytes, uint32_t chunk, int scattered) {
  uint64_t *o; hipMalloc(&o, 256 * 1024 * 8);
  int rounds = 600;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<W><<<256, 1024>>>(buf, bytes / chunk, chunk, 10, o, scattered);
  hipEventRecord(e0); k<W><<<256, 1024>>>(buf, bytes / chunk, chunk, rounds, o, scattered); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double gb = 256.0 * rounds * 1024 * W * 16 / 1e9;
  printf("W=%2d B/lane chunk=%5u %s: %7.2f ms  %6.2f TB/s\n", W, chunk, scattered ? "scattered " : "sequential", ms, gb / ms);
  hipFree(o);
}
int main() {
  uint64_t bytes = 48ULL << 30;
  uint8_t *buf; if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 1, bytes);
  for (int sc : {0, 1}) for (uint32_t chunk : {4096u, 8192u, 65536u}) { run<8>(buf, bytes, chunk, sc); run<16>(buf, bytes, chunk, sc); }
  return 0;
}
