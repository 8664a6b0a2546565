// How should the level-1 arena be allocated so that level 1's write pattern always runs at its fast rate?  Round 2
// found the same kernel 25 % apart on different hipMalloc allocations of the same size (fixed for the life of the
// allocation; the first process on a fresh box draws slow ones most) and papered over it by probing and re-allocating.
// This program times level 1's store pattern alone (the library's kc_arena_probe_kernel, copied) on a 54 GB arena obtained
//   malloc       : hipMalloc
//   scrub        : hipMalloc, written end to end, freed, hipMalloc again (what round 2's library did)
//   vmm <chunk>  : one virtual range (hipMemAddressReserve, aligned to the chunk size) backed by physical handles of
//                  <chunk> MiB each (hipMemCreate / hipMemMap / hipMemSetAccess)
// and prints the allocation's address, the driver's granularities and the rate.  Run it as the FIRST process on a box
// and again: scripts/arena_alloc_probe.sh does both.
//   hipcc -O2 --offload-arch=gfx950 -o arena_alloc_probe scripts/arena_alloc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x)                                                                                  \
  do {                                                                                         \
    hipError_t e_ = (x);                                                                       \
    if (e_ != hipSuccess) {                                                                    \
      printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);                \
      return 1;                                                                                \
    }                                                                                          \
  } while (0)

// every workgroup appends 64-byte runs round-robin to a window of 1024 open chunks of its own part of the arena; the
// window jumps through the whole part (32 places, eight rounds at each)
__global__ __launch_bounds__(1024) void probe(uint64_t *arena, size_t words_per_wg, uint32_t rounds, uint32_t window, uint32_t stagger) {
  uint64_t *mine = arena + (size_t)blockIdx.x * words_per_wg;
  const uint32_t t = threadIdx.x, b = t >> 3, w = t & 7u;
  const size_t chunks = words_per_wg / 512u;
  const size_t hop = chunks > 1024u ? (chunks - 1024u) / 31u : 0u;
  for (uint32_t r = 0; r < rounds; r++) {
    // stagger: the workgroups' windows do not sit at the same offset of their parts (as they do when all writers fill
    // their arenas at the same pace), but up to 16 MiB apart
    const size_t first = (size_t)((r >> 3) & 31u) * hop + (stagger ? ((size_t)blockIdx.x * 7919u) % 4096u : 0u);
#pragma unroll
    for (uint32_t j = 0; j < 8; j++) {
      const size_t chunk = first + (b + 128u * j) % window;  // window < 1024: several runs of a round share a chunk (other words of it)
      const size_t off = (((size_t)(r & 7u) + 8u * ((b + 128u * j) / window)) * 8u + w) % 512u;
      if (stagger == 2) {
        // block-cyclic: the writers' arenas interleaved in blocks of 512 chunks (2 MiB), so that what all writers fill at
        // one time is one dense stretch of memory whatever the physical layout behind it
        const size_t gchunk = (((chunk >> 9) * gridDim.x + blockIdx.x) << 9) | (chunk & 511u);
        if ((chunk | 511u) < chunks) arena[gchunk * 512u + off] = (uint64_t)r;
      } else {
        const size_t at = chunk * 512u + off;
        if (at < words_per_wg) mine[at] = (uint64_t)r;
      }
    }
  }
}

static double rate(uint64_t *p, size_t bytes, uint32_t window = 1024, uint32_t stagger = 0) {
  const uint32_t G = 256, rounds = 256;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 3; rep++) {
    (void)hipEventRecord(e0);
    probe<<<G, 1024>>>(p, bytes / 8 / G, rounds, window, stagger);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return (double)G * rounds * 65536.0 / (best * 1e-3) / 1e12;
}

int main(int argc, char **argv) {
  const char *mode = argc > 1 ? argv[1] : "malloc";
  const size_t bytes = (size_t)54 << 30;
  CK(hipSetDevice(0));
  uint64_t *p = nullptr;
  if (!strcmp(mode, "contig")) {  // physically contiguous, if the driver can find that much in one piece
    CK(hipExtMallocWithFlags((void **)&p, bytes, hipDeviceMallocContiguous));
  } else if (!strcmp(mode, "malloc") || !strcmp(mode, "scrub")) {
    CK(hipMalloc((void **)&p, bytes));
    if (!strcmp(mode, "scrub")) {
      CK(hipMemset(p, 0, bytes));
      CK(hipDeviceSynchronize());
      CK(hipFree(p));
      CK(hipMalloc((void **)&p, bytes));
    }
  } else {
    const size_t chunk = (size_t)(argc > 2 ? atoi(argv[2]) : 1024) << 20;
    hipMemAllocationProp prop;
    memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
    CK(hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    const size_t n = (bytes + chunk - 1) / chunk;
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, n * chunk, chunk, nullptr, 0));
    std::vector<hipMemGenericAllocationHandle_t> h(n);
    for (size_t i = 0; i < n; i++) {
      CK(hipMemCreate(&h[i], chunk, &prop, 0));
      CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
    }
    hipMemAccessDesc acc;
    memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, n * chunk, &acc, 1));
    p = (uint64_t *)va;
    if (argc > 3) {
      // the same physical pieces mapped in another order: does the rate depend on which piece backs which writer's part?
      // (no memory is held beyond the arena itself: unmap, shuffle the handles, map again)
      uint64_t seed = 12345;
      for (int t = 0; t < atoi(argv[3]); t++) {
        printf("   order %d: %.2f TB/s\n", t, rate(p, bytes));
        CK(hipMemUnmap(va, n * chunk));
        for (size_t i = n - 1; i > 0; i--) {
          seed = seed * 6364136223846793005ULL + 1442695040888963407ULL;
          std::swap(h[i], h[(seed >> 33) % (i + 1)]);
        }
        for (size_t i = 0; i < n; i++) CK(hipMemMap((char *)va + i * chunk, chunk, 0, h[i], 0));
        CK(hipMemSetAccess(va, n * chunk, &acc, 1));
      }
    }
  }
  if (!strcmp(mode, "vmm") || !strcmp(mode, "malloc")) {
    // is slowness a property of WHERE in the allocation?  every GiB of it by itself (all 256 workgroups inside it)
    const size_t piece = (size_t)1 << 30;
    std::vector<double> rr;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (size_t i = 0; i + piece <= bytes; i += piece) {
      float best = 1e30f;
      for (int rep = 0; rep < 2; rep++) {
        (void)hipEventRecord(e0);
        probe<<<256, 1024>>>((uint64_t *)((char *)p + i), piece / 8 / 256, 64, 1024, 0);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      rr.push_back(256.0 * 64 * 65536.0 / (best * 1e-3) / 1e12);
    }
    printf("   per GiB:");
    for (double x : rr) printf(" %.1f", x);
    printf("\n");
  }
  const double r1 = rate(p, bytes);
  CK(hipMemset(p, 0, bytes));  // does having been written change it?
  CK(hipDeviceSynchronize());
  const double r2 = rate(p, bytes);
  printf("   open chunks per workgroup 1024 / 256 / 64 / 16: %.2f / %.2f / %.2f / %.2f TB/s\n", r2, rate(p, bytes, 256), rate(p, bytes, 64), rate(p, bytes, 16));
  printf("   staggered windows, open chunks 1024 / 64: %.2f / %.2f TB/s\n", rate(p, bytes, 1024, 1), rate(p, bytes, 64, 1));
  printf("   writers interleaved in 2 MiB blocks, open chunks 1024 / 64: %.2f / %.2f TB/s\n", rate(p, bytes, 1024, 2), rate(p, bytes, 64, 2));
  printf("%-6s %s  %p  (address mod 1 GiB: %zu MiB)  %.2f TB/s fresh, %.2f TB/s after being written once\n", mode, argc > 2 ? argv[2] : "", (void *)p,
         ((size_t)p & (((size_t)1 << 30) - 1)) >> 20, r1, r2);
  return 0;
}
