"""Pinned host -> device copy rate by piece size and number of streams (what should the host pipe of the library do?)."""
import time, torch
n = 7_500_000_000
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
for streams in (1, 2, 4):
    ss = [torch.cuda.Stream() for _ in range(streams)]
    for piece in (32 << 20, 128 << 20, 512 << 20, n):
        best = 1e9
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            i = 0
            for at in range(0, n, piece):
                with torch.cuda.stream(ss[i % streams]):
                    d[at:at + piece].copy_(h[at:at + piece], non_blocking=True)
                i += 1
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        print("%d stream(s), pieces of %5d MiB: %.1f ms = %.1f GB/s" % (streams, min(piece, n) >> 20, best * 1e3, n / best / 1e9), flush=True)
