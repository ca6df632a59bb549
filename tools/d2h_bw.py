#!/usr/bin/env python3
"""Device-to-pinned-host copy rate of the box: one stream, two streams, several copy sizes
(the ceiling the trajectory recorder works against)."""
import time

import torch

dev = "cuda:0"
for mb in (1, 8, 64, 256):
    n = mb << 20
    src = [torch.empty(n, dtype=torch.uint8, device=dev) for _ in range(2)]
    dst = [torch.empty(n, dtype=torch.uint8).pin_memory() for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for k in (1, 2):
        reps = max(4, 512 // mb)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for r in range(reps):
            for s in range(k):
                with torch.cuda.stream(streams[s]):
                    dst[s].copy_(src[s], non_blocking=True)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print("%4d MiB copies, %d stream(s): %.1f GB/s" % (mb, k, reps * k * n / dt / 1e9), flush=True)
