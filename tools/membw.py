#!/usr/bin/env python3
"""Measured HBM ceilings of this box: device-to-device copy and fill, with
torch's own kernels, on buffers far larger than the 256 MiB Infinity Cache."""
import json
import torch

def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3

def main():
    n = 1 << 28  # 1 GiB of fp32 per buffer
    srcs = [torch.empty(n, device="cuda").normal_() for _ in range(2)]
    dsts = [torch.empty(n, device="cuda") for _ in range(2)]
    i = [0]
    def copy():
        dsts[i[0] & 1].copy_(srcs[i[0] & 1]); i[0] += 1
    def fill():
        dsts[i[0] & 1].fill_(1.0); i[0] += 1
    def read():
        srcs[i[0] & 1].sum(); i[0] += 1
    t = timeit(copy)
    out = {"copy_GBps": 2 * 4 * n / t / 1e9}
    t = timeit(fill)
    out["fill_GBps"] = 4 * n / t / 1e9
    t = timeit(read)
    out["read_sum_GBps"] = 4 * n / t / 1e9
    print(json.dumps(out))

if __name__ == "__main__":
    main()
