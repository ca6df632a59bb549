#!/usr/bin/env python3
"""Does the placement of the buffers in HBM matter?  Times igx_ant_step on state whose tensors
are (a) separate torch allocations and (b) carved out of one arena at staggered offsets.
Usage: PYTHONPATH=. python tools/stagger_test.py [stagger_bytes ...]"""
import sys

import torch

from isaacgym_amd import ops, synthetic

DEV = "cuda:0"
N = 1 << 20


def timed(fn, steps=100, warmup=10):
    for i in range(warmup):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(steps):
        fn(warmup + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps * 1e3


def carve(state, stagger):
    """Copy every tensor of `state` into one arena; tensor k starts at a 2 MiB boundary + k * stagger."""
    total = sum((t.numel() * t.element_size() + (4 << 20)) for t in state.values() if t is not None)
    arena = torch.empty(total, dtype=torch.uint8, device=DEV)
    base = (arena.data_ptr() + (2 << 20) - 1) // (2 << 20) * (2 << 20) - arena.data_ptr()
    out, off = {"_arena": arena}, base
    for k, (name, t) in enumerate(state.items()):
        if t is None:
            out[name] = None
            continue
        nbytes = t.numel() * t.element_size()
        start = off + k * stagger
        view = arena[start:start + nbytes].view(t.dtype).view(t.shape)
        view.copy_(t)
        out[name] = view
        off = (start + nbytes + (2 << 20) - 1) // (2 << 20) * (2 << 20)
    return out


def main():
    staggers = [int(x) for x in sys.argv[1:]] or [0, 256, 4352, 69888]
    p = synthetic.loco_params(synthetic.ANT, 1, 0)
    sets = [synthetic.loco_state(synthetic.ANT, N, s, DEV) for s in range(4)]
    print("separate allocations: %.2f us" % timed(lambda i: ops.ant_step(p, sets[i % 4], N)))
    for st in staggers:
        carved = [carve(s, st) for s in sets]
        arenas = [c.pop("_arena") for c in carved]
        print("arena, stagger %6d B: %.2f us" % (st, timed(lambda i: ops.ant_step(p, carved[i % 4], N))))
        del carved, arenas


if __name__ == "__main__":
    main()
