#!/usr/bin/env python3
"""Does the Humanoid step's time depend on where its buffers lie?  Re-allocates the four rotating input sets several
times in one process (a pad of a different size in front each time, so that every array's base address moves) and
times 100 launches per placement.  Usage (repository root): python tools/humanoid_spread.py [ant]"""
import sys

import torch

sys.path.insert(0, ".")
from isaacgym_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"
N = 1 << 20
ant = len(sys.argv) > 1 and sys.argv[1] == "ant"
spec, step = (synthetic.ANT, ops.ant_step) if ant else (synthetic.HUMANOID, ops.humanoid_step)


def timed(fn, k=100, warm=12):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(k):
        fn(warm + i)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


for trial, pad_kb in enumerate((0, 4, 68, 1024, 2048 + 36, 0, 300, 8192 + 4)):
    torch.cuda.empty_cache()
    pad = torch.empty(pad_kb * 1024 + 16, dtype=torch.uint8, device=DEV)
    p = synthetic.loco_params(spec, 1, 0)
    sets = [synthetic.loco_state(spec, N, s, DEV) for s in range(4)]
    for s in sets[1:]:
        for k in ("potentials", "prev_potentials", "reset", "progress", "randomize", "obs", "up_vec", "heading_vec", "rew"):
            s[k] = sets[0][k]
    caches = [dict() for _ in sets]

    def fn(i):
        p.rng_offset = i
        step(p, sets[i % 4], N, caches[i % 4])
    us = [timed(fn) for _ in range(3)]
    bases = {k: (sets[0][k].data_ptr() >> 12) & 0x1ff for k in ("root_states", "dof_state", "obs", "actions", "sensors")}
    print("placement %d pad %5d KB: %s us   4K-page index mod 512 of a few bases: %s" %
          (trial, pad_kb, " ".join("%.1f" % u for u in us), bases), flush=True)
    del sets, caches, pad
