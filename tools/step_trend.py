#!/usr/bin/env python3
"""How a step kernel's time settles after its buffers are allocated: blocks of 25 launches, one HIP event pair per
block (events between single launches would drain the queue), from the very first launch on.
Usage (repository root): python tools/step_trend.py [ant|humanoid] [blocks]"""
import sys

import torch

sys.path.insert(0, ".")
from isaacgym_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"
N = 1 << 20
task = sys.argv[1] if len(sys.argv) > 1 else "humanoid"
blocks = int(sys.argv[2]) if len(sys.argv) > 2 else 24
spec, step = (synthetic.ANT, ops.ant_step) if task == "ant" else (synthetic.HUMANOID, ops.humanoid_step)
p = synthetic.loco_params(spec, seed=77, rng_offset=0)
sets = [synthetic.loco_state(spec, N, s, DEV) for s in range(4)]
for s in sets[1:]:
    for k in ("potentials", "prev_potentials", "reset", "progress", "randomize", "obs", "up_vec", "heading_vec", "rew"):
        s[k] = sets[0][k]
caches = [dict() for _ in sets]
torch.cuda.synchronize()
B = 25
events = [torch.cuda.Event(enable_timing=True) for _ in range(blocks + 1)]
events[0].record()
for b in range(blocks):
    for i in range(B):
        j = b * B + i
        p.rng_offset = j
        step(p, sets[j % 4], N, caches[j % 4])
    events[b + 1].record()
torch.cuda.synchronize()
print(task, "us per launch, blocks of %d:" % B, " ".join("%.1f" % (events[b].elapsed_time(events[b + 1]) / B * 1e3) for b in range(blocks)))
