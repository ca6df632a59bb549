#!/usr/bin/env python3
"""igx_humanoid_amp_reset alone at 2^20 environments: every environment listed, a compact list of 3 000, and a
masked list (-1 holes) with 0.3 % set.  Usage (repository root): python tools/amp_reset_kernel_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from isaacgym_amd import ops, synthetic  # noqa: E402

DEV = "cuda:0"
n = 1 << 20
p = synthetic.amp_params(2)
t = synthetic.amp_state(n, 1, DEV, 2)
f = dict(dtype=torch.float32, device=DEV)
state = dict(root_pos=torch.randn(n, 3, **f), root_rot=torch.nn.functional.normalize(torch.randn(n, 4, **f), dim=-1),
             dof_pos=torch.randn(n, 28, **f), root_vel=torch.randn(n, 3, **f), root_ang_vel=torch.randn(n, 3, **f),
             dof_vel=torch.randn(n, 28, **f))
every = torch.arange(n, device=DEV)
few = torch.randperm(n, device=DEV)[:3000].sort().values.contiguous()
masked = torch.where(torch.rand(n, device=DEV) < 0.003, every, torch.full_like(every, -1))


def timed(ids, st, k=30):
    for _ in range(3):
        ops.humanoid_amp_reset(p, t, ids, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        ops.humanoid_amp_reset(p, t, ids, st)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / k * 1e3


print("all %d listed      %8.1f us" % (n, timed(every, state)))
print("compact list of 3000   %8.1f us" % timed(few, {k: v[:3000].contiguous() for k, v in state.items()}))
print("masked list, 0.3 %% set %8.1f us" % timed(masked, state))
