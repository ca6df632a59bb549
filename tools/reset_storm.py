#!/usr/bin/env python3
"""Step-kernel time against the fraction of environments whose reset flag is set: the steady
state has about 1 % per step, the first step of a run (reset_buf starts at ones) and synchronised
time-outs have 100 %.  Usage: PYTHONPATH=. python tools/reset_storm.py [envs]"""
import sys

import torch

from isaacgym_amd import ops, synthetic

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20


def timed(fn, prepare, steps=20):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    tot = 0.0
    for i in range(steps + 3):
        prepare()
        torch.cuda.synchronize()
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            tot += e0.elapsed_time(e1)
    return tot / steps * 1e3


def flags(t, frac, keys=("reset",)):
    g = torch.Generator(device=DEV).manual_seed(1)
    mask = (torch.rand(t[keys[0]].shape[0], generator=g, device=DEV) < frac).to(torch.int64)

    def prepare():
        for k in keys:
            t[k].copy_(mask)
    return prepare


for frac in (0.0, 0.01, 0.25, 1.0):
    row = ["resets %5.1f %%" % (100 * frac)]
    t = synthetic.loco_state(synthetic.ANT, n, 1, DEV)
    p = synthetic.loco_params(synthetic.ANT, 1, 1)
    row.append("ant %.0f us" % timed(lambda: ops.ant_step(p, t, n), flags(t, frac)))
    t = synthetic.loco_state(synthetic.HUMANOID, n, 1, DEV)
    p = synthetic.loco_params(synthetic.HUMANOID, 1, 1)
    row.append("humanoid %.0f us" % timed(lambda: ops.humanoid_step(p, t, n), flags(t, frac)))
    t = synthetic.allegro_state(n, 1, DEV, "full_state", True)
    p = synthetic.allegro_params("full_state", 1, 1)
    row.append("allegro_pre %.0f us" % timed(lambda: ops.allegro_pre_physics(p, t, n), flags(t, frac, ("reset", "reset_goal"))))
    m = n // 2
    t = synthetic.shadow_state(m, 1, DEV, "full_state", True)
    p = synthetic.shadow_params("full_state", 1, 1)
    row.append("shadow_pre(n/2) %.0f us" % timed(lambda: ops.shadow_pre_physics(p, t, m), flags(t, frac, ("reset", "reset_goal"))))
    t = synthetic.him_state(m, 1, DEV)
    p = synthetic.him_params(1, 1)
    row.append("humanoid_imitate(n/2) %.0f us" % timed(lambda: ops.humanoid_imitate_step(p, t, m), flags(t, frac)))
    t = synthetic.cartpole_state(n, 1, DEV)
    p = synthetic.cartpole_params(1, 1)
    row.append("cartpole %.0f us" % timed(lambda: ops.cartpole_step(p, t, n), flags(t, frac)))
    print("  ".join(row), flush=True)
