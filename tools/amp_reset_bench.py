#!/usr/bin/env python3
"""HumanoidAMP resets between steps: reset_done() (nonzero + numpy sampling on the host, as the reference does)
against reset_flagged() (everything on the device).  Usage: PYTHONPATH=. python tools/amp_reset_bench.py [envs]"""
import sys
import tempfile
import time

import numpy as np
import torch

from isaacgym_amd import cfg as C
from isaacgym_amd import tasks
from isaacgym_amd.sim import SyntheticSim
from isaacgym_amd.tasks.amp.utils_amp.motion_lib import MotionLib
from tests import helpers as H

DEV = "cuda:0"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
with tempfile.TemporaryDirectory() as d:  # the two synthetic motions of the test recordings
    lib = MotionLib(H.write_golden_motions(H.load_golden("motion"), d), 28, [5, 8, 11, 14], DEV)
cfg = C.task_config("HumanoidAMP", n)
task = tasks.HumanoidAMP(cfg, DEV, DEV, sim=SyntheticSim("HumanoidAMP", n, DEV, seed=7, num_amp_obs_steps=2), motion_lib=lib)
a = torch.zeros(n, 28, device=DEV)
mask = (torch.rand(n, device=DEV) < 0.003).to(torch.int64)
for name, fn in (("reset_done (host sampling)", task.reset_done), ("reset_flagged (device)", task.reset_flagged)):
    for _ in range(5):
        task.reset_buf.copy_(mask)
        fn()
        task.step(a)
    torch.cuda.synchronize()
    k = 50
    t0 = time.perf_counter()
    for _ in range(k):
        task.reset_buf.copy_(mask)
        fn()
        task.step(a)
    torch.cuda.synchronize()
    print("%-28s %8.1f us per reset + step (%d envs, 0.3 %% flagged)" % (name, (time.perf_counter() - t0) / k * 1e6, n), flush=True)
