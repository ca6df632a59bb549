#!/usr/bin/env python3
"""Host profile of HumanoidAMP.reset_flagged (or, with the argument `done`, reset_done) at 4 096 environments.
Usage (repository root): python tools/amp_reset_profile.py [done]"""
import cProfile
import io
import pstats
import sys
import tempfile
import time

import torch

sys.path.insert(0, ".")
from isaacgym_amd import cfg as C  # noqa: E402
from isaacgym_amd import tasks  # noqa: E402
from isaacgym_amd.sim import SyntheticSim  # noqa: E402
from isaacgym_amd.tasks.amp.utils_amp.motion_lib import MotionLib  # noqa: E402
from tests import helpers as H  # noqa: E402

DEV = "cuda:0"
n = 4096
with tempfile.TemporaryDirectory() as d:
    lib = MotionLib(H.write_golden_motions(H.load_golden("motion"), d), 28, [5, 8, 11, 14], DEV)
task = tasks.HumanoidAMP(C.task_config("HumanoidAMP", n), DEV, DEV,
                         sim=SyntheticSim("HumanoidAMP", n, DEV, seed=7, num_amp_obs_steps=2), motion_lib=lib)
mask = (torch.rand(n, device=DEV) < 0.003).to(torch.int64)
for _ in range(20):
    task.reset_buf.copy_(mask)
    task.reset_flagged()
torch.cuda.synchronize()
k = 2000
t0 = time.perf_counter()
for _ in range(k):
    task.reset_buf.copy_(mask)
    task.reset_flagged()
host = (time.perf_counter() - t0) / k * 1e6
torch.cuda.synchronize()
print("reset_flagged + flag copy: host %.1f us, wall %.1f us per call" % (host, (time.perf_counter() - t0) / k * 1e6))
which = sys.argv[1] if len(sys.argv) > 1 else "flagged"
pr = cProfile.Profile()
pr.enable()
for _ in range(k):
    if which == "flagged":
        task.reset_flagged()
    else:  # the reference's sequence: nonzero() on the host, numpy draws
        task.reset_buf.copy_(mask)
        task.reset_done()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(26)
print(s.getvalue())
