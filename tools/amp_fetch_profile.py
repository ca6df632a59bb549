#!/usr/bin/env python3
"""HumanoidAMP.fetch_amp_obs_demo (host numpy sampling, as the reference) and fetch_amp_obs_demo_device at 512
trajectories: wall time per fetch and a host profile.  Usage (repository root): python tools/amp_fetch_profile.py"""
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
for name, fn in (("fetch_amp_obs_demo (numpy draws)", lambda: task.fetch_amp_obs_demo(512)),
                 ("fetch_amp_obs_demo_device", lambda: task.fetch_amp_obs_demo_device(512))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    k = 2000
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    host = (time.perf_counter() - t0) / k * 1e6
    torch.cuda.synchronize()
    print("%-36s host %.1f us, wall %.1f us per fetch" % (name, host, (time.perf_counter() - t0) / k * 1e6))
pr = cProfile.Profile()
pr.enable()
for _ in range(1000):
    task.fetch_amp_obs_demo(512)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14)
print(s.getvalue())
