#!/usr/bin/env python3
"""Wall time of a whole VecTask.step (host code + launches) next to the device time of its kernels.
Usage: PYTHONPATH=. python tools/step_overhead.py"""
import time

import torch

from isaacgym_amd import tasks
from isaacgym_amd.graph import GraphedStep

DEV = "cuda:0"
_motion_lib = None


def make(name, n):
    """tasks.make, with a synthetic clip for HumanoidAMP (the reference's motion files are not in the tree)."""
    global _motion_lib
    if name != "HumanoidAMP":
        return tasks.make(name, n, DEV)
    import os
    import tempfile
    from isaacgym_amd import cfg as C
    from isaacgym_amd.sim import SyntheticSim
    from isaacgym_amd.tasks.amp.utils_amp.motion_lib import MotionLib
    if _motion_lib is None:
        import sys
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
        from tests import helpers as H
        path = H.write_golden_motions(H.load_golden("motion"), tempfile.mkdtemp())
        _motion_lib = MotionLib(path, 28, [5, 8, 11, 14], DEV)
    cfg = C.task_config("HumanoidAMP", n)
    sim = SyntheticSim("HumanoidAMP", n, DEV, seed=7, num_amp_obs_steps=cfg["env"]["numAMPObsSteps"])
    return tasks.HumanoidAMP(cfg, DEV, DEV, sim=sim, motion_lib=_motion_lib)


for name, n, acts in (("Cartpole", 4096, 1), ("Ant", 4096, 8), ("Ant", 1 << 20, 8), ("Humanoid", 4096, 21),
                      ("AllegroHand", 4096, 16), ("ShadowHand", 4096, 20), ("HumanoidAMP", 4096, 28),
                      ("AntImitate", 4096, 8), ("HumanoidImitate", 4096, 63)):
    t = make(name, n)
    a = torch.zeros(n, acts, device=DEV)
    for _ in range(20):
        t.step(a)
    torch.cuda.synchronize()
    k = 300
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(k):
        t.step(a)
    e1.record()
    host = (time.perf_counter() - t0) / k * 1e6
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / k * 1e6
    line = ("%-12s n=%8d  host %.1f us/step  wall %.1f us/step  device span %.1f us/step" %
            (name, n, host, wall, e0.elapsed_time(e1) / k * 1e3))
    # the stand-in simulator copies a fresh state set into the shared tensors every simulate(); with an engine
    # that writes its tensors in place (here: a simulate() that does nothing) the step is this package's code alone
    t2 = make(name, n)
    t2.sim.simulate = lambda: None
    for _ in range(20):
        t2.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        t2.step(a)
    torch.cuda.synchronize()
    line += "  | in-place sim: wall %.1f us/step" % ((time.perf_counter() - t0) / k * 1e6)
    # the same step recorded once into a HIP graph and replayed
    g = GraphedStep(make(name, n))
    for _ in range(20):
        g.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        g.step(a)
    torch.cuda.synchronize()
    print(line + "  | graph replay wall %.1f us/step" % ((time.perf_counter() - t0) / k * 1e6), flush=True)
