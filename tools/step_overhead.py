#!/usr/bin/env python3
"""Wall time of a whole VecTask.step (host code + launches) next to the device time of its kernels.
Usage: PYTHONPATH=. python tools/step_overhead.py"""
import time

import torch

from isaacgym_amd import tasks
from isaacgym_amd.graph import GraphedStep

DEV = "cuda:0"
for name, n, acts in (("Cartpole", 4096, 1), ("Ant", 4096, 8), ("Ant", 1 << 20, 8), ("Humanoid", 4096, 21),
                      ("AllegroHand", 4096, 16), ("ShadowHand", 4096, 20)):
    t = tasks.make(name, n, DEV)
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
    t2 = tasks.make(name, n, DEV)
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
    g = GraphedStep(tasks.make(name, n, DEV))
    for _ in range(20):
        g.step(a)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        g.step(a)
    torch.cuda.synchronize()
    print(line + "  | graph replay wall %.1f us/step" % ((time.perf_counter() - t0) / k * 1e6), flush=True)
