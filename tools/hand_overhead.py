#!/usr/bin/env python3
"""Host cost of a whole VecTask.step at 4 096 environments with an in-place simulator, with and without the per-call
re-validation of the cached argument structs (where every held tensor points), and a cProfile of the AllegroHand step.
Usage (repository root): python tools/hand_overhead.py"""
import time, sys, cProfile, pstats, io
import torch
sys.path.insert(0, ".")
from isaacgym_amd import tasks, _lib
DEV="cuda:0"
def run(name, acts, patch):
    t = tasks.make(name, 4096, DEV)
    t.sim.simulate = lambda: None
    a = torch.zeros(4096, acts, device=DEV)
    if patch:
        class EQ:
            def __eq__(self, o): return True
        eq = EQ()
        _lib._pointers_saved = _lib._pointers
        _lib._pointers = lambda h: eq
    for _ in range(50): t.step(a)
    torch.cuda.synchronize()
    k=2000
    t0=time.perf_counter()
    for _ in range(k): t.step(a)
    host=(time.perf_counter()-t0)/k*1e6
    torch.cuda.synchronize()
    wall=(time.perf_counter()-t0)/k*1e6
    if patch: _lib._pointers = _lib._pointers_saved
    return host, wall, t, a
for name, acts in (("AllegroHand",16),("ShadowHand",20),("Ant",8)):
    for patch in (False, True, False, True):
        h,w,t,a = run(name, acts, patch)
        print(name, "no pointer check" if patch else "pointer check   ", "host %.1f wall %.1f us" % (h,w), flush=True)
t = tasks.make("AllegroHand", 4096, DEV); t.sim.simulate = lambda: None
a = torch.zeros(4096,16,device=DEV)
for _ in range(50): t.step(a)
pr=cProfile.Profile(); pr.enable()
for _ in range(2000): t.step(a)
pr.disable(); torch.cuda.synchronize()
s=io.StringIO(); pstats.Stats(pr,stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue())
