"""Host profile (cProfile) of task.step at the reference's default size; with --kernels only runs the steps
(for rocprofv3 --kernel-trace --stats).  Usage: PYTHONPATH=. python tools/prof_step.py <Task> [--kernels]"""
import cProfile
import pstats
import sys

import torch

from isaacgym_amd import tasks

name = sys.argv[1] if len(sys.argv) > 1 else "Ant"
n = 4096
t = tasks.make(name, n, "cuda:0")
a = torch.zeros(n, t.num_actions, device="cuda:0")
for _ in range(20):
    t.step(a)
torch.cuda.synchronize()
if "--kernels" in sys.argv:
    for _ in range(200):
        t.step(a)
    torch.cuda.synchronize()
else:
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(500):
        t.step(a)
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
