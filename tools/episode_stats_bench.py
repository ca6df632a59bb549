#!/usr/bin/env python3
"""The episode-statistics kernels (igx_episode_stats_step) at 2^20 environments for several fractions of finished
episodes per step.  Usage (repository root): python tools/episode_stats_bench.py"""
import sys, torch
sys.path.insert(0, ".")
from isaacgym_amd.episode_stats import EpisodeStats
DEV="cuda:0"; n=1<<20
es=EpisodeStats(n, DEV)
rew=torch.randn(n, device=DEV)
for frac in (0.0, 0.01, 0.04, 0.5, 1.0):
    done=(torch.rand(n, device=DEV) < frac).to(torch.int64)
    for _ in range(5): es.step(rew, done)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): es.step(rew, done)
    e1.record(); torch.cuda.synchronize()
    print("done fraction %.2f: %.1f us per step" % (frac, e0.elapsed_time(e1)/50*1e3))
