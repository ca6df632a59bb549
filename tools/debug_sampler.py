#!/usr/bin/env python3
"""Device-side motion draws against the checker for a few (seed, offset, environment offset) corners: how many entries
differ, and by how much.  usage (repository root): python tools/debug_sampler.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from isaacgym_amd import ops  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402

DEV = "cuda:0"
g = H.load_golden("motion")
tabs_np = H.motion_tabs_from_golden(g)
tables = ops.motion_tables({k: torch.from_numpy(v).to(DEV) for k, v in tabs_np.items()})
cdf = np.cumsum(g["motion_weights"]) / np.sum(g["motion_weights"])
cdf_dev = torch.from_numpy(cdf).to(DEV)
n = 5000
for seed, offset, env_offset in ((77, 5, 0), (77, 5, 11), (1 << 39, 5, 0), (77, 1 << 19, 0), (77, 5, 1 << 31), (77, 5, (1 << 32) - 100),
                                 (77, 5, 1 << 32), (77, 5, (1 << 33) - 1), ((1 << 40) - 1, (1 << 20) - 1, (1 << 33) - 1)):
    for random_time in (True, False):
        mi = torch.empty(n, dtype=torch.int64, device=DEV)
        mt = torch.empty(n, dtype=torch.float64, device=DEV)
        ops.motion_sample(tables, cdf_dev, mi, mt, random_time=random_time, seed=seed, rng_offset=offset, env_offset=env_offset)
        want = oracle.motion_sample(tabs_np, cdf, n, random_time, seed=seed, rng_offset=offset, env_offset=env_offset)
        a, b = mt.cpu().numpy(), want[1]
        bad = np.nonzero(a != b)[0]
        print("seed %d offset %d env_offset %d random_time %d: ids differ %d, times differ %d%s" % (
            seed, offset, env_offset, random_time, int((mi.cpu().numpy() != want[0]).sum()), len(bad),
            "" if not len(bad) else "  first at %d: %.17g vs %.17g" % (bad[0], a[bad[0]], b[bad[0]])))
