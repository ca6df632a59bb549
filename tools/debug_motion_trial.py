#!/usr/bin/env python3
"""Re-run one trial of tests/test_gpu_fuzz.py::test_motion_fuzz and print the entries of the fused observation that
differ most from the checker's, with the joint they belong to and that joint's rotation angle.
usage (repository root): python tools/debug_motion_trial.py <IGX_FUZZ_SEED> <IGX_FUZZ_TRIALS> <trial>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ["IGX_FUZZ_SEED"], os.environ["IGX_FUZZ_TRIALS"] = sys.argv[1], sys.argv[2]
from isaacgym_amd import ops  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402

want_trial = int(sys.argv[3])
g = H.load_golden("motion")
tabs_np = H.motion_tabs_from_golden(g)
tabs_dev = {k: torch.from_numpy(v).to(F.DEV) for k, v in tabs_np.items()}
tables = ops.motion_tables(tabs_dev)
rng = np.random.default_rng(43 + F.SEED)
for trial, n in enumerate(F._sizes(rng, 16, 10 * F.TRIALS)):
    ids = rng.integers(0, 2, n).astype(np.int64)
    times = rng.uniform(-0.04, 1.2, n) * tabs_np["motion_length"][ids]
    K, dt, lro = int(rng.integers(1, 5)), float(rng.uniform(0.005, 0.08)), bool(rng.integers(2))
    if trial != want_trial:
        continue
    out = torch.zeros(n, K * 105, device=F.DEV)
    ops.motion_amp_obs_steps(tables, torch.from_numpy(ids).to(F.DEV), torch.from_numpy(times).to(F.DEV), K, dt, lro, out)
    want = oracle.motion_amp_obs_steps(tabs_np, ids, times, K, dt, lro).reshape(-1, 105)
    got = out.cpu().numpy().reshape(-1, 105)
    ids_t = np.tile(np.expand_dims(ids, -1), [1, K]).flatten()
    times_t = (np.expand_dims(times, -1) + (-dt * np.arange(0, K))).flatten()
    dp = torch.zeros(n * K, 28, device=F.DEV)
    ops.motion_state(tables, torch.from_numpy(ids_t).to(F.DEV), torch.from_numpy(times_t).to(F.DEV), dict(dof_pos=dp))
    dpo = oracle.motion_state(tabs_np, ids_t, times_t)["dof_pos"]
    dpg = dp.cpu().numpy()
    diff = np.abs(got - want)
    diff[:, 65:93] = 0
    print("n", n, "K", K, "dt", dt, "local_root_obs", lro)
    obs_cols, pos_cols = H.amp_joint_columns("obs"), H.amp_joint_columns("dof_pos")
    for flat in np.argsort(-diff, axis=None)[:6]:
        r, c = divmod(int(flat), 105)
        joint = next((j for j, cols in enumerate(obs_cols) if c in list(cols)), None)
        line = "row %d (sample %d step %d) col %d err %.3e got %.8f want %.8f" % (r, r // K, r % K, c, diff[r, c], got[r, c], want[r, c])
        if joint is not None:
            pc = list(pos_cols[joint])
            line += "\n   joint %d dof_pos gpu %s\n           checker %s  |angle| %.3e  motion %d time %.6f" % (
                joint, dpg[r, pc].tolist(), dpo[r, pc].tolist(), float(np.linalg.norm(dpo[r, pc])), ids_t[r], times_t[r])
        print(line)
    break
