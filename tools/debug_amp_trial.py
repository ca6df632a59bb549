#!/usr/bin/env python3
"""Re-run one trial of tests/test_gpu_fuzz.py::test_amp_fuzz and print the environments whose observation differs
most from the checker's.  usage (repository root): python tools/debug_amp_trial.py <IGX_FUZZ_SEED> <IGX_FUZZ_TRIALS> <trial>"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
os.environ["IGX_FUZZ_SEED"], os.environ["IGX_FUZZ_TRIALS"] = sys.argv[1], sys.argv[2]
from isaacgym_amd import ops, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402
from tests import test_gpu_fuzz as F  # noqa: E402

want_trial = int(sys.argv[3])
for trial, n, K, p, t in F.amp_trials():
    if trial != want_trial:
        continue
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    root = ref["root_states"].copy()
    ops.humanoid_amp_step(p, t, n)
    torch.cuda.synchronize()
    oracle.humanoid_amp_step(p, ref, n)
    got, want = t["obs"].cpu().numpy(), ref["obs"]
    err = H.amp_obs_errors(got, want)
    print("local_root_obs", p.local_root_obs, "n", n, "K", K)
    for e in np.argsort(-np.nan_to_num(np.nanmax(np.where(np.isnan(err[:, :13]), -1.0, err[:, :13]), axis=1)))[:3]:
        print("env", e, "max err", np.nanmax(err[e, :13]), "\n root", root[e].tolist(), "\n got ", got[e, :13].tolist(),
              "\n want", want[e, :13].tolist())
    break
