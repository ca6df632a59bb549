"""Seeding and small helpers of the reference's utils/utils.py that the step path's callers rely on.

`set_seed` is what makes the host-side sampling of the AMP motion library reproducible: `MotionLib.sample_motions`
/ `sample_time` draw from numpy's global generator, exactly as the reference does (utils/utils.py:87-115,
tasks/amp/utils_amp/motion_lib.py:61-78)."""
import os
import random

import numpy as np
import torch

from .rlgames_utils import flatten_dict  # noqa: F401  (utils/utils.py:69-77 lives next to the observer that uses it)


def resolve_seed(seed, torch_deterministic=False, rank=0):
    """The seed a rank ends up with (utils/utils.py:89-94): -1 asks for 42 + rank when determinism is wanted and
    for a random seed otherwise; any other value is offset by the rank, so that the ranks of a multi-GPU job
    draw different streams."""
    if seed == -1:
        return 42 + rank if torch_deterministic else int(np.random.randint(0, 10000))
    return seed + rank


def set_seed(seed, torch_deterministic=False, rank=0):
    """Seed python, numpy and torch (host and every visible GPU) and return the seed used."""
    seed = resolve_seed(seed, torch_deterministic, rank)
    print("Setting seed: {}".format(seed))
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    # the kernels of this package have no nondeterministic reductions; the switch is passed on for the
    # torch code around them
    torch.use_deterministic_algorithms(bool(torch_deterministic))
    return seed


def set_np_formatting():
    """utils/utils.py:79-84"""
    np.set_printoptions(edgeitems=30, infstr="inf", linewidth=4000, nanstr="nan", precision=2, suppress=False,
                        threshold=10000, formatter=None)
