"""Two ranks on one GPU over gloo, started by isaacgym_amd.launch: the product's shard entry points
(AllegroHand / ShadowHand shard kernels + queued records, the sharded quantile) against the unsharded calls."""
import json
import os
import sys

import pytest

from isaacgym_amd import launch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_two_ranks_on_one_gpu_equal_the_unsharded_calls():
    rc, out = launch.spawn_ranks([sys.executable, "-u", os.path.join(HERE, "shard_child.py")], 2, timeout=110)
    assert rc == 0, out
    d = json.loads(out.strip().splitlines()[-1])
    assert set(d) == {"allegro", "shadow", "quantile", "window_quantile", "ant_imitate"}


def test_rccl_paths_on_one_rank():
    """The collectives of the package through the nccl (= RCCL) backend on device tensors - one rank is all a one-GPU
    box offers, so the world-of-one shortcuts are switched off: histogram all-reduce and slot all-gather of the
    two-collective quantile, the queued shard records on their side stream."""
    tool = os.path.join(os.path.dirname(HERE), "tools", "nccl_one_rank_check.py")
    env = dict(os.environ, PYTHONPATH=os.path.dirname(HERE) + os.pathsep + os.environ.get("PYTHONPATH", ""))
    rc, out = launch.spawn_ranks([sys.executable, "-u", tool], 1, timeout=110, env=env)
    assert rc == 0, out
    assert "window quantile over nccl ok, collectives per call: 2" in out and "queued shard records over nccl ok" in out
