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
