"""reset_idx on an explicit id list (CPU checker): it must be the same operation as
the flagged reset inside the fused step, which the reference recordings pin."""
import numpy as np
import pytest

from isaacgym_amd import synthetic
from oracle import oracle


@pytest.mark.parametrize("spec,ostep", [(synthetic.ANT, oracle.ant_step), (synthetic.HUMANOID, oracle.humanoid_step)])
@pytest.mark.parametrize("with_rand", [False, True])
def test_loco_reset_then_step_equals_flagged_step(spec, ostep, with_rand):
    n = 3001
    p = synthetic.loco_params(spec, seed=5, rng_offset=2)
    a = {k: v.numpy().copy() for k, v in synthetic.loco_state(spec, n, 4, "cpu", with_rand).items()}
    b = {k: v.copy() for k, v in a.items()}
    ids = np.nonzero(a["reset"])[0]
    assert 0 < len(ids) < n
    oracle.loco_reset(p, a, ids)
    assert not a["reset"].any() and not a["progress"][ids].any()
    assert np.array_equal(a["potentials"][ids], a["prev_potentials"][ids])
    ostep(p, a, n)
    ostep(p, b, n)
    for k in ("obs", "rew", "dof_state", "potentials", "prev_potentials", "up_vec", "heading_vec", "reset"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert np.array_equal(a["progress"][ids], b["progress"][ids] + 1)
    keep = np.ones(n, bool)
    keep[ids] = False
    assert np.array_equal(a["progress"][keep], b["progress"][keep])


@pytest.mark.parametrize("with_rand", [False, True])
def test_cartpole_reset_then_step_equals_flagged_step(with_rand):
    n = 4096
    p = synthetic.cartpole_params(seed=8, rng_offset=1)
    a = {k: v.numpy().copy() for k, v in synthetic.cartpole_state(n, 2, "cpu", with_rand).items()}
    b = {k: v.copy() for k, v in a.items()}
    ids = np.nonzero(a["reset"])[0]
    oracle.cartpole_reset(p, a, ids)
    assert not a["reset"].any()
    oracle.cartpole_step(p, a, n)
    oracle.cartpole_step(p, b, n)
    for k in ("obs", "rew", "dof_state", "reset"):
        assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k
    assert np.array_equal(a["progress"][ids], b["progress"][ids] + 1)
