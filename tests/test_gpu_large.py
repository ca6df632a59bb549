"""Sizes beyond 2^31 bytes per buffer: byte offsets into the observation buffer no longer
fit 32 bits.  Environments are independent, so the CPU checker runs on windows of the batch
(the first rows, rows straddling the 2 GiB and 4 GiB byte offsets of the observation buffer,
the ragged last tile) with the caller-supplied uniforms of the same rows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from isaacgym_amd import ops, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402

DEV = "cuda:0"


@pytest.mark.parametrize("spec,step,ostep,n", [
    (synthetic.ANT, ops.ant_step, oracle.ant_step, (1 << 24) + 37),        # obs: 4.03 GB
    (synthetic.HUMANOID, ops.humanoid_step, oracle.humanoid_step, (1 << 23) + 5),  # obs: 3.6 GB
])
def test_step_beyond_32_bit_offsets(spec, step, ostep, n):
    t = synthetic.loco_state(spec, n, 3, DEV, with_rand=True)
    t["reset"][::5] = 1
    if not spec["humanoid"]:
        t["true_objective"] = torch.zeros(n, device=DEV)
    before = {k: v.clone() for k, v in t.items()}
    p = synthetic.loco_params(spec, seed=1, rng_offset=1)
    step(p, t, n)
    torch.cuda.synchronize()
    row_bytes = 4 * spec["num_obs"]
    starts = {0, n - 3000}
    for boundary in (1 << 31, 1 << 32):
        if boundary // row_bytes + 1500 < n:
            starts.add(boundary // row_bytes - 1500)
    nonangle = [c for c in range(spec["num_obs"]) if c not in H.LOCO_ANGLE_COLS]
    for s in sorted(starts):
        m = min(3000, n - s)
        ref = {k: v[s:s + m].cpu().numpy().copy() for k, v in before.items()}
        ostep(p, ref, m)
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k][s:s + m].cpu().numpy(), ref[k]), (k, s)
        for k in ("dof_state", "potentials", "prev_potentials", "up_vec", "heading_vec", "rew"):
            assert H.bit_equal_fraction(t[k][s:s + m].cpu().numpy(), ref[k]) == 1.0, (k, s)
        obs = t["obs"][s:s + m].cpu().numpy()
        assert H.bit_equal_fraction(obs[:, nonangle], ref["obs"][:, nonangle]) == 1.0, s
        H.assert_angle_close(obs[:, list(H.LOCO_ANGLE_COLS)], ref["obs"][:, list(H.LOCO_ANGLE_COLS)], what="angles at %d" % s,
                             branch_cut=H.loco_branch_cut(spec["humanoid"]))


def _window(before, n, s, m, rebase=()):
    """Rows of environments [s, s + m) of every per-environment tensor (any number of rows per
    environment); tensors that are not per-environment are copied whole."""
    out = {}
    for k, v in before.items():
        if v is None:
            out[k] = None
            continue
        rows = v.shape[0] // n if v.shape[0] % n == 0 and v.shape[0] >= n else 0
        a = v[s * rows:(s + m) * rows] if rows else v
        a = a.cpu().numpy().copy()
        if k in rebase:
            a -= rebase[k] * s
        out[k] = a
    return out


def _starts(n, row_bytes):
    starts = {0, n - 2000}
    for boundary in (1 << 31, 1 << 32):
        if boundary // row_bytes + 1000 < n:
            starts.add(boundary // row_bytes - 1000)
    return sorted(starts)


def test_amp_step_beyond_32_bit_offsets():
    n, K = (1 << 22) + 9, 2       # amp_obs: 3.5 GB, rigid_body_state: 3.3 GB
    p = synthetic.amp_params(K, False, True)
    t = synthetic.amp_state(n, 5, DEV, K)
    before = {k: v.clone() for k, v in t.items()}
    ops.humanoid_amp_step(p, t, n)
    torch.cuda.synchronize()
    for s in _starts(n, 4 * 105 * K):
        m = min(2000, n - s)
        ref = _window(before, n, s, m)
        oracle.humanoid_amp_step(p, ref, m)
        got = _window(t, n, s, m)
        for k in ("reset", "terminate", "progress", "rew"):
            assert np.array_equal(got[k], ref[k]), (k, s)
        H.assert_amp_obs_close(got["obs"], ref["obs"], what="obs at %d" % s)
        H.assert_amp_obs_close(got["amp_obs"][:, 0], ref["amp_obs"][:, 0], what="amp_obs at %d" % s, quiet=True)
        assert np.array_equal(got["amp_obs"][:, 1:].view(np.uint32), ref["amp_obs"][:, 1:].view(np.uint32)), s


def test_allegro_post_beyond_32_bit_offsets():
    n = (1 << 23) + 11            # obs: 2.95 GB, root_state_tensor (3 actors per env): 1.3 GB
    p = synthetic.allegro_params("full_state", seed=9, rng_offset=2)
    t = synthetic.allegro_state(n, 21, DEV, "full_state", False)
    before = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in t.items()}
    ops.allegro_post_physics(p, t, n)
    torch.cuda.synchronize()
    rebase = dict(object_indices=3, goal_object_indices=3)  # actor rows of the window start at 3 s
    for s in _starts(n, 4 * 88):
        m = min(2000, n - s)
        ref = _window(before, n, s, m, rebase)
        oracle.allegro_post_physics(p, ref, m)
        got = _window(t, n, s, m)
        for k in ("reset", "reset_goal", "progress", "randomize"):
            assert np.array_equal(got[k], ref[k]), (k, s)
        assert H.bit_equal_fraction(got["successes"], ref["successes"]) == 1.0, s
        assert H.bit_equal_fraction(got["obs"], ref["obs"]) == 1.0, s
        H.assert_close(got["rew"], ref["rew"], what="rew at %d" % s)
    # the global average over all 8.4 M environments, against the same sums taken by torch in fp64
    resets, cons0 = t["reset"].double(), float(before["consecutive_successes"][0])
    num = float(resets.sum())
    want = 0.1 * float((t["successes"].double() * resets).sum()) / num + 0.9 * cons0 if num > 0 else cons0
    assert abs(float(t["consecutive_successes"][0]) - want) <= 1e-4 * max(1.0, abs(want))
