"""The remainder wrap of get_euler_xyz on the GPU, compared STRICTLY (a yaw / roll of 6.2831855 where the
reference has 1e-9 is an observation error of 2 pi): the reference's own recording at the wrap
(tests/golden/angles_edge.npz) and 2^20 environments of such rotations against the CPU checker."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, synthetic
from isaacgym_amd.utils import torch_jit_utils as tj
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_kernels_on_the_reference_recording_at_the_wrap():
    g = H.load_golden("angles_edge")

    def euler(q):
        return tuple(t.cpu().numpy() for t in tj.get_euler_xyz(_dev(q)))

    seen = H.angles_edge_check(g, euler, ops.ant_step, ops.humanoid_step, to_buf=_dev, to_np=lambda t: t.cpu().numpy())
    print(seen)
    assert seen["euler_roll"] > 100 and seen["euler_yaw"] > 100 and seen["angles_edge ant at 2 pi"] > 200


@pytest.mark.parametrize("humanoid", [False, True])
@pytest.mark.parametrize("tilted", [False, True])
def test_step_at_the_wrap_vs_oracle_large(humanoid, tilted):
    """2^20 environments whose torso rotations sit on the wrap: zero entries 2 pi away from the checker."""
    n = 1 << 20
    spec = synthetic.HUMANOID if humanoid else synthetic.ANT
    step, ostep = (ops.humanoid_step, oracle.humanoid_step) if humanoid else (ops.ant_step, oracle.ant_step)
    p = synthetic.loco_params(spec, seed=3, rng_offset=2)
    if tilted:
        q = np.array([0.1, -0.2, 0.3, 0.9])
        p.set("inv_start_rot", (q / np.linalg.norm(q) * [-1, -1, -1, 1]).tolist())
    t = synthetic.loco_state(spec, n, 17, DEV)
    t["root_states"][:, 3:7] = _dev(H.edge_quats(n, 23 + humanoid))
    t["reset"].zero_()
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    step(p, t, n)
    torch.cuda.synchronize()
    ostep(p, ref, n)
    obs = t["obs"].cpu().numpy()
    ang = list(H.LOCO_ANGLE_COLS)
    nonangle = [c for c in range(spec["num_obs"]) if c not in ang]
    assert H.bit_equal_fraction(obs[:, nonangle], ref["obs"][:, nonangle]) == 1.0
    d = np.abs(obs[:, ang].astype(np.float64) - ref["obs"][:, ang])
    print("humanoid" if humanoid else "ant", "tilted" if tilted else "identity", "entries:", d.size, "at 2 pi in the checker:",
          int((ref["obs"][:, 7:9] > 6.28).sum()), "more than 1 rad apart:", int((d > 1.0).sum()), "max residual: %.3g" % d[d < 1.0].max())
    flips = H.assert_angle_close(obs[:, ang], ref["obs"][:, ang], what="angles", branch_cut=H.loco_branch_cut(humanoid))
    print("+-pi flips behind normalize_angle:", flips)
    for k in ("potentials", "rew", "up_vec", "heading_vec"):
        assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, k


def test_get_euler_xyz_at_the_wrap_vs_oracle_large():
    from tests.test_oracle_math import OracleTJ
    q = H.edge_quats(1 << 20, 5)
    got = tj.get_euler_xyz(_dev(q))
    ref = OracleTJ.get_euler_xyz(q)
    for name, a, b in zip(("roll", "pitch", "yaw"), got, ref):
        H.assert_angle_close(a.cpu().numpy().reshape(-1), np.asarray(b).reshape(-1), what=name)


def test_ant_2_to_24_environments_strict_against_oracle():
    """The whole 16.7 M-environment batch (4 GB of observations), every yaw / roll / angle_to_target entry
    compared strictly with the CPU checker; a quarter of the torso rotations sit on the remainder wrap."""
    n = 1 << 24
    spec = synthetic.ANT
    p = synthetic.loco_params(spec, seed=11, rng_offset=4)
    t = synthetic.loco_state(spec, n, 29, DEV)
    quarter = n // 4
    t["root_states"][:quarter, 3:7] = _dev(H.edge_quats(quarter, 31))
    ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
    ops.ant_step(p, t, n)
    torch.cuda.synchronize()
    oracle.ant_step(p, ref, n)
    ang = list(H.LOCO_ANGLE_COLS)
    got = t["obs"][:, 7:10].cpu().numpy()
    want = ref["obs"][:, 7:10]
    d = np.abs(got.astype(np.float64) - want)
    print("entries:", d.size, "reference values within 3e-3 of 2 pi:", int((want[:, 0:2] > 6.28).sum()), "more than 1 rad apart:",
          int((d > 1.0).sum()), "max residual: %.3g" % d.max())
    H.assert_angle_close(got, want, what="2^24 environments")
    for k in ("reset", "progress"):
        assert np.array_equal(t[k].cpu().numpy(), ref[k]), k
    for k in ("potentials", "rew"):
        assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, k
    nonangle = [c for c in range(60) if c not in ang]
    for s0 in range(0, n, 1 << 22):  # the other 57 columns bit for bit, in slices (host memory)
        a = t["obs"][s0:s0 + (1 << 22)].cpu().numpy()
        assert H.bit_equal_fraction(a[:, nonangle], ref["obs"][s0:s0 + (1 << 22)][:, nonangle]) == 1.0, s0
