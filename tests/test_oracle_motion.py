"""CPU checker's motion-library and AMP-observation functions against golden
vectors produced by the reference MotionLib / build_amp_observations on two
synthetic motions (tests/golden/motion.npz)."""
import numpy as np

from oracle import oracle
from tests import helpers as H


def test_motion_state_matches_reference():
    g = H.load_golden("motion")
    tabs = H.motion_tabs_from_golden(g)
    got = oracle.motion_state(tabs, g["ids"], g["times"])
    H.check_motion_state(got, g)


def test_amp_observations_match_reference():
    g = H.load_golden("motion")
    root_states = np.concatenate([g["root_pos"], g["root_rot"], g["root_vel"], g["root_ang_vel"]], axis=-1)
    for lro in (0, 1):
        obs = oracle.amp_observations(root_states, g["dof_pos"], g["dof_vel"], g["key_pos"].reshape(-1, 12), lro)
        H.assert_close(obs, g["amp_obs_local%d" % lro], what="amp obs local_root_obs=%d" % lro)


def test_fused_motion_amp_obs_matches_reference():
    g = H.load_golden("motion")
    tabs = H.motion_tabs_from_golden(g)
    n = len(g["ids"])
    out = np.zeros((n, 105), np.float32)
    oracle.motion_amp_obs(tabs, g["ids"], g["times"], 0, out)
    H.assert_close(out, g["amp_obs_local0"], what="fused amp obs")
    # strided / indexed output (history slot 1 of a [n, 2, 105] buffer, rows permuted)
    buf = np.zeros((n, 2, 105), np.float32)
    rows = np.random.default_rng(0).permutation(n).astype(np.int64)
    oracle.motion_amp_obs(tabs, g["ids"], g["times"], 0, buf, out_rows=rows, out_row_stride=210, out_col_offset=105)
    assert np.array_equal(buf[rows, 1], out)
    assert not buf[:, 0].any()
