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


def test_flagged_sampler_follows_numpy_choice_and_uniform_phase():
    """The device-side reset sampling, restated: np.random.choice(m, p=w) is searchsorted(cumsum(w) / total,
    u, side="right"), sample_time is u * length; holes for the environments without the flag."""
    g = H.load_golden("motion")
    tabs = H.motion_tabs_from_golden(g)
    lengths = np.asarray(g["motion_lengths"], np.float64)
    w = np.array([1.0, 3.0])
    cdf = w.cumsum()
    cdf /= cdf[-1]
    n = 20000
    reset = (np.random.default_rng(3).random(n) < 0.6).astype(np.int64)
    env_ids, ids, times, dflt = oracle.motion_sample_flagged(tabs, cdf, reset, True, seed=11, rng_offset=4)
    assert (dflt == -1).all()
    hole = reset == 0
    assert (env_ids[hole] == -1).all() and (ids[hole] == -1).all() and (times[hole] == 0).all()
    flagged = np.nonzero(reset)[0]
    assert np.array_equal(env_ids[flagged], flagged)
    u0 = np.array([oracle.philox_uniform(11, 4, int(e), 0) for e in flagged[:500]], np.float64)
    u1 = np.array([oracle.philox_uniform(11, 4, int(e), 1) for e in flagged[:500]], np.float64)
    assert np.array_equal(ids[flagged[:500]], np.searchsorted(cdf, u0, side="right"))
    assert np.array_equal(times[flagged[:500]], u1 * lengths[ids[flagged[:500]]])
    assert abs((ids[flagged] == 1).mean() - 0.75) < 0.02
    phase = times[flagged] / lengths[ids[flagged]]
    assert 0.0 <= phase.min() and phase.max() < 1.0 and abs(phase.mean() - 0.5) < 0.02
    _, ids0, times0, _ = oracle.motion_sample_flagged(tabs, cdf, reset, False, seed=11, rng_offset=4)
    assert np.array_equal(ids0, ids) and (times0 == 0).all()     # stateInit Start
    _, ids2, _, _ = oracle.motion_sample_flagged(tabs, cdf, reset, True, seed=11, rng_offset=5)
    assert not np.array_equal(ids2, ids)
    # stateInit Hybrid: a flagged environment takes a reference state with probability p, else the default pose
    env_h, ids_h, times_h, dflt_h = oracle.motion_sample_flagged(tabs, cdf, reset, True, seed=11, rng_offset=4, ref_init_prob=0.3)
    ref_h, def_h = env_h >= 0, dflt_h >= 0
    assert not (ref_h & def_h).any() and np.array_equal(ref_h | def_h, reset != 0)
    assert abs(ref_h.sum() / max(1, (reset != 0).sum()) - 0.3) < 0.02
    assert np.array_equal(ids_h[ref_h], ids[ref_h]) and np.array_equal(times_h[ref_h], times[ref_h])   # same draws
    assert (ids_h[~ref_h] == -1).all() and np.array_equal(dflt_h[def_h], np.nonzero(def_h)[0])
    env_d, ids_d, _, dflt_d = oracle.motion_sample_flagged(tabs, cdf, reset, True, seed=11, rng_offset=4, ref_init_prob=0.0)
    assert (env_d == -1).all() and (ids_d == -1).all() and np.array_equal(dflt_d >= 0, reset != 0)  # stateInit Default


def test_holes_are_skipped_by_motion_state_and_reset():
    g = H.load_golden("motion")
    tabs = H.motion_tabs_from_golden(g)
    ids = np.array([0, -1, 1, 7, 1], np.int64)       # -1 and 7 are not motions
    times = np.array([0.3, 0.1, 0.2, 0.0, 0.5])
    out = oracle.motion_state(tabs, ids, times)
    ref = oracle.motion_state(tabs, np.array([0, 1, 1], np.int64), times[[0, 2, 4]])
    for k in ("root_pos", "dof_pos", "root_rot"):
        assert np.array_equal(out[k][[0, 2, 4]], ref[k]) and (out[k][[1, 3]] == 0).all(), k
    obs = np.full((5, 105), 9.0, np.float32)
    oracle.motion_amp_obs(tabs, ids, times, 0, obs)
    assert (obs[[1, 3]] == 9.0).all() and (obs[[0, 2, 4]] != 9.0).any(axis=1).all()


def test_fetch_amp_obs_demo_matches_reference():
    """amp_demo.npz: the reference class's own fetch_amp_obs_demo (numpy-seeded draws, tiling over the history
    steps in fp64, get_motion_state, build_amp_observations) for 2 / 3 / 4 history steps."""
    g = H.load_golden("amp_demo")
    tabs = H.motion_tabs_from_golden(H.load_golden("motion"))
    for c, case in enumerate(g["meta"]["cases"]):
        got = oracle.motion_amp_obs_steps(tabs, g["c%d_ids" % c], g["c%d_times0" % c], case["K"], case["dt"],
                                          case["local_root_obs"])
        H.assert_amp_obs_close(got, g["c%d_demo" % c], what="fetch_amp_obs_demo case %d" % c)
        # the fused tiling is the tiling of the reference's host code, bit for bit
        K = case["K"]
        ids = np.tile(np.expand_dims(g["c%d_ids" % c], -1), [1, K]).flatten()
        times = (np.expand_dims(g["c%d_times0" % c], -1) + (-case["dt"] * np.arange(0, K))).flatten()
        want = np.zeros((len(ids), 105), np.float32)
        oracle.motion_amp_obs(tabs, ids, times, case["local_root_obs"], want)
        assert np.array_equal(got.view(np.uint32), want.reshape(got.shape).view(np.uint32))


def test_device_sampler_for_demo_batches_follows_the_distributions():
    g = H.load_golden("motion")
    tabs = H.motion_tabs_from_golden(g)
    cdf = np.cumsum(g["motion_weights"]) / np.sum(g["motion_weights"])
    n = 200000
    ids, times = oracle.motion_sample(tabs, cdf, n, True, seed=9, rng_offset=3)
    assert ids.min() == 0 and ids.max() == 1
    assert abs((ids == 1).mean() - g["motion_weights"][1]) < 5e-3       # motion chosen by weight
    phase = times / tabs["motion_length"][ids]
    assert 0.0 <= phase.min() and phase.max() < 1.0 and abs(phase.mean() - 0.5) < 5e-3   # uniform phase
    again = oracle.motion_sample(tabs, cdf, n, True, seed=9, rng_offset=3)
    assert np.array_equal(again[0], ids) and np.array_equal(again[1], times)
    other = oracle.motion_sample(tabs, cdf, n, True, seed=9, rng_offset=4)
    assert not np.array_equal(other[1], times)
    # the flagged sampler with every flag set draws the same numbers
    e, fi, ft, _ = oracle.motion_sample_flagged(tabs, cdf, np.ones(n, np.int64), True, 9, 3)
    assert np.array_equal(fi, ids) and np.array_equal(ft, times) and np.array_equal(e, np.arange(n))
    assert not oracle.motion_sample(tabs, cdf, 100, False, seed=9, rng_offset=3)[1].any()   # stateInit Start: time 0
