"""CPU checker (oracle/igx_oracle.c) against the golden vectors produced by the
reference's own task classes (oracle/gen_golden.py).  Runs without a GPU."""
import numpy as np
import pytest

from oracle import oracle
from tests import helpers as H


import pytest


@pytest.mark.parametrize("name", ["cartpole", "cartpole_sample", "cartpole_nonfinite", "cartpole_extreme"])
def test_cartpole_oracle_matches_reference(name):
    g = H.load_golden(name)
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = H.cartpole_params(m)
    reset = g["reset0"].copy()
    progress = g["progress0"].copy()
    for t in range(T):
        ds = g["in_dof_state"][t].copy()
        obs = np.zeros((N, 4), np.float32)
        rew = np.zeros(N, np.float32)
        oracle.cartpole_step(p, dict(dof_state=ds, rand=np.ascontiguousarray(g["rand"][t]), obs=obs,
                                     rew=rew, reset=reset, progress=progress), N)
        assert np.array_equal(reset, g["reset"][t])
        assert np.array_equal(progress, g["progress"][t])
        # elementwise IEEE ops only: bit-exact
        assert H.bit_equal_fraction(ds, g["out_dof_state"][t]) == 1.0
        assert H.bit_equal_fraction(obs, g["obs"][t]) == 1.0
        assert H.bit_equal_fraction(rew, g["rew"][t]) == 1.0


def _run_loco(name, step_fn, nobs, humanoid):
    g = H.load_golden(name)
    m = g["meta"]
    N, T, D = m["N"], m["T"], m["num_dof"]
    p = H.loco_params(m, g)
    reset = g["reset0"].copy()
    progress = g["progress0"].copy()
    pot = g["potentials0"].copy()
    randomize = np.zeros(N, np.int64)
    stats = {}
    for t in range(T):
        ds = g["in_dof_state"][t].copy()
        out = dict(obs=np.zeros((N, nobs), np.float32), prev_potentials=np.zeros(N, np.float32),
                   up_vec=np.zeros((N, 3), np.float32), heading_vec=np.zeros((N, 3), np.float32),
                   rew=np.zeros(N, np.float32))
        bufs = dict(root_states=np.ascontiguousarray(g["in_root"][t]), dof_state=ds,
                    sensors=np.ascontiguousarray(g["in_sensors"][t]),
                    actions=np.ascontiguousarray(g["in_actions"][t]),
                    initial_root_states=g["initial_root_states"], initial_dof_pos=g["initial_dof_pos"],
                    rand=np.ascontiguousarray(g["rand"][t]), potentials=pot, reset=reset,
                    progress=progress, randomize=randomize, **out)
        if humanoid:
            bufs["dof_force"] = np.ascontiguousarray(g["in_dof_force"][t])
        else:
            bufs["true_objective"] = np.zeros(N, np.float32)
        step_fn(p, bufs, N)
        # flags: exact
        assert np.array_equal(reset, g["reset"][t]), "reset step %d" % t
        assert np.array_equal(progress, g["progress"][t]), "progress step %d" % t
        assert np.array_equal(randomize, g["randomize"][t])
        # everything not passing through atan2/sin/cos follows the exact
        # arithmetic model: require bit equality
        nonangle = [c for c in range(nobs) if c not in H.LOCO_ANGLE_COLS]
        for key, arr in (("out_dof_state", ds), ("potentials", pot),
                         ("prev_potentials", out["prev_potentials"]), ("up_vec", out["up_vec"]),
                         ("heading_vec", out["heading_vec"]), ("rew", out["rew"])):
            f = H.bit_equal_fraction(arr, g[key][t])
            stats.setdefault(key, []).append(f)
            assert f == 1.0, "%s step %d bit-equal fraction %f" % (key, t, f)
        f = H.bit_equal_fraction(out["obs"][:, nonangle], g["obs"][t][:, nonangle])
        assert f == 1.0, "obs non-angle columns step %d: %f" % (t, f)
        H.assert_angle_close(out["obs"][:, list(H.LOCO_ANGLE_COLS)],
                             g["obs"][t][:, list(H.LOCO_ANGLE_COLS)], what="obs angles step %d" % t,
                             branch_cut=H.loco_branch_cut(humanoid))
        if not humanoid:
            assert H.bit_equal_fraction(bufs["true_objective"], g["true_objective"][t]) == 1.0
    return stats


def test_ant_oracle_matches_reference():
    _run_loco("ant", oracle.ant_step, 60, False)


def test_humanoid_oracle_matches_reference():
    _run_loco("humanoid", oracle.humanoid_step, 108, True)


def test_ant_oracle_matches_reference_with_every_parameter_changed():
    """Recorded with every scale, weight, cost, the termination height, dt, the target
    and the start rotation off their YAML values: pins where each one enters."""
    _run_loco("ant_alt", oracle.ant_step, 60, False)


def test_humanoid_oracle_matches_reference_with_every_parameter_changed():
    _run_loco("humanoid_alt", oracle.humanoid_step, 108, True)


def test_ant_oracle_matches_reference_on_non_finite_state():
    """NaN / +-Inf in root rows, joint state, actions and sensors of some environments (a physics blow-up): flags,
    rewards and every finite or NaN output as the reference's own arithmetic leaves them."""
    _run_loco("ant_nonfinite", oracle.ant_step, 60, False)


def test_humanoid_oracle_matches_reference_on_non_finite_state():
    _run_loco("humanoid_nonfinite", oracle.humanoid_step, 108, True)


@pytest.mark.parametrize("name,step,nobs,humanoid", [("ant_extreme", "ant_step", 60, False),
                                                     ("humanoid_extreme", "humanoid_step", 108, True),
                                                     ("ant_nonunit", "ant_step", 60, False),  # root quaternions of
                                                     ("humanoid_nonunit", "humanoid_step", 108, True)])  # length 0.5 ... 2
def test_loco_oracle_matches_reference_on_extreme_values(name, step, nobs, humanoid):
    """+-3e38, +-1e30, 1e-30, subnormals, signed zeros, NaN and +-Inf scattered over 2 % of the simulator state and
    the actions: overflow to Inf, underflow, subnormal arithmetic and the sign of zero as the reference has them."""
    _run_loco(name, getattr(oracle, step), nobs, humanoid)
