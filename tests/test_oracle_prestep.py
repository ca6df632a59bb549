"""Action scaling and observation / action noise: the numpy oracle and the host-side
schedule arithmetic against a recording of the reference (tests/golden/prestep.npz:
the tasks' own pre_physics_step methods and the real VecTask.apply_randomizations + step)."""
import numpy as np
import pytest

from isaacgym_amd import randomization
from oracle import oracle
from tests import helpers as H


@pytest.fixture(scope="module")
def g():
    return H.load_golden("prestep")


def _eq(a, b):
    np.testing.assert_array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))


def test_force_scaling_matches_reference_classes(g):
    for tag in ("ant", "ant_ps037"):
        a, f = oracle.action_stage(g[tag + "_actions"], "force", gain=g[tag + "_gain"], power_scale=float(g[tag + "_power_scale"]))
        _eq(f, g[tag + "_forces"])
        _eq(a, g[tag + "_self_actions"])
    _, f = oracle.action_stage(g["humanoid_actions"], "force", gain=g["humanoid_gain"],
                               power_scale=float(g["humanoid_power_scale"]))
    _eq(f, g["humanoid_forces"])
    _, f = oracle.action_stage(g["amp_actions"], "force", gain=g["amp_efforts"], power_scale=float(g["amp_power_scale"]))
    _eq(f, g["amp_forces"])


def test_pd_targets_and_cartpole(g):
    _, t = oracle.action_stage(g["amp_actions"], "pd_target", gain=g["amp_scale"], offset=g["amp_offset"])
    _eq(t, g["amp_pd_targets"])
    _, f = oracle.action_stage(g["cartpole_actions"], "first_dof", power_scale=float(g["cartpole_max_push"]), num_dof=2)
    _eq(f.reshape(-1), g["cartpole_forces"])


def _variants(g):
    return g["meta"]["variants"], g["meta"]


def test_schedule_arithmetic_matches_apply_randomizations(g):
    variants, _ = _variants(g)
    for name, v in variants.items():
        last_rand, first = -1, True
        for step, frame in enumerate(v["frames"]):
            do = first or (frame - last_rand) >= v["frequency"]
            assert int(do) == int(g[name + "_did_randomize"][step])
            if do:
                last_rand = frame
                pa = randomization.noise_params(v["actions"], frame)
                po = randomization.noise_params(v["observations"], frame)
            for p, rec in ((pa, g[name + "_act_params"][step]), (po, g[name + "_obs_params"][step])):
                keys = ("mu", "var", "mu_corr", "var_corr") if "mu" in p else ("lo", "hi", "lo_corr", "hi_corr")
                np.testing.assert_array_equal(np.array([p[k] for k in keys], np.float64), rec)  # doubles, bit for bit
            # randomize_buf bookkeeping (:643-646)
            before, after = g[name + "_randomize_before"][step], g[name + "_randomize_after"][step]
            if first:
                np.testing.assert_array_equal(before, after)
            else:
                reset_prev = g[name + "_in_reset"][step - 1]
                expect = before.copy()
                expect[(before >= v["frequency"]) & (reset_prev != 0)] = 0
                np.testing.assert_array_equal(expect, after)
            first = False


def test_noise_lambdas_and_step_tail(g):
    variants, meta = _variants(g)
    for name, v in variants.items():
        clip_obs = np.inf if v["clip_obs"] is None else v["clip_obs"]
        made = [randomization.noise_params(v[k], 0) for k in randomization.NONPHYSICAL]
        randomization.bind_operations(made)
        assert made[0]["operation"] == made[1]["operation"] != v["observations"]["operation"]
        for step, frame in enumerate(v["frames"]):
            def coeff(kind, rec):
                keys = ("mu", "var", "mu_corr", "var_corr") if v[kind]["distribution"] == "gaussian" else ("lo", "hi", "lo_corr", "hi_corr")
                p = dict(zip(keys, rec))
                cs, csh, sc, sh = randomization.noise_coefficients(p)
                # both lambdas apply the operation of the last parameter set processed ("actions")
                return dict(corr_scale=cs, corr_shift=csh, scale=sc, shift=sh, operation=made[-1]["operation"])
            na = dict(coeff("actions", g[name + "_act_params"][step]), corr=g[name + "_act_corr"][step],
                      fresh=g[name + "_act_fresh"][step])
            a, _ = oracle.action_stage(g[name + "_in_actions"][step], "force", gain=np.ones(5, np.float32),
                                       clip_actions=v["clip_actions"], noise=na)
            _eq(a, g[name + "_self_actions"][step])
            no = coeff("observations", g[name + "_obs_params"][step])
            noised = oracle.noise_apply(g[name + "_in_obs"][step], g[name + "_obs_corr"][step], g[name + "_obs_fresh"][step], **no)
            _eq(noised, g[name + "_obs_buf_after"][step])
            out, time_outs = oracle.vec_task_tail(noised, clip_obs, g[name + "_in_progress"][step], g[name + "_in_reset"][step],
                                                  meta["max_episode_length"])
            _eq(out, g[name + "_out_obs"][step])
            np.testing.assert_array_equal(time_outs, g[name + "_time_outs"][step])
