"""Action stage and observation / action noise on the GPU, through the C ABI: bit-exact against
the recording of the reference's task classes and real VecTask (tests/golden/prestep.npz) when
the recorded draws are replayed; distribution checks for the draws made inside the kernels."""
import numpy as np
import pytest
import torch

from isaacgym_amd import ops, randomization
from isaacgym_amd.sim import TensorSim
from isaacgym_amd.tasks.base.vec_task import VecTask
from oracle import oracle
from tests import helpers as H

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.cpu().numpy()


def eq(a, b, what=""):
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    assert a.shape == b.shape, what
    assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), what


@pytest.fixture(scope="module")
def g():
    return H.load_golden("prestep")


def test_force_scaling_golden(g):
    for tag, gain in (("ant", "ant_gain"), ("ant_ps037", "ant_ps037_gain"), ("humanoid", "humanoid_gain")):
        a = dev(g[tag + "_actions"])
        keep, f = torch.zeros_like(a), torch.zeros_like(a)
        ops.action_stage(a, f, "force", gain=dev(g[gain]), power_scale=float(g[tag + "_power_scale"]), actions_out=keep)
        eq(host(f), g[tag + "_forces"], tag)
        eq(host(keep), g[tag + "_actions"], tag)
    a = dev(g["amp_actions"])
    f = torch.zeros_like(a)
    ops.action_stage(a, f, "force", gain=dev(g["amp_efforts"]), power_scale=float(g["amp_power_scale"]))
    eq(host(f), g["amp_forces"])


def test_pd_targets_and_cartpole_golden(g):
    a = dev(g["amp_actions"])
    t = torch.zeros_like(a)
    ops.action_stage(a, t, "pd_target", gain=dev(g["amp_scale"]), offset=dev(g["amp_offset"]))
    eq(host(t), g["amp_pd_targets"])
    a = dev(g["cartpole_actions"])
    f = torch.full((a.numel() * 2,), 7.0, device=DEV)  # every element is written, zeros included
    ops.action_stage(a, f, "first_dof", power_scale=float(g["cartpole_max_push"]), num_dof=2)
    eq(host(f), g["cartpole_forces"])


def _spec(v, kind, rec, corr, op):
    keys = ("mu", "var", "mu_corr", "var_corr") if v[kind]["distribution"] == "gaussian" else ("lo", "hi", "lo_corr", "hi_corr")
    cs, csh, sc, sh = randomization.noise_coefficients(dict(zip(keys, rec)))
    return ops.NoiseSpec(dev(corr), cs, csh, sc, sh, op, v[kind]["distribution"])


@pytest.mark.parametrize("name", ["v1", "v2"])
def test_noise_kernels_golden(g, name):
    v, meta = g["meta"]["variants"][name], g["meta"]
    op = "additive" if v["actions"]["operation"] == "additive" else "scaling"  # the one both lambdas apply
    clip_obs = float("inf") if v["clip_obs"] is None else v["clip_obs"]
    for step in range(meta["T"]):
        # actions: noise + clamp in the action stage
        a = dev(g[name + "_in_actions"][step])
        keep, f = torch.zeros_like(a), torch.zeros_like(a)
        spec = _spec(v, "actions", g[name + "_act_params"][step], g[name + "_act_corr"][step], op)
        ops.action_stage(a, f, "force", gain=torch.ones(5, device=DEV), clip_actions=v["clip_actions"], actions_out=keep,
                         noise=spec, fresh=dev(g[name + "_act_fresh"][step]))
        eq(host(keep), g[name + "_self_actions"][step], "actions %d" % step)
        eq(host(f), g[name + "_self_actions"][step], "unit gain %d" % step)
        # observations: stand-alone lambda, then fused into the step tail
        spec = _spec(v, "observations", g[name + "_obs_params"][step], g[name + "_obs_corr"][step], op)
        x, fresh = dev(g[name + "_in_obs"][step]), dev(g[name + "_obs_fresh"][step])
        eq(host(ops.noise_apply(spec, x, fresh=fresh)), g[name + "_obs_buf_after"][step], "lambda %d" % step)
        out = torch.zeros_like(x)
        tout = torch.zeros(meta["N"], dtype=torch.bool, device=DEV)
        ops.vec_task_tail(x, out, clip_obs, dev(g[name + "_in_progress"][step]), dev(g[name + "_in_reset"][step]), tout,
                          meta["max_episode_length"], noise=spec, fresh=fresh, obs_noised=x)
        eq(host(x), g[name + "_obs_buf_after"][step], "obs_buf %d" % step)
        eq(host(out), g[name + "_out_obs"][step], "obs %d" % step)
        eq(host(tout), g[name + "_time_outs"][step], "time_outs %d" % step)


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("name", ["v1", "v2"])
def test_vec_task_randomized_step_golden(g, name, fused):
    """apply_randomizations + step of the VecTask class, the recorded torch draws replayed."""
    v, meta = g["meta"]["variants"][name], g["meta"]
    N, T = meta["N"], meta["T"]

    class Task(VecTask):
        FUSED_ACTION_STAGE = fused

        def pre_physics_step(self, actions):
            if fused:
                self._scale_actions(actions, self._forces, "force", gain=self._ones, keep=self.actions)
            else:
                self.actions.copy_(actions)

        def post_physics_step(self):
            t = self.control_steps
            self.obs_buf.copy_(dev(g[name + "_in_obs"][t]))
            self.reset_buf.copy_(dev(g[name + "_in_reset"][t]))
            self.progress_buf.copy_(dev(g[name + "_in_progress"][t]))

    env = dict(numEnvs=N, numObservations=v["num_obs"], numActions=5, clipActions=v["clip_actions"])
    if v["clip_obs"] is not None:
        env["clipObservations"] = v["clip_obs"]
    sim = TensorSim(N, DEV)
    task = Task(dict(env=env, sim=dict(dt=0.0166)), DEV, DEV, sim=sim)
    task.max_episode_length = meta["max_episode_length"]
    task.actions = torch.zeros(N, 5, device=DEV)
    task._forces, task._ones = torch.zeros(N, 5, device=DEV), torch.ones(5, device=DEV)
    did = g[name + "_did_randomize"]
    task.replay_noise = {
        "actions": [dev(x) for x in g[name + "_act_fresh"]], "observations": [dev(x) for x in g[name + "_obs_fresh"]],
        "actions_corr": [dev(g[name + "_act_corr"][t]) for t in range(T) if did[t]],
        "observations_corr": [dev(g[name + "_obs_corr"][t]) for t in range(T) if did[t]]}
    dr = dict(frequency=v["frequency"], observations=v["observations"], actions=v["actions"], actor_params={})
    for t in range(T):
        sim.num_simulate_calls = v["frames"][t]  # gym.get_frame_count
        task.randomize_buf.copy_(dev(g[name + "_randomize_before"][t]))
        before = task.last_rand_step
        task.apply_randomizations(dr)
        assert int(task.last_rand_step != before) == int(did[t])
        eq(host(task.randomize_buf), g[name + "_randomize_after"][t], "randomize_buf %d" % t)
        for kind, rec in (("actions", g[name + "_act_params"][t]), ("observations", g[name + "_obs_params"][t])):
            p = task.dr_randomizations[kind]
            keys = ("mu", "var", "mu_corr", "var_corr") if "mu" in p else ("lo", "hi", "lo_corr", "hi_corr")
            assert [p[k] for k in keys] == list(rec), (kind, t)
        obs_dict, rew, reset, extras = task.step(dev(g[name + "_in_actions"][t]))
        eq(host(task.actions), g[name + "_self_actions"][t], "actions %d" % t)
        eq(host(task.obs_buf), g[name + "_obs_buf_after"][t], "obs_buf %d" % t)
        eq(host(obs_dict["obs"]), g[name + "_out_obs"][t], "obs %d" % t)
        eq(host(extras["time_outs"]), g[name + "_time_outs"][t], "time_outs %d" % t)
    assert not any(task.replay_noise.values())  # every recorded draw was consumed, in the reference's order


@pytest.mark.parametrize("count,offset", [(1, 0), (5, 1), (1023, 3), (4099, 2)])
def test_ragged_and_unaligned(count, offset):
    """Element counts that are no multiple of four and pointers off the 16-byte grid."""
    rng = np.random.default_rng(count)
    def arr():
        return rng.standard_normal(count + offset).astype(np.float32)
    x, corr, fresh = arr(), arr(), arr()
    spec = ops.NoiseSpec(dev(corr)[offset:], 0.3, -0.1, 0.7, 0.05, "scaling", "gaussian")
    out = torch.zeros(count + offset, device=DEV)
    ops.noise_apply(spec, dev(x)[offset:], out=out[offset:], fresh=dev(fresh)[offset:])
    want = oracle.noise_apply(x[offset:], corr[offset:], fresh[offset:], 0.3, -0.1, 0.7, 0.05, "scaling")
    eq(host(out)[offset:], want)
    assert not host(out)[:offset].any()
    # action stage with 7 actions per env
    n = max(1, count // 7)
    a = rng.standard_normal(n * 7 + offset).astype(np.float32) * 2
    gain, off = rng.standard_normal(7).astype(np.float32), rng.standard_normal(7).astype(np.float32)
    for mode in ("force", "pd_target"):
        keep, t = torch.zeros(n * 7 + offset, device=DEV), torch.zeros(n * 7 + offset, device=DEV)
        ops.action_stage(dev(a)[offset:], t[offset:], mode, gain=dev(gain), offset=dev(off), power_scale=0.37,
                         clip_actions=1.5, actions_out=keep[offset:])
        wa, wt = oracle.action_stage(a[offset:].reshape(n, 7), mode, gain=gain, offset=off, power_scale=0.37, clip_actions=1.5)
        eq(host(keep)[offset:], wa.reshape(-1), mode)
        eq(host(t)[offset:], wt.reshape(-1), mode)


def test_first_dof_with_noise_matches_oracle():
    rng = np.random.default_rng(3)
    n = 1001
    a, corr, fresh = (rng.standard_normal(n).astype(np.float32) for _ in range(3))
    spec = ops.NoiseSpec(dev(corr), 0.02, 0.01, 0.1, 0.0, "additive", "uniform")
    f, keep = torch.zeros(n * 3, device=DEV), torch.zeros(n, device=DEV)
    ops.action_stage(dev(a), f, "first_dof", power_scale=400.0, clip_actions=1.0, num_dof=3, actions_out=keep,
                     noise=spec, fresh=dev(fresh))
    wa, wt = oracle.action_stage(a, "first_dof", power_scale=400.0, clip_actions=1.0, num_dof=3,
                                 noise=dict(corr=corr, fresh=fresh, corr_scale=0.02, corr_shift=0.01, scale=0.1, shift=0.0,
                                            operation="additive"))
    eq(host(keep), wa.reshape(-1))
    eq(host(f), wt.reshape(-1))


@pytest.mark.parametrize("distribution", ["gaussian", "uniform"])
def test_in_kernel_draws(distribution):
    """fresh == NULL: Philox draws in the kernel.  Not torch's stream, so the checks are on the
    distribution (Kolmogorov-Smirnov, moments), on reproducibility and on counter separation."""
    from scipy import stats
    n = 1 << 20
    zeros = torch.zeros(n, device=DEV)
    spec = ops.NoiseSpec(zeros, 0.0, 0.0, 1.0, 0.0, "additive", distribution, seed=1234)
    d0 = host(ops.noise_apply(spec, zeros))   # offset 0
    d1 = host(ops.noise_apply(spec, zeros))   # offset 1
    again = ops.NoiseSpec(zeros, 0.0, 0.0, 1.0, 0.0, "additive", distribution, seed=1234)
    eq(host(ops.noise_apply(again, zeros)), d0)
    other = host(ops.noise_apply(ops.NoiseSpec(zeros, 0.0, 0.0, 1.0, 0.0, "additive", distribution, seed=1235), zeros))
    assert np.isfinite(d0).all()
    for a, b in ((d0, d1), (d0, other)):
        assert abs(np.corrcoef(a, b)[0, 1]) < 5e-3 and not np.array_equal(a, b)
    # neighbours inside one Philox quad and across quads are uncorrelated
    assert abs(np.corrcoef(d0[:-1], d0[1:])[0, 1]) < 5e-3 and abs(np.corrcoef(d0[:-4], d0[4:])[0, 1]) < 5e-3
    if distribution == "gaussian":
        assert abs(d0.mean()) < 5e-3 and abs(d0.std() - 1.0) < 5e-3
        assert stats.kstest(d0[:200000].astype(np.float64), "norm").pvalue > 1e-3
        assert abs(stats.kurtosis(d0.astype(np.float64))) < 0.05
    else:
        assert d0.min() >= 0.0 and d0.max() < 1.0
        assert abs(d0.mean() - 0.5) < 2e-3 and abs(d0.var() - 1.0 / 12.0) < 1e-3
        assert stats.kstest(d0[:200000].astype(np.float64), "uniform").pvalue > 1e-3
    # the draw enters the same arithmetic as a supplied one
    corr = torch.randn(n, device=DEV)
    x = torch.randn(n, device=DEV)
    spec = ops.NoiseSpec(corr, 0.3, -0.1, 0.7, 0.05, "scaling", distribution, seed=1234)
    got = host(ops.noise_apply(spec, x))
    eq(got, oracle.noise_apply(host(x), host(corr), d0, 0.3, -0.1, 0.7, 0.05, "scaling"))


def test_bad_arguments_raise():
    x = torch.zeros(8, device=DEV)
    with pytest.raises(ValueError):
        ops.NoiseSpec(x, 0, 0, 1, 0, operation="subtract")
    with pytest.raises(ValueError):
        ops.noise_apply(ops.NoiseSpec(x[:4], 0, 0, 1, 0), x)
    with pytest.raises(ValueError):
        ops.action_stage(x, x.clone(), "force", gain=torch.ones(3, device=DEV))
    with pytest.raises(RuntimeError):
        ops.action_stage(x, x.clone(), "force", gain=torch.ones(4, device=DEV), clip_actions=-1.0)


@pytest.mark.filterwarnings("ignore:overflow encountered:RuntimeWarning")  # 3e38 times a gain: Inf on both sides, as intended
def test_extreme_values_through_the_wrappers():
    """NaN, +-Inf, +-3e38, 1e30, 1e-30, subnormals and signed zeros in the actions, the observations and the
    persistent noise draws: the action stage, the noise lambdas, the clamp and the step tail against the CPU
    restatement (numpy, one rounding per operation, NaN-propagating minimum / maximum like torch.clamp)."""
    rng = np.random.default_rng(11)
    specials = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30, -1e30, 1e-30, 1e-40, -1e-45, 0.0, -0.0, 1e19, 1e-20],
                        np.float32)

    def arr(n, frac=0.05):
        x = rng.standard_normal(n).astype(np.float32) * 2
        idx = rng.integers(0, n, max(1, int(n * frac)))
        x[idx] = specials[rng.integers(0, len(specials), len(idx))]
        return x

    def same(got, want, what):
        assert H.bit_equal_fraction(got, want) == 1.0, what

    n = 4099
    x, corr, fresh = arr(n), arr(n), arr(n)
    for operation in ("additive", "scaling"):
        spec = ops.NoiseSpec(dev(corr), 0.3, -0.1, 0.7, 0.05, operation, "gaussian")
        out = torch.zeros(n, device=DEV)
        ops.noise_apply(spec, dev(x), out=out, fresh=dev(fresh))
        same(host(out), oracle.noise_apply(x, corr, fresh, 0.3, -0.1, 0.7, 0.05, operation), "noise " + operation)
    envs = n // 7
    a = arr(envs * 7)
    gain, off = arr(7, 0.3), arr(7, 0.3)
    for mode in ("force", "pd_target"):
        keep, t = torch.zeros(envs * 7, device=DEV), torch.zeros(envs * 7, device=DEV)
        ops.action_stage(dev(a), t, mode, gain=dev(gain), offset=dev(off), power_scale=0.37, clip_actions=1.5, actions_out=keep)
        wa, wt = oracle.action_stage(a.reshape(envs, 7), mode, gain=gain, offset=off, power_scale=0.37, clip_actions=1.5)
        same(host(keep), wa.reshape(-1), mode + " actions")
        same(host(t), wt.reshape(-1), mode + " targets")
    clamped = torch.zeros(n, device=DEV)
    ops.clamp(dev(x), -1.25, 1.25, clamped)
    assert torch.equal(torch.nan_to_num(clamped, nan=7.0), torch.nan_to_num(torch.clamp(dev(x), -1.25, 1.25), nan=7.0))
    assert bool((clamped.isnan() == dev(x).isnan()).all())
    obs = arr(envs * 7).reshape(envs, 7)
    progress, reset = rng.integers(0, 40, envs), rng.integers(0, 2, envs)
    out, to = torch.zeros(envs, 7, device=DEV), torch.zeros(envs, dtype=torch.bool, device=DEV)
    ops.vec_task_tail(dev(obs), out, 2.5, dev(progress), dev(reset), to, 30)
    want_obs, want_to = oracle.vec_task_tail(obs, 2.5, progress, reset, 30)
    same(host(out), want_obs, "tail obs")
    assert np.array_equal(host(to), want_to)
