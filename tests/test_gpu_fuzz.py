"""Seeded differential fuzzing of the step kernels against the CPU checker: random sizes around the
tile boundaries, random values for every tunable of the parameter structs, random reset fractions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from isaacgym_amd import ops, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402
from tests import helpers as H  # noqa: E402

DEV = "cuda:0"
# a longer campaign: IGX_FUZZ_TRIALS multiplies the number of trials of every test, IGX_FUZZ_SEED moves the seeds
import os  # noqa: E402
TRIALS = max(1, int(os.environ.get("IGX_FUZZ_TRIALS", "1")))
SEED = int(os.environ.get("IGX_FUZZ_SEED", "0")) * 1000


def _sizes(rng, tile, count):
    out = []
    for _ in range(count):
        k = int(rng.integers(0, 40))
        out.append(max(1, k * tile + int(rng.integers(-2, 3))))
    return out


def _same(a, b):
    """torch.equal that takes two NaNs in one place as equal."""
    return a.shape == b.shape and bool(((a == b) | (a.isnan() & b.isnan())).all())


def _poison(rng, t, names, frac=0.01):
    """NaN, +-Inf, huge, tiny and subnormal values into about `frac` of the entries of the named float tensors."""
    specials = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30, 1e-30, 1e-40, -1e-45, 0.0, -0.0], np.float32)
    for k in names:
        v = t.get(k)
        if v is None or v.numel() == 0:
            continue
        m = max(1, int(v.numel() * frac))
        idx = torch.from_numpy(rng.integers(0, v.numel(), m)).to(v.device)
        vals = torch.from_numpy(specials[rng.integers(0, len(specials), m)]).to(v.device)
        v.view(-1)[idx] = vals


@pytest.mark.parametrize("humanoid", [False, True])
def test_loco_fuzz(humanoid):
    spec = synthetic.HUMANOID if humanoid else synthetic.ANT
    step, ostep = (ops.humanoid_step, oracle.humanoid_step) if humanoid else (ops.ant_step, oracle.ant_step)
    rng = np.random.default_rng(100 + humanoid + SEED)
    nobs = spec["num_obs"]
    nonangle = [c for c in range(nobs) if c not in H.LOCO_ANGLE_COLS]
    for trial, n in enumerate(_sizes(rng, 32 if humanoid else 64, 12 * TRIALS)):
        p = synthetic.loco_params(spec, seed=int(rng.integers(1 << 40)), rng_offset=int(rng.integers(1 << 20)))
        for name in ("dof_vel_scale", "contact_force_scale", "angular_velocity_scale", "up_weight", "heading_weight",
                     "actions_cost_scale", "energy_cost_scale", "joints_at_limit_cost_scale"):
            setattr(p, name, float(rng.uniform(0.01, 2.0)))
        p.dt = float(rng.uniform(0.005, 0.05))
        p.termination_height = float(rng.uniform(0.2, 1.4))
        p.death_cost = float(rng.uniform(-5.0, 0.0))
        p.max_episode_length_m1 = float(rng.integers(5, 1500))
        p.set("targets", rng.uniform(-500, 500, 3).tolist())
        q = rng.standard_normal(4)
        p.set("inv_start_rot", (q / np.linalg.norm(q)).tolist())
        with_rand = bool(rng.integers(2))
        t = synthetic.loco_state(spec, n, int(rng.integers(1 << 30)), DEV, with_rand)
        t["reset"] = (torch.rand(n, device=DEV) < float(rng.choice([0.0, 0.05, 0.5, 1.0]))).to(torch.int64)
        if trial % 3 == 2:  # non-finite, huge, tiny and subnormal simulator state and actions
            _poison(rng, t, ("root_states", "dof_state", "sensors", "actions", "dof_force", "potentials"))
        if not humanoid and rng.integers(2):
            t["true_objective"] = torch.zeros(n, device=DEV)
        if rng.integers(2):
            p.clip_obs = float(rng.uniform(0.1, 3.0))
            t["obs_clamped"] = torch.zeros(n, nobs, device=DEV)
            t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        step(p, t, n)
        torch.cuda.synchronize()
        ostep(p, ref, n)
        what = "trial %d n %d" % (trial, n)
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        for k in ("dof_state", "potentials", "prev_potentials", "up_vec", "heading_vec", "rew"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)
        obs = t["obs"].cpu().numpy()
        assert H.bit_equal_fraction(obs[:, nonangle], ref["obs"][:, nonangle]) == 1.0, what
        H.assert_angle_close(obs[:, list(H.LOCO_ANGLE_COLS)], ref["obs"][:, list(H.LOCO_ANGLE_COLS)], what=what,
                             branch_cut=H.loco_branch_cut(humanoid))
        if "time_outs" in t:
            assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), what
            assert _same(t["obs_clamped"], torch.clamp(t["obs"], -p.clip_obs, p.clip_obs)), what


def test_cartpole_fuzz():
    rng = np.random.default_rng(7 + SEED)
    for trial, n in enumerate(_sizes(rng, 64, 12 * TRIALS)):
        p = synthetic.cartpole_params(seed=int(rng.integers(1 << 40)), rng_offset=int(rng.integers(1 << 20)))
        p.reset_dist = float(rng.uniform(0.5, 4.0))
        p.max_episode_length_m1 = float(rng.integers(3, 600))
        p.reward_variant = int(rng.integers(2))
        t = synthetic.cartpole_state(n, int(rng.integers(1 << 30)), DEV, bool(rng.integers(2)))
        t["reset"] = (torch.rand(n, device=DEV) < float(rng.choice([0.0, 0.1, 1.0]))).to(torch.int64)
        if trial % 3 == 2:
            _poison(rng, t, ("dof_state",), 0.03)
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        ops.cartpole_step(p, t, n)
        torch.cuda.synchronize()
        oracle.cartpole_step(p, ref, n)
        for k in ("reset", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, trial, n)
        for k in ("dof_state", "obs", "rew"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, trial, n)


@pytest.mark.parametrize("hand", ["allegro", "shadow"])
def test_hand_fuzz(hand):
    rng = np.random.default_rng((31 if hand == "allegro" else 32) + SEED)
    obs_types = ["full_no_vel", "full", "full_state"] + (["openai"] if hand == "shadow" else [])
    mk_p, mk_s = (synthetic.allegro_params, synthetic.allegro_state) if hand == "allegro" else (synthetic.shadow_params, synthetic.shadow_state)
    pre, post = (ops.allegro_pre_physics, ops.allegro_post_physics) if hand == "allegro" else (ops.shadow_pre_physics, ops.shadow_post_physics)
    opre, opost = ((oracle.allegro_pre_physics, oracle.allegro_post_physics) if hand == "allegro"
                   else (oracle.shadow_pre_physics, oracle.shadow_post_physics))
    for trial, n in enumerate(_sizes(rng, 64, 10 * TRIALS)):
        obs_type = str(rng.choice(obs_types))
        asym = bool(rng.integers(2))
        p = mk_p(obs_type, seed=int(rng.integers(1 << 40)), rng_offset=int(rng.integers(1 << 20)))
        p.use_relative_control = int(rng.integers(2))
        p.max_consecutive_successes = int(rng.choice([0, 3]))
        for name in ("dist_reward_scale", "action_penalty_scale", "fall_penalty"):
            setattr(p, name, -float(rng.uniform(0.0, 12.0)))
        p.rot_reward_scale = float(rng.uniform(0.1, 2.0))
        p.reach_goal_bonus = float(rng.uniform(1.0, 300.0))
        p.fall_dist = float(rng.uniform(0.1, 0.4))
        ama = float(rng.uniform(0.1, 1.0))
        p.act_moving_average, p.one_minus_act_moving_average = ama, 1.0 - ama
        p.max_episode_length_m1 = float(rng.integers(3, 700))
        t = mk_s(n, int(rng.integers(1 << 30)), DEV, obs_type, asym)
        frac = float(rng.choice([0.0, 0.1, 1.0]))
        t["reset"] = (torch.rand(n, device=DEV) < frac).to(torch.int64)
        t["reset_goal"] = (torch.rand(n, device=DEV) < frac).to(torch.int64)
        if trial % 3 == 2:
            _poison(rng, t, ("dof_state", "dof_force", "root_state_tensor", "goal_states", "actions", "prev_targets",
                             "rigid_body_states", "force_sensors"), 0.003)
        ref = {k: (v.cpu().numpy().copy() if v is not None else None) for k, v in t.items()}
        what = "%s trial %d n %d %s%s" % (hand, trial, n, obs_type, " poisoned" if trial % 3 == 2 else "")
        pre(p, t, n)
        torch.cuda.synchronize()
        opre(p, ref, n)
        for k in ("reset", "reset_goal", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        for k in ("prev_targets", "cur_targets", "dof_state", "successes"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)
        H.assert_close(t["goal_states"].cpu().numpy(), ref["goal_states"], what=what)
        H.assert_close(t["root_state_tensor"].cpu().numpy(), ref["root_state_tensor"], what=what)
        # both sides go on from the same (sin / cos dependent) reset rotations
        t["goal_states"].copy_(torch.from_numpy(ref["goal_states"]))
        t["root_state_tensor"].copy_(torch.from_numpy(ref["root_state_tensor"]))
        post(p, t, n)
        torch.cuda.synchronize()
        opost(p, ref, n)
        for k in ("reset", "reset_goal", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        assert H.bit_equal_fraction(t["successes"].cpu().numpy(), ref["successes"]) == 1.0, what
        assert H.bit_equal_fraction(t["obs"].cpu().numpy(), ref["obs"]) == 1.0, what
        if asym:
            assert H.bit_equal_fraction(t["states"].cpu().numpy(), ref["states"]) == 1.0, what
        H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what=what)
        H.assert_close(t["consecutive_successes"].cpu().numpy(), ref["consecutive_successes"], what=what)


def amp_trials():
    """The inputs of test_amp_fuzz, trial by trial: (trial, n, K, params, tensors).  (A generator of its own so that
    tools/debug_amp_trial.py replays exactly these.)"""
    rng = np.random.default_rng(41 + SEED)
    for trial, n in enumerate(_sizes(rng, 32, 10 * TRIALS)):
        K = int(rng.choice([2, 2, 3, 4, 5]))
        p = synthetic.amp_params(K, bool(rng.integers(2)), bool(rng.integers(2)), int(rng.integers(3, 400)))
        p.termination_height = float(rng.uniform(0.2, 1.0))
        if rng.integers(2):  # other contact-exempt bodies (0 - 4 of them) and other key bodies than the hands and feet
            nc = int(rng.integers(0, 5))
            p.num_contact_bodies = nc
            p.set("contact_body_ids", ([int(v) for v in rng.choice(15, nc, replace=False)] + [0, 0, 0, 0])[:4])
            p.set("key_body_ids", [int(v) for v in rng.integers(0, 15, 4)])
        t = synthetic.amp_state(n, int(rng.integers(1 << 30)), DEV, K, int(p.max_episode_length_m1) + 1)
        if rng.integers(2):
            p.clip_obs = float(rng.uniform(0.1, 3.0))
            t["obs_clamped"] = torch.zeros(n, 105, device=DEV)
            t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
        if trial % 3 == 2:
            _poison(rng, t, ("root_states", "dof_state", "rigid_body_state", "contact_forces"), 0.002)
        if trial % 4 == 1:  # output buffers that are only 4-byte aligned (views one float into a larger allocation)
            for k in ("obs", "amp_obs", "obs_clamped"):
                if k in t:
                    big = torch.zeros(t[k].numel() + 1, device=DEV)
                    big[1:] = t[k].reshape(-1)
                    t[k] = big[1:].view(t[k].shape)
        yield trial, n, K, p, t


def test_amp_fuzz():
    """HumanoidAMP step: random batch sizes around the 32-row tile, 2 - 5 history steps (the three specialised
    kernels and the general one), both root-observation frames, random thresholds, optional tail outputs."""
    for trial, n, K, p, t in amp_trials():
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        root_quat = ref["root_states"][:, 3:7].copy()
        what = "amp trial %d n %d K %d%s" % (trial, n, K, " poisoned" if trial % 3 == 2 else "")
        ops.humanoid_amp_step(p, t, n)
        torch.cuda.synchronize()
        oracle.humanoid_amp_step(p, ref, n)
        for k in ("reset", "terminate", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        assert np.array_equal(t["rew"].cpu().numpy(), ref["rew"]), what
        H.assert_amp_obs_close(t["obs"].cpu().numpy(), ref["obs"], what=what, quiet=True, root_quat=root_quat)
        got, want = t["amp_obs"].cpu().numpy(), ref["amp_obs"]
        H.assert_amp_obs_close(got[:, 0], want[:, 0], what=what + " amp_obs", quiet=True, root_quat=root_quat)
        assert H.bit_equal_fraction(got[:, 1:], want[:, 1:]) == 1.0, what  # shifted history: copies
        if "time_outs" in t:
            assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), what
            assert _same(t["obs_clamped"], torch.clamp(t["obs"], -p.clip_obs, p.clip_obs)), what


def test_motion_fuzz():
    """Motion library: random sample counts around the 16-sample block, times beyond both ends of the clips,
    the state kernel and the fused observation kernels (one step and tiled over 1 - 4 history steps)."""
    g = H.load_golden("motion")
    tabs_np = H.motion_tabs_from_golden(g)
    tabs_dev = {k: torch.from_numpy(v).to(DEV) for k, v in tabs_np.items()}
    tables = ops.motion_tables(tabs_dev)
    rng = np.random.default_rng(43 + SEED)
    for trial, n in enumerate(_sizes(rng, 16, 10 * TRIALS)):
        ids = rng.integers(0, 2, n).astype(np.int64)
        # up to ~0.12 s before the clip's start (the reference reaches back numAMPObsSteps * dt, humanoid_amp.py:158-159,
        # :283-284: the blend is then negative and slerp extrapolates) and 20 % beyond its end (clamped to the last frame).
        # Far before the start the extrapolation multiplies the last-bit differences of acos / sin by |blend|: at
        # -0.3 clip lengths (blend -27 ... -54) the root tan-norm columns of this kernel and the checker are 1.5e-5 apart.
        times = rng.uniform(-0.04, 1.2, n) * tabs_np["motion_length"][ids]
        K, dt, lro = int(rng.integers(1, 5)), float(rng.uniform(0.005, 0.08)), bool(rng.integers(2))
        what = "motion trial %d n %d K %d" % (trial, n, K)
        out = torch.zeros(n, K * 105, device=DEV)
        ops.motion_amp_obs_steps(tables, torch.from_numpy(ids).to(DEV), torch.from_numpy(times).to(DEV), K, dt, lro, out)
        want = oracle.motion_amp_obs_steps(tabs_np, ids, times, K, dt, lro)
        got = out.cpu().numpy()
        # frames are exact (fp64 index arithmetic), so the copied joint velocities are bit-equal; the rest goes
        # through slerp / acos / sin: 1e-5 per group, except joints on the identity knife edge of quat_to_angle_axis
        gv, wv = got.reshape(-1, 105)[:, 65:93], want.reshape(-1, 105)[:, 65:93]
        assert np.array_equal(gv.view(np.uint32), wv.view(np.uint32)), what
        # which joints sit on the knife edge (masked to zero on one side only): from the joint angles of the same samples
        ids_t = np.tile(np.expand_dims(ids, -1), [1, K]).flatten()
        times_t = (np.expand_dims(times, -1) + (-dt * np.arange(0, K))).flatten()
        dp = torch.zeros(n * K, 28, device=DEV)
        ops.motion_state(tables, torch.from_numpy(ids_t).to(DEV), torch.from_numpy(times_t).to(DEV), dict(dof_pos=dp))
        dp_gpu, dp_cpu = dp.cpu().numpy(), oracle.motion_state(tabs_np, ids_t, times_t)["dof_pos"]
        disagree, _ = H.knife_edge_joints(dp_gpu, dp_cpu)
        # the reference's second edge: a joint turned by pi is an identity observation one ulp above pi (H.half_turn_joints)
        half_turn = H.half_turn_joints(dp_gpu, dp_cpu)
        exempt, exempt_pi = np.zeros((n * K, 105), bool), np.zeros((n * K, 105), bool)
        for j, cols in enumerate(H.amp_joint_columns("obs")):
            exempt[:, cols] |= disagree[:, j:j + 1]
            exempt_pi[:, cols] |= half_turn[:, j:j + 1]
        g2, w2 = got.reshape(-1, 105), want.reshape(-1, 105)
        H.assert_amp_obs_close(g2, w2, what=what, exempt=exempt | exempt_pi, quiet=True)
        jump = np.abs(g2 - w2)[exempt & ~exempt_pi]
        # (the recording's clips hold joints at exactly zero rotation; a few samples per thousand land on the edge)
        assert disagree.sum() <= 2 + (n * K) // 250 and (jump.size == 0 or jump.max() < 2e-2), (what, int(disagree.sum()))
        # a half-turn joint either agrees or is the identity (tangent x, normal z) on exactly one side
        identity = np.array([1, 0, 0, 0, 0, 1], np.float32)
        for r, j in zip(*np.nonzero(half_turn)):
            cols = H.amp_joint_columns("obs")[j]
            a, b = g2[r, cols], w2[r, cols]
            assert np.abs(a - b).max() < 1e-5 or np.array_equal(a, identity) != np.array_equal(b, identity), (what, r, j, a, b)
        assert half_turn.sum() <= 2 + (n * K) // 250, (what, int(half_turn.sum()))


def test_humanoid_imitate_fuzz():
    """HumanoidImitate step: random batch sizes around the 16-row tile, random thresholds and weights, host-given or
    in-kernel reset draws, optional tail outputs."""
    rng = np.random.default_rng(47 + SEED)
    for trial, n in enumerate(_sizes(rng, 16, 8 * TRIALS)):
        p = synthetic.him_params(seed=int(rng.integers(1 << 30)), rng_offset=int(rng.integers(1000)),
                                 termination_radius=float(rng.uniform(0.3, 2.0)))
        p.termination_height = float(rng.uniform(0.4, 1.1))
        p.up_weight = float(rng.uniform(0.0, 1.0))
        p.actions_cost_scale = float(rng.uniform(0.0, 0.1))
        p.dof_vel_scale = float(rng.uniform(0.05, 1.0))
        p.max_episode_length_m1 = float(rng.integers(3, 1200))
        t = synthetic.him_state(n, int(rng.integers(1 << 30)), DEV, bool(rng.integers(2)))
        t["progress"] = torch.from_numpy(rng.integers(0, int(p.max_episode_length_m1) + 2, n)).to(DEV)
        t["reset"] = torch.from_numpy((rng.random(n) < rng.choice([0.0, 0.05, 1.0])).astype(np.int64)).to(DEV)
        if rng.integers(2):
            p.clip_obs = float(rng.uniform(0.1, 3.0))
            t["obs_clamped"] = torch.zeros(n, 205, device=DEV)
            t["time_outs"] = torch.zeros(n, dtype=torch.bool, device=DEV)
        if trial % 3 == 2:
            _poison(rng, t, ("root_states", "dof_state", "rigid_body_states", "actions"), 0.002)
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        what = "him trial %d n %d%s" % (trial, n, " poisoned" if trial % 3 == 2 else "")
        ops.humanoid_imitate_step(p, t, n)
        torch.cuda.synchronize()
        oracle.humanoid_imitate_step(p, ref, n)
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        for k in ("obs", "dof_state", "potentials", "prev_potentials"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)
        H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what=what + " rew")
        if "time_outs" in t:
            assert np.array_equal(t["time_outs"].cpu().numpy(), ref["time_outs"]), what
            assert _same(t["obs_clamped"], torch.clamp(t["obs"], -p.clip_obs, p.clip_obs)), what


def test_ant_imitate_fuzz():
    """AntImitate step (observations, exact quantile over all environments, reward, resets): random batch sizes on
    both sides of the one-workgroup select (<= 4096 values) and the multi-launch one, random quantile, distance
    floor, clip length, thresholds; the quantile value and the pose differences are compared bit for bit."""
    from isaacgym_amd import _abi
    rng = np.random.default_rng(53 + SEED)
    lo, hi = synthetic.loco_limits(8)
    lo_t, hi_t = torch.tensor(lo, device=DEV, dtype=torch.float64), torch.tensor(hi, device=DEV, dtype=torch.float64)
    sizes = _sizes(rng, 64, 6 * TRIALS) + [int(rng.integers(4000, 4200)) for _ in range(2 * TRIALS)] + \
        [int(rng.integers(5000, 60000)) for _ in range(2 * TRIALS)]
    for trial, n in enumerate(sizes):
        F = int(rng.integers(4, 80))  # the clip is used up to its last three frames (ant_imitate.py:547)
        p = synthetic.loco_params(synthetic.ANT, seed=int(rng.integers(1 << 30)), rng_offset=int(rng.integers(1000)))
        p.termination_height = float(rng.uniform(0.1, 0.7))
        p.max_episode_length_m1 = float(rng.integers(3, 1200))
        t = synthetic.loco_state(synthetic.ANT, n, int(rng.integers(1 << 30)), DEV, with_rand=bool(rng.integers(2)))
        gen = torch.Generator(device=DEV).manual_seed(int(rng.integers(1 << 30)))
        arrays = dict(motion=lo_t + (hi_t - lo_t) * torch.rand(F, 8, generator=gen, device=DEV, dtype=torch.float64),
                      target_pose_idx=torch.randint(0, F - 1, (n,), generator=gen, device=DEV),
                      pose_diff=torch.zeros(n, dtype=torch.float64, device=DEV))
        q, floor, death = float(rng.choice([0.0, 0.5, 0.9, 0.99, 1.0, rng.random()])), float(rng.uniform(0.0, 4.0)), -2.0
        if trial % 3 == 2:
            _poison(rng, t, ("root_states", "dof_state", "sensors", "actions"), 0.002)
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        ref_arrays = {k: v.cpu().numpy().copy() for k, v in arrays.items()}
        ws = torch.zeros(_abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
        what = "ant imitate trial %d n %d F %d q %r%s" % (trial, n, F, q, " poisoned" if trial % 3 == 2 else "")
        for step in range(2):
            p.rng_offset += 1
            im = ops.imitate_args(arrays["motion"], arrays["target_pose_idx"], arrays["pose_diff"], ws, q, floor, death)
            ops.ant_imitate_step(p, t, im, n)
            torch.cuda.synchronize()
            qv = oracle.ant_imitate_step(p, ref, oracle.imitate_args(ref_arrays, F, q, floor, death, n), ref_arrays, n)
            got_q = float(ws.view(torch.float64)[5])
            assert got_q == qv or (np.isnan(got_q) and np.isnan(qv)), (what, got_q, qv)
            gd, wd = arrays["pose_diff"].cpu().numpy(), ref_arrays["pose_diff"]
            assert bool(((gd.view(np.uint64) == wd.view(np.uint64)) | (np.isnan(gd) & np.isnan(wd))).all()), what
            assert np.array_equal(arrays["target_pose_idx"].cpu().numpy(), ref_arrays["target_pose_idx"]), what
            for k in ("reset", "progress", "randomize"):
                assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
            H.assert_close(t["rew"].cpu().numpy(), ref["rew"], what=what + " rew")
            for k in ("dof_state", "potentials", "prev_potentials"):
                assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)


def test_math_fuzz():
    """torch_jit_utils: random batch sizes around the 256-row tile, magnitudes over six decades, quaternions off unit
    length, and - every third trial - NaN / Inf / huge / tiny / subnormal entries.  Functions built from IEEE add / mul /
    div / sqrt only must equal the CPU checker bit for bit whatever the input; the ones that pass through a math
    library are held to 1e-5 on the finite, unit-quaternion trials."""
    from isaacgym_amd.utils import torch_jit_utils as tj
    from tests.test_oracle_math import OracleTJ
    rng = np.random.default_rng(53 + SEED)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)

    for trial, n in enumerate(_sizes(rng, 256, 6 * TRIALS)):
        poisoned, nonunit = trial % 3 == 2, trial % 2 == 1

        def quats():
            q = rng.standard_normal((n, 4)).astype(np.float32)
            q /= np.maximum(np.linalg.norm(q, axis=-1, keepdims=True), 1e-9).astype(np.float32)
            return (q * np.float32(10.0 ** rng.uniform(-1, 1))) if nonunit else q

        def vecs():
            return (rng.standard_normal((n, 3)) * 10.0 ** rng.uniform(-3, 3, (n, 1))).astype(np.float32)

        qa, qb, v, w, x = quats(), quats(), vecs(), vecs(), vecs()
        if poisoned:
            t = {k: dev(a) for k, a in dict(qa=qa, qb=qb, v=v, w=w, x=x).items()}
            _poison(rng, t, tuple(t), 0.01)
            qa, qb, v, w, x = (t[k].cpu().numpy() for k in ("qa", "qb", "v", "w", "x"))
        pose = np.concatenate([x, qb], -1)
        lo = -np.abs(vecs()[0]) - np.float32(0.1)
        hi = np.abs(vecs()[0]) + np.float32(0.1)
        what = "math trial %d n %d%s%s" % (trial, n, " poisoned" if poisoned else "", " nonunit" if nonunit else "")
        exact = [("quat_mul", (qa, qb)), ("quat_rotate", (qa, v)), ("quat_rotate_inverse", (qa, v)), ("quat_conjugate", (qa,)),
                 ("quat_to_tan_norm", (qa,)), ("quat_apply", (qa, v)), ("tf_inverse", (qa, v)), ("tf_combine", (qa, v, qb, w)),
                 ("tf_apply", (qa, v, w)), ("tf_vector", (qa, v)), ("get_basis_vector", (qa, v)),
                 ("compute_heading_and_up", (qa, qb, v, w, x, int(rng.integers(0, 3)))), ("local_to_world_space", (v, pose)),
                 ("quat_axis", (qa, int(rng.integers(0, 3)))), ("normalize", (v,)), ("copysign", (float(rng.uniform(-3, 3)), v[:, 0])),
                 ("scale", (v, lo, hi)), ("unscale", (v, lo, hi)), ("tensor_clamp", (v, lo, hi)), ("scale_transform", (v, lo, hi)),
                 ("unscale_transform", (v, lo, hi)), ("saturate", (v, lo, hi))]
        for name, args in exact:
            got = getattr(tj, name)(*[dev(a) if isinstance(a, np.ndarray) else a for a in args])
            got = tuple(g.cpu().numpy() for g in (got if isinstance(got, tuple) else (got,)))
            if name in ("quat_mul", "quat_rotate", "quat_rotate_inverse", "quat_conjugate", "quat_to_tan_norm"):
                ref = (oracle.rowwise(name, *args),)
            elif name in ("scale", "unscale", "tensor_clamp"):
                ref = (oracle.affine(name, *args),)
            else:
                ref = getattr(OracleTJ, name)(*args)
                ref = ref if isinstance(ref, tuple) else (ref,)
            for k, (a, b) in enumerate(zip(got, ref)):
                assert H.bit_equal_fraction(a, b.reshape(a.shape)) == 1.0, (what, name, k)
        # compute_rot: the two rotated vectors are exact whatever the input
        got = tj.compute_rot(dev(qa), dev(v), dev(w), dev(x), dev(v))
        ref = OracleTJ.compute_rot(qa, v, w, x, v)
        for k in range(2):
            assert H.bit_equal_fraction(got[k].cpu().numpy(), ref[k]) == 1.0, (what, "compute_rot", k)
        if poisoned or nonunit:
            continue
        for k in range(2, 6):
            H.assert_angle_close(got[k].cpu().numpy(), ref[k].reshape(got[k].shape), what=what + " compute_rot")
        for a, b in zip(tj.get_euler_xyz(dev(qa)), OracleTJ.get_euler_xyz(qa)):
            H.assert_angle_close(a.cpu().numpy(), b.reshape(a.shape), what=what + " get_euler_xyz")
        tt = rng.uniform(-0.5, 1.5, (n, 1)).astype(np.float32)
        small = (rng.standard_normal((n, 3)) * rng.uniform(0.01, 0.6, (n, 1))).astype(np.float32)  # exp maps below pi
        for name, args, scale in (("slerp", (qa, qb, tt), 1.0), ("quat_to_exp_map", (qa,), np.pi),
                                  ("exp_map_to_quat", (small,), 1.0), ("calc_heading_quat_inv", (qa,), 1.0)):
            got1 = getattr(tj, name)(*[dev(a) for a in args]).cpu().numpy()
            # (a half turn, w = 0 to the last bit, is +pi or -pi times the axis: the reference's edge, left out)
            rows = np.abs(qa[:, 3]) > 1e-5 if name == "quat_to_exp_map" else slice(None)
            H.assert_close(got1[rows], oracle.rowwise(name, *args)[rows], scale=scale, what=what + " " + name)
        for name, args, scale in (("quat_diff_rad", (qa, qb), np.pi), ("calc_heading_quat", (qa,), 1.0)):
            got1 = getattr(tj, name)(*[dev(a) for a in args]).cpu().numpy()
            H.assert_close(got1, getattr(OracleTJ, name)(*args).reshape(got1.shape), scale=scale, what=what + " " + name)


def test_amp_reset_fuzz():
    """HumanoidAMP reset_idx on a list: compact lists (sorted or shuffled), masked lists with -1 holes (entry i =
    environment i), empty lists; any subset of the six state arrays given; both launch shapes of the kernel (lists
    shorter and longer than 32 768 entries); poisoned state in every third trial."""
    rng = np.random.default_rng(59 + SEED)
    for trial, n in enumerate(_sizes(rng, 32, 8 * TRIALS)):
        if trial % 16 == 7:
            n = int(rng.integers(33000, 70000))  # the long-list launch shape
        K = int(rng.choice([2, 3, 5]))
        p = synthetic.amp_params(K, bool(rng.integers(2)), bool(rng.integers(2)), 300)
        if rng.integers(2):
            p.set("key_body_ids", [int(v) for v in rng.integers(0, 15, 4)])
        t = synthetic.amp_state(n, int(rng.integers(1 << 30)), DEV, K, 301)
        style = str(rng.choice(["compact", "shuffled", "masked", "empty"]))
        frac = float(rng.choice([0.003, 0.1, 1.0]))
        flagged = np.nonzero(rng.random(n) < frac)[0].astype(np.int64)
        if style == "masked":
            ids = np.where(np.isin(np.arange(n), flagged), np.arange(n), -1).astype(np.int64)
        elif style == "empty":
            ids = np.zeros(0, np.int64)
        else:
            ids = flagged.copy()
            if style == "shuffled":
                rng.shuffle(ids)
        m = len(ids)
        widths = dict(root_pos=3, root_rot=4, dof_pos=28, root_vel=3, root_ang_vel=3, dof_vel=28)
        state_np = {}
        for k, w in widths.items():
            if rng.random() < 0.85:
                a = rng.standard_normal((m, w)).astype(np.float32)
                if k == "root_rot":
                    a /= np.maximum(np.linalg.norm(a, axis=-1, keepdims=True), 1e-9).astype(np.float32)
                if k == "dof_pos":
                    a *= np.float32(0.6)
                state_np[k] = a
        state = {k: torch.from_numpy(v).to(DEV) for k, v in state_np.items()}
        if trial % 3 == 2:
            _poison(rng, state, tuple(state), 0.003)
            _poison(rng, t, ("root_states", "dof_state", "rigid_body_state"), 0.002)
            state_np = {k: v.cpu().numpy() for k, v in state.items()}
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        what = "amp reset trial %d n %d K %d %s m %d%s" % (trial, n, K, style, m, " poisoned" if trial % 3 == 2 else "")
        ops.humanoid_amp_reset(p, t, torch.from_numpy(ids).to(DEV), state)
        torch.cuda.synchronize()
        oracle.humanoid_amp_reset(p, ref, ids, state_np)
        for k in ("reset", "terminate", "progress"):
            assert np.array_equal(t[k].cpu().numpy(), ref[k]), (k, what)
        for k in ("root_states", "dof_state"):
            assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0, (k, what)
        touched = np.zeros(n, bool)
        touched[ids[ids >= 0]] = True
        got, want = t["obs"].cpu().numpy(), ref["obs"]
        assert H.bit_equal_fraction(got[~touched], want[~touched]) == 1.0, what  # the others: not written
        root_quat = ref["root_states"][:, 3:7]
        if touched.any():
            H.assert_amp_obs_close(got[touched], want[touched], what=what, quiet=True, root_quat=root_quat[touched])
        ga, wa = t["amp_obs"].cpu().numpy(), ref["amp_obs"]
        assert H.bit_equal_fraction(ga[~touched], wa[~touched]) == 1.0, what
        for k in range(K):  # every history slot of a reset environment holds the new observation
            assert H.bit_equal_fraction(ga[touched, k], got[touched]) == 1.0, (what, k)


def test_reset_lists_and_sampling_fuzz():
    """reset_idx of Cartpole / Ant / Humanoid on random id lists (sorted, shuffled, empty) with host-given or in-kernel
    draws, and the motion library's device-side draws (plain and for flagged environments, all three state-init
    modes): everything here is integer work, Philox and IEEE arithmetic, so bit for bit."""
    rng = np.random.default_rng(61 + SEED)
    g = H.load_golden("motion")
    tabs_np = H.motion_tabs_from_golden(g)
    tables = ops.motion_tables({k: torch.from_numpy(v).to(DEV) for k, v in tabs_np.items()})
    cdf = np.cumsum(g["motion_weights"]) / np.sum(g["motion_weights"])
    cdf_dev = torch.from_numpy(cdf).to(DEV)
    for trial, n in enumerate(_sizes(rng, 64, 8 * TRIALS)):
        seed, offset = int(rng.integers(1 << 40)), int(rng.integers(1 << 20))
        m = int(n * float(rng.choice([0.0, 0.02, 0.5, 1.0])))
        ids = rng.choice(n, size=m, replace=False).astype(np.int64)
        if rng.integers(2):
            ids.sort()
        ids_dev = torch.from_numpy(ids).to(DEV)
        what = "trial %d n %d m %d" % (trial, n, m)
        with_rand = bool(rng.integers(2))
        # Cartpole
        p = synthetic.cartpole_params(seed=seed, rng_offset=offset)
        t = synthetic.cartpole_state(n, int(rng.integers(1 << 30)), DEV, with_rand)
        ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
        ops.cartpole_reset(p, t, ids_dev)
        oracle.cartpole_reset(p, ref, ids)
        for k in ("dof_state", "progress", "reset"):
            assert np.array_equal(t[k].cpu().numpy().view(np.uint8), ref[k].view(np.uint8)), ("cartpole", k, what)
        # Ant / Humanoid
        for spec, reset in ((synthetic.ANT, ops.ant_reset), (synthetic.HUMANOID, ops.humanoid_reset)):
            p = synthetic.loco_params(spec, seed=seed, rng_offset=offset)
            p.dt = float(rng.uniform(0.005, 0.05))
            p.set("targets", rng.uniform(-500, 500, 3).tolist())
            t = synthetic.loco_state(spec, n, int(rng.integers(1 << 30)), DEV, with_rand)
            if trial % 3 == 2:
                _poison(rng, t, ("root_states", "dof_state", "potentials"), 0.01)
            ref = {k: v.cpu().numpy().copy() for k, v in t.items()}
            reset(p, t, ids_dev)
            oracle.loco_reset(p, ref, ids)
            for k in ("dof_state", "potentials", "prev_potentials", "progress", "reset", "root_states", "obs"):
                assert H.bit_equal_fraction(t[k].cpu().numpy(), ref[k]) == 1.0 if ref[k].dtype == np.float32 else \
                    np.array_equal(t[k].cpu().numpy(), ref[k]), (spec["name"] if "name" in spec else "loco", k, what)
        # the motion library's draws
        random_time, env_offset = bool(rng.integers(2)), int(rng.integers(0, 1 << 33))
        mi = torch.empty(n, dtype=torch.int64, device=DEV)
        mt = torch.empty(n, dtype=torch.float64, device=DEV)
        ops.motion_sample(tables, cdf_dev, mi, mt, random_time=random_time, seed=seed, rng_offset=offset, env_offset=env_offset)
        want = oracle.motion_sample(tabs_np, cdf, n, random_time, seed=seed, rng_offset=offset, env_offset=env_offset)
        assert np.array_equal(mi.cpu().numpy(), want[0]) and np.array_equal(mt.cpu().numpy(), want[1]), ("sample", what)
        flags = (rng.random(n) < float(rng.choice([0.0, 0.05, 1.0]))).astype(np.int64)
        prob = float(rng.choice([0.0, 0.35, 1.0]))
        env_ids, dflt = torch.empty_like(mi), torch.empty_like(mi)
        ops.motion_sample_flagged(tables, cdf_dev, torch.from_numpy(flags).to(DEV), env_ids, mi, mt, random_time=random_time,
                                  seed=seed, rng_offset=offset, ref_init_prob=prob, default_env_ids=dflt)
        want = oracle.motion_sample_flagged(tabs_np, cdf, flags, random_time, seed, offset, prob)
        for a, b, name in zip((env_ids, mi, mt, dflt), want, ("env_ids", "motion_ids", "motion_times", "default_env_ids")):
            assert np.array_equal(a.cpu().numpy(), b), ("sample_flagged", name, what)


@pytest.mark.filterwarnings("ignore::RuntimeWarning")  # numpy announces the overflows and NaNs the poisoned trials are about
def test_step_wrappers_fuzz():
    """The action stage (noise, clamp, copy, scaling in its three modes), the stand-alone noise lambda and the step
    tail on random shapes (1 - 40 actions or columns, bases that are only 4-byte aligned), random gains, clips and
    noise parameters, poisoned entries in every third trial: plain IEEE arithmetic, so bit for bit."""
    rng = np.random.default_rng(67 + SEED)

    def dev(a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)

    specials = np.array([np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30, 1e-30, 1e-40, -1e-45, 0.0, -0.0], np.float32)
    for trial in range(20 * TRIALS):
        envs, width, shift = int(rng.integers(1, 3000)), int(rng.integers(1, 41)), int(rng.integers(0, 2))
        count = envs * width
        frac = 0.02 if trial % 3 == 2 else 0.0

        def arr(n, scale=1.0):
            x = (rng.standard_normal(n) * scale).astype(np.float32)
            if frac:
                idx = rng.integers(0, n, max(1, int(n * frac)))
                x[idx] = specials[rng.integers(0, len(specials), len(idx))]
            return x

        def shifted(a):  # the same values at a base one float into an allocation
            return dev(np.concatenate([np.zeros(shift, np.float32), a]))[shift:]

        what = "wrappers trial %d envs %d width %d shift %d%s" % (trial, envs, width, shift, " poisoned" if frac else "")
        a, corr, fresh = arr(count, 2.0), arr(count), arr(count)
        gain, off = arr(width, 3.0), arr(width)
        power, clip = float(rng.uniform(0.1, 500.0)), float(rng.choice([np.inf, 1.0, 0.5, 3.0]))
        operation = str(rng.choice(["additive", "scaling"]))
        nz = [float(v) for v in rng.uniform(-1, 1, 4)]
        noised = bool(rng.integers(2))
        noise_np = dict(corr=corr, fresh=fresh, corr_scale=nz[0], corr_shift=nz[1], scale=nz[2], shift=nz[3], operation=operation)
        for mode in ("force", "pd_target", "first_dof"):
            num_dof = int(rng.integers(1, 5)) if mode == "first_dof" else 0
            acts = a[:envs] if mode == "first_dof" else a
            spec = ops.NoiseSpec(dev(corr[:acts.size]), nz[0], nz[1], nz[2], nz[3], operation, "gaussian") if noised else None
            keep = torch.zeros(acts.size + shift, device=DEV)[shift:]
            tgt = torch.full((envs * num_dof if mode == "first_dof" else acts.size,), 7.0, device=DEV)
            ops.action_stage(shifted(acts), tgt, mode, gain=None if mode == "first_dof" else dev(gain),
                             offset=dev(off) if mode == "pd_target" else None, power_scale=power, clip_actions=clip,
                             actions_out=keep, noise=spec, fresh=dev(fresh[:acts.size]) if noised else None, num_dof=num_dof)
            nn = dict(noise_np, corr=corr[:acts.size], fresh=fresh[:acts.size]) if noised else None
            wa, wt = oracle.action_stage(acts if mode == "first_dof" else acts.reshape(envs, width), mode, gain=gain, offset=off,
                                         power_scale=power, clip_actions=clip, num_dof=num_dof, noise=None if nn is None else
                                         {k: (v if not isinstance(v, np.ndarray) else v.reshape(wshape(acts, mode, envs, width)))
                                          for k, v in nn.items()})
            assert H.bit_equal_fraction(keep.cpu().numpy(), wa.reshape(-1)) == 1.0, (what, mode, "actions")
            assert H.bit_equal_fraction(tgt.cpu().numpy(), wt.reshape(-1)) == 1.0, (what, mode, "targets")
        # the noise lambda on its own, and the tail
        spec = ops.NoiseSpec(dev(corr), nz[0], nz[1], nz[2], nz[3], operation, "gaussian")
        out = torch.zeros(count + shift, device=DEV)[shift:]
        ops.noise_apply(spec, shifted(a), out=out, fresh=dev(fresh))
        assert H.bit_equal_fraction(out.cpu().numpy(), oracle.noise_apply(a, corr, fresh, nz[0], nz[1], nz[2], nz[3], operation)) == 1.0, what
        obs = a.reshape(envs, width)
        limit = int(rng.integers(2, 60))
        progress, reset = rng.integers(0, 70, envs), rng.integers(0, 2, envs)
        clamped, to = torch.zeros(envs, width, device=DEV), torch.zeros(envs, dtype=torch.bool, device=DEV)
        clip_obs = float(rng.uniform(0.1, 6.0))
        ops.vec_task_tail(dev(obs), clamped, clip_obs, dev(progress), dev(reset), to, limit)
        want_obs, want_to = oracle.vec_task_tail(obs, clip_obs, progress, reset, limit)
        assert H.bit_equal_fraction(clamped.cpu().numpy(), want_obs) == 1.0, (what, "tail")
        assert np.array_equal(to.cpu().numpy(), want_to), (what, "time_outs")


def wshape(acts, mode, envs, width):
    return (envs,) if mode == "first_dof" else (envs, width)


def test_quantile_fuzz():
    """igx_quantile_f64 and its windowed form against torch.quantile (fp64, linear interpolation): random sizes on
    both sides of the one-block path, random q including 0 and 1, continuous values, heavy ties, signed zeros,
    infinities, values packed into a few ulps, subnormals, and a NaN now and then; the windowed select with one
    workspace kept across trials, as a task keeps it (its window then starts wherever the last data left it)."""
    rng = np.random.default_rng(71 + SEED)
    window_ws = torch.zeros(ops._abi.IGX_QUANTILE_WORKSPACE_BYTES // 8, dtype=torch.int64, device=DEV)
    for trial in range(40 * TRIALS):
        n = int(rng.choice([rng.integers(1, 70), rng.integers(1, 5000), rng.integers(4000, 300000)]))
        kind = int(rng.integers(0, 7))
        x = rng.standard_normal(n) * 10.0 ** rng.uniform(-3, 3)
        if kind == 1:
            x = np.round(x)                                   # ties, +0.0 and -0.0
        elif kind == 2:
            x[rng.random(n) < 0.2] = np.inf
            x[rng.random(n) < 0.1] = -np.inf
        elif kind == 3:
            x = 1.0 + rng.integers(0, 6, n) * np.spacing(1.0)   # six neighbouring doubles
        elif kind == 4:
            x = rng.integers(-3, 4, n) * 4.9e-324              # subnormals of both signs
        elif kind == 5:
            x = np.full(n, float(rng.standard_normal()))
        elif kind == 6 and rng.integers(2):
            x[int(rng.integers(0, n))] = np.nan
        q = float(rng.choice([0.0, 1.0, 0.99, 0.5, rng.random()]))
        xt = torch.from_numpy(np.ascontiguousarray(x, np.float64))
        want = float(torch.quantile(xt, q))
        for name, got in (("select", float(ops.quantile_f64(xt.to(DEV), q))),
                          ("windowed", float(ops.quantile_f64_windowed(xt.to(DEV), q, window_ws)))):
            # (a zero result may carry either sign: among tied -0.0 and +0.0 torch's sort keeps the input order, the
            # select orders by bit pattern; the two compare equal, which is what the caller's max(1.8, q) sees)
            assert got == want or (np.isnan(got) and np.isnan(want)), \
                ("quantile trial %d n %d kind %d q %r %s" % (trial, n, kind, q, name), got, want)
