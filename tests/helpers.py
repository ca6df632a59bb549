"""Shared helpers for the parity tests.

Tolerance convention used everywhere: integers / flags / indices are compared
exactly; fp32 values must satisfy |x - ref| <= 1e-5 * max(|ref|, s) where s is
the natural scale of the column (1 for unit-range observations and rewards,
pi for angles, the input's own scale for copied / scaled state).
"""
import json
import os

import numpy as np

from isaacgym_amd import _abi

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
RTOL = 1e-5


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def close(x, ref, scale=1.0, rtol=RTOL):
    """elementwise |x-ref| <= rtol*max(|ref|, scale); NaN == NaN."""
    x = np.asarray(x)
    ref = np.asarray(ref)
    tol = rtol * np.maximum(np.abs(ref), scale)
    with np.errstate(invalid="ignore"):  # inf - inf of equal infinities: caught by x == ref below
        ok = np.abs(x.astype(np.float64) - ref.astype(np.float64)) <= tol
    ok |= np.isnan(x) & np.isnan(ref)
    ok |= (x == ref)
    return ok


def assert_close(x, ref, scale=1.0, rtol=RTOL, what=""):
    ok = close(x, ref, scale, rtol)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        raise AssertionError("%s: %d/%d outside tolerance; first at %s: got %r want %r" %
                             (what, len(bad), ok.size, i, x[i], ref[i]))


def assert_angle_close(x, ref, rtol=RTOL, what="", branch_cut=None):
    """Angles at scale pi, STRICT: |x - ref| <= rtol * max(|ref|, pi) with no modulo-2-pi allowance.  The yaw /
    roll / pitch of get_euler_xyz are `atan2(..) % 2 pi`: whether the 2 pi is added hangs on the sign of the
    atan2 numerator, which is plain IEEE arithmetic and therefore the same everywhere; a result that sits 2 pi
    away from the reference's is a wrong observation and fails here.

    branch_cut="pi" is for values the reference itself sends through normalize_angle = atan2(sin a, cos a)
    (Humanoid's yaw / roll / angle_to_target): within 1e-3 of +-pi the sign of sin(a) hangs on the last bit
    of a and of the library's sin, so -pi and +pi are the same answer there, and only there.  Returns the
    number of entries that used that allowance (0 without branch_cut)."""
    x = np.asarray(x, np.float64)
    ref = np.asarray(ref, np.float64)
    d = np.abs(x - ref)
    tol = rtol * np.maximum(np.abs(ref), np.pi)
    ok = (d <= tol) | (np.isnan(x) & np.isnan(ref)) | (x == ref)  # (x == ref: equal infinities, whose difference is NaN)
    flipped = np.zeros_like(ok)
    if branch_cut == "pi":
        flipped = ~ok & (np.abs(ref) >= np.pi - 1e-3) & (np.abs(d - 2 * np.pi) <= tol)
        ok = ok | flipped
    elif branch_cut is not None:
        raise ValueError(branch_cut)
    if not ok.all():
        bad = np.argwhere(~ok)
        i = tuple(bad[0])
        wraps = int((~ok & (np.abs(d - 2 * np.pi) <= 1e-3)).sum())
        raise AssertionError("%s: %d/%d angles outside tolerance (%d of them 2 pi away: wrap errors); first at %s: got %r want %r" %
                             (what, len(bad), ok.size, wraps, i, x[i], ref[i]))
    return int(flipped.sum())


def bit_equal_fraction(x, ref):
    """Fraction of entries with identical bits; two NaNs count as identical whatever their sign and payload
    (x86 produces the negative default NaN for an invalid operation, the GPU the positive one)."""
    xf = np.ascontiguousarray(x, np.float32)
    rf = np.ascontiguousarray(ref, np.float32)
    same = (xf.view(np.uint32) == rf.view(np.uint32)) | (np.isnan(xf) & np.isnan(rf))
    return float(same.mean()) if same.size else 1.0


def cartpole_params(meta, seed=0, rng_offset=0):
    return _abi.CartpoleParams(reset_dist=meta["reset_dist"],
                               max_episode_length_m1=float(meta["max_episode_length"] - 1),
                               reward_variant=1 if meta["task"] == "CartpoleSample" else 0,
                               seed=seed, rng_offset=rng_offset)


def loco_params(meta, g, seed=0, rng_offset=0):
    p = _abi.LocoParams(
        num_dof=meta["num_dof"], num_sensor_floats=meta["num_sensor_floats"], dt=meta["dt"],
        dof_vel_scale=meta["dof_vel_scale"], contact_force_scale=meta["contact_force_scale"],
        angular_velocity_scale=meta["angular_velocity_scale"],
        up_weight=meta["up_weight"], heading_weight=meta["heading_weight"],
        actions_cost_scale=meta["actions_cost_scale"], energy_cost_scale=meta["energy_cost_scale"],
        joints_at_limit_cost_scale=meta["joints_at_limit_cost_scale"],
        termination_height=meta["termination_height"], death_cost=meta["death_cost"],
        max_episode_length_m1=float(meta["max_episode_length"] - 1),
        targets=g["targets"], inv_start_rot=g["inv_start_rot"],
        basis_vec0=[1.0, 0.0, 0.0], basis_vec1=[0.0, 0.0, 1.0],
        dof_limits_lower=g["dof_limits_lower"], dof_limits_upper=g["dof_limits_upper"],
        seed=seed, rng_offset=rng_offset)
    if "motor_efforts" in g:
        p.set("motor_efforts", g["motor_efforts"])
        p.max_motor_effort = meta["max_motor_effort"]
    return p


# observation columns that go through atan2 / remainder (library-dependent
# transcendental functions): yaw, roll, angle_to_target
LOCO_ANGLE_COLS = (7, 8, 9)


def loco_branch_cut(humanoid):
    """Ant publishes the raw `% 2 pi` angles (strict comparison); Humanoid sends the three through
    normalize_angle (humanoid.py:513-515), whose +-pi ends are one answer."""
    return "pi" if humanoid else None


ALLEGRO_VARIANTS = ["default", "full", "no_vel_asym", "pen_maxsucc", "relative", "ema", "force",
                    "nonfinite",  # NaN / +-Inf actions and simulator state in a few environments
                    "extreme",    # huge, tiny, subnormal, signed-zero and non-finite values scattered over 1 % of them
                    "nonunit"]    # object quaternions of length 0.5 ... 2


def allegro_params(meta, g, seed=0, rng_offset=0):
    ama = meta["act_moving_average"]
    return _abi.AllegroParams(
        obs_type=_abi.ALLEGRO_OBS_TYPES[meta["obs_type"]],
        max_consecutive_successes=meta["max_consecutive_successes"],
        ignore_z_rot=1 if meta["object_type"] == "pen" else 0,
        use_relative_control=1 if meta["use_relative_control"] else 0,
        dist_reward_scale=meta["dist_reward_scale"], rot_reward_scale=meta["rot_reward_scale"],
        rot_eps=meta["rot_eps"], action_penalty_scale=meta["action_penalty_scale"],
        success_tolerance=meta["success_tolerance"], success_threshold_c=meta["success_threshold_c"],
        reach_goal_bonus=meta["reach_goal_bonus"], fall_dist=meta["fall_dist"], fall_penalty=meta["fall_penalty"],
        max_episode_length_m1=float(meta["max_episode_length"] - 1),
        av_factor=meta["av_factor"], one_minus_av_factor=meta["one_minus_av_factor"],
        vel_obs_scale=0.2, force_torque_obs_scale=10.0,
        reset_position_noise=meta["reset_position_noise"], reset_dof_pos_noise=meta["reset_dof_pos_noise"],
        reset_dof_vel_noise=meta["reset_dof_vel_noise"],
        act_moving_average=ama, one_minus_act_moving_average=1.0 - ama,
        dof_speed_scale_dt=meta["dof_speed_scale_dt"], goal_displacement=[-0.2, -0.06, 0.12],
        dof_lower=g["dof_lower"], dof_upper=g["dof_upper"], seed=seed, rng_offset=rng_offset)


class ForceReplay:
    """The random-force block of a hand task's pre_physics_step (forceScale > 0 recordings): runs
    force_fn ahead of the pre-physics call of every step, on the reference's own draws, and checks
    rb_forces / random_force_prob against what the reference class left behind."""

    def __init__(self, g, force_fn, to_buf, to_np, zeros):
        from isaacgym_amd import ops
        m = g["meta"]
        self.g, self.fn, self.to_buf, self.to_np = g, force_fn, to_buf, to_np
        self.on = force_fn is not None and "force_rand" in g
        if "force_rand" in g and force_fn is None:
            raise AssertionError("this recording has forceScale > 0: pass force_fn")
        if not self.on:
            return
        N, NB = m["N"], m["num_rb_force_bodies"]
        self.rb = zeros((N, NB, 3), np.float32)
        self.prob = zeros((N,), np.float32)
        self.rand = zeros((N, 5), np.float32)
        decay, lo, hi = ops.hand_force_constants(m["force_decay"], m["dt"], m["force_decay_interval"], m["force_prob_range"])
        self.kw = dict(object_rb_handle=m["object_rb_handle"], object_rb_mass=m["object_rb_mass"],
                       force_scale=m["force_scale"], decay=decay, log_prob_lo=lo, log_prob_hi=hi)

    def _put(self, dst, arr):
        src = self.to_buf(np.ascontiguousarray(arr).reshape(tuple(dst.shape)))
        if hasattr(dst, "copy_"):
            dst.copy_(src)
        else:
            dst[...] = src

    def before_pre(self, t, reset):
        if self.on:
            self._put(self.rand, self.g["force_rand"][t])
            self.fn(self.rb, self.prob, reset, rand=self.rand, **self.kw)

    def after_pre(self, t):
        if not self.on:
            return
        what = "pre step %d " % t
        # the probability goes through exp (1e-6 class); the forces are products of the given draws: bit for bit
        assert_close(self.to_np(self.prob), self.g["pre_random_force_prob"][t], rtol=2e-6, what=what + "random_force_prob")
        assert bit_equal_fraction(self.to_np(self.rb), self.g["pre_rb_forces"][t]) == 1.0, what + "rb_forces"
        assert (self.g["pre_rb_forces"][t] != 0).any()
        self._put(self.prob, self.g["pre_random_force_prob"][t])  # continue from the reference's own values


def allegro_replay(g, pre_fn, post_fn, to_buf, to_np, zeros, force_fn=None):
    """Replay an AllegroHand golden file through pre_fn / post_fn.

    to_buf(array) -> backend buffer, to_np(buffer) -> numpy, zeros(shape, dtype)
    -> backend buffer.  Asserts parity at every step."""
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = allegro_params(m, g)
    nobs = _abi.ALLEGRO_NUM_OBS[p.obs_type]
    asym = bool(m["asymmetric"])
    b = dict(dof_state=zeros((N, 16, 2), np.float32), dof_force=zeros((N, 16), np.float32),
             root_state_tensor=to_buf(g["root0"]), object_indices=to_buf(g["object_indices"]),
             goal_object_indices=to_buf(g["goal_object_indices"]), goal_states=to_buf(g["goal_states0"]),
             goal_init_state=to_buf(g["goal_init_state"]), object_init_state=to_buf(g["object_init_state"]),
             actions=zeros((N, 16), np.float32), prev_targets=zeros((N, 16), np.float32),
             cur_targets=zeros((N, 16), np.float32), rand=zeros((N, 45), np.float32),
             obs=zeros((N, nobs), np.float32), states=zeros((N, 88), np.float32) if asym else None,
             rew=zeros((N,), np.float32), reset=to_buf(np.ones(N, np.int64)), reset_goal=to_buf(np.ones(N, np.int64)),
             progress=zeros((N,), np.int64), randomize=zeros((N,), np.int64), successes=zeros((N,), np.float32),
             consecutive_successes=zeros((1,), np.float32), workspace=zeros((4 * ((N + 15) // 16 + 1),), np.float32))

    def put(name, arr):
        dst = b[name]
        src = to_buf(np.ascontiguousarray(arr).reshape(tuple(dst.shape)))
        if hasattr(dst, "copy_"):
            dst.copy_(src)
        else:
            dst[...] = src

    forces = ForceReplay(g, force_fn, to_buf, to_np, zeros)
    rot_cols = [3, 4, 5, 6]
    for t in range(T):
        if t == 1:
            prog = to_np(b["progress"]).copy()
            prog[:4] = [596, 597, 598, 599]
            put("progress", prog)
        put("actions", g["actions"][t])
        put("rand", g["rand"][t])
        forces.before_pre(t, b["reset"])
        pre_fn(p, b, N)
        forces.after_pre(t)
        what = "pre step %d " % t
        for k, ref in (("reset", g["pre_reset"][t]), ("reset_goal", g["pre_reset_goal"][t]),
                       ("progress", g["pre_progress"][t])):
            assert np.array_equal(to_np(b[k]), ref), what + k
        for k, ref in (("dof_state", g["pre_dof_state"][t]), ("prev_targets", g["pre_prev_targets"][t]),
                       ("cur_targets", g["pre_cur_targets"][t]), ("successes", g["pre_successes"][t])):
            assert bit_equal_fraction(to_np(b[k]).reshape(ref.shape), ref) == 1.0, what + k
        # rows: positions / velocities are exact arithmetic, rotations go through sin/cos
        for k, ref in (("root_state_tensor", g["pre_root"][t]), ("goal_states", g["pre_goal_states"][t])):
            got = to_np(b[k])
            nonrot = [c for c in range(13) if c not in rot_cols]
            assert bit_equal_fraction(got[:, nonrot], ref[:, nonrot]) == 1.0, what + k
            assert_close(got[:, rot_cols], ref[:, rot_cols], what=what + k + " rot")
        # continue from the reference's own post-reset rotations so later steps do not
        # accumulate library-level differences
        put("goal_states", g["pre_goal_states"][t])
        put("dof_state", g["in_dof_state"][t])
        put("dof_force", g["in_dof_force"][t])
        put("root_state_tensor", g["in_root"][t])
        post_fn(p, b, N)
        what = "post step %d " % t
        for k in ("reset", "reset_goal", "progress", "randomize"):
            assert np.array_equal(to_np(b[k]), g[k][t]), what + k
        assert bit_equal_fraction(to_np(b["successes"]), g["successes"][t]) == 1.0, what + "successes"
        # observations are copies, scalings and quaternion products: exact arithmetic, bit for bit
        assert bit_equal_fraction(to_np(b["obs"]), g["obs"][t]) == 1.0, what + "obs"
        if asym:
            assert bit_equal_fraction(to_np(b["states"]), g["states"][t]) == 1.0, what + "states"
        # reward: -10*dist + 1/(|rot_dist|+0.1) (+250 bonus): natural scale 1
        assert_close(to_np(b["rew"]), g["rew"][t], what=what + "rew")
        assert_close(to_np(b["consecutive_successes"]), g["consecutive_successes"][t], what=what + "cons")


SHADOW_VARIANTS = ["default", "openai_asym", "full", "no_vel_pen", "force", "nonfinite", "extreme", "nonunit"]


def shadow_params(meta, g, seed=0, rng_offset=0):
    ama = meta["act_moving_average"]
    return _abi.ShadowParams(
        obs_type=_abi.IGX_SHADOW_OBS[meta["obs_type"]],
        max_consecutive_successes=meta["max_consecutive_successes"],
        ignore_z_rot=1 if meta["object_type"] == "pen" else 0,
        use_relative_control=1 if meta["use_relative_control"] else 0,
        dist_reward_scale=meta["dist_reward_scale"], rot_reward_scale=meta["rot_reward_scale"],
        rot_eps=meta["rot_eps"], action_penalty_scale=meta["action_penalty_scale"],
        success_tolerance=meta["success_tolerance"], success_threshold_c=meta["success_threshold_c"],
        reach_goal_bonus=meta["reach_goal_bonus"], fall_dist=meta["fall_dist"], fall_penalty=meta["fall_penalty"],
        max_episode_length_m1=float(meta["max_episode_length"] - 1),
        av_factor=meta["av_factor"], one_minus_av_factor=meta["one_minus_av_factor"],
        vel_obs_scale=0.2, force_torque_obs_scale=10.0,
        reset_position_noise=meta["reset_position_noise"], reset_dof_pos_noise=meta["reset_dof_pos_noise"],
        reset_dof_vel_noise=meta["reset_dof_vel_noise"],
        act_moving_average=ama, one_minus_act_moving_average=1.0 - ama,
        dof_speed_scale_dt=meta["dof_speed_scale_dt"], goal_displacement=[-0.2, -0.06, 0.12],
        num_bodies=meta["num_bodies"], dof_lower=g["dof_lower"], dof_upper=g["dof_upper"],
        actuated_dof_indices=[int(x) for x in g["actuated_dof_indices"]],
        fingertip_handles=[int(x) for x in g["fingertip_handles"]], seed=seed, rng_offset=rng_offset)


def shadow_replay(g, pre_fn, post_fn, to_buf, to_np, zeros, force_fn=None):
    """Replay a ShadowHand golden file through pre_fn / post_fn (see allegro_replay)."""
    m = g["meta"]
    N, T, NB = m["N"], m["T"], m["num_bodies"]
    p = shadow_params(m, g)
    nobs = _abi.IGX_SHADOW_NUM_OBS[m["obs_type"]]
    asym = bool(m["asymmetric"])
    b = dict(dof_state=zeros((N, 24, 2), np.float32), dof_force=zeros((N, 24), np.float32),
             vec_sensor_tensor=zeros((N, 30), np.float32), rigid_body_states=zeros((N, NB, 13), np.float32),
             root_state_tensor=to_buf(g["root0"]), object_indices=to_buf(g["object_indices"]),
             goal_object_indices=to_buf(g["goal_object_indices"]), goal_states=to_buf(g["goal_states0"]),
             goal_init_state=to_buf(g["goal_init_state"]), object_init_state=to_buf(g["object_init_state"]),
             actions=zeros((N, 20), np.float32), prev_targets=zeros((N, 24), np.float32),
             cur_targets=zeros((N, 24), np.float32), rand=zeros((N, 61), np.float32),
             obs=zeros((N, nobs), np.float32), states=zeros((N, 211), np.float32) if asym else None,
             rew=zeros((N,), np.float32), reset=to_buf(np.ones(N, np.int64)), reset_goal=to_buf(np.ones(N, np.int64)),
             progress=zeros((N,), np.int64), randomize=zeros((N,), np.int64), successes=zeros((N,), np.float32),
             consecutive_successes=zeros((1,), np.float32), workspace=zeros((4 * ((N + 15) // 16 + 1),), np.float32))

    def put(name, arr):
        dst = b[name]
        src = to_buf(np.ascontiguousarray(arr).reshape(tuple(dst.shape)))
        if hasattr(dst, "copy_"):
            dst.copy_(src)
        else:
            dst[...] = src

    forces = ForceReplay(g, force_fn, to_buf, to_np, zeros)
    rot_cols = [3, 4, 5, 6]
    L = m["max_episode_length"]
    for t in range(T):
        if t == 1:
            prog = to_np(b["progress"]).copy()
            prog[:4] = [L - 4, L - 3, L - 2, L - 1]
            put("progress", prog)
        put("actions", g["actions"][t])
        put("rand", g["rand"][t])
        forces.before_pre(t, b["reset"])
        pre_fn(p, b, N)
        forces.after_pre(t)
        what = "pre step %d " % t
        for k, ref in (("reset", g["pre_reset"][t]), ("reset_goal", g["pre_reset_goal"][t]),
                       ("progress", g["pre_progress"][t])):
            assert np.array_equal(to_np(b[k]), ref), what + k
        for k, ref in (("dof_state", g["pre_dof_state"][t]), ("prev_targets", g["pre_prev_targets"][t]),
                       ("cur_targets", g["pre_cur_targets"][t]), ("successes", g["pre_successes"][t])):
            assert bit_equal_fraction(to_np(b[k]).reshape(ref.shape), ref) == 1.0, what + k
        for k, ref in (("root_state_tensor", g["pre_root"][t]), ("goal_states", g["pre_goal_states"][t])):
            got = to_np(b[k])
            nonrot = [c for c in range(13) if c not in rot_cols]
            assert bit_equal_fraction(got[:, nonrot], ref[:, nonrot]) == 1.0, what + k
            assert_close(got[:, rot_cols], ref[:, rot_cols], what=what + k + " rot")
        put("goal_states", g["pre_goal_states"][t])
        put("dof_state", g["in_dof_state"][t])
        put("dof_force", g["in_dof_force"][t])
        put("vec_sensor_tensor", g["in_sensors"][t])
        put("rigid_body_states", g["in_bodies"][t])
        put("root_state_tensor", g["in_root"][t])
        post_fn(p, b, N)
        what = "post step %d " % t
        for k in ("reset", "reset_goal", "progress", "randomize"):
            assert np.array_equal(to_np(b[k]), g[k][t]), what + k
        assert bit_equal_fraction(to_np(b["successes"]), g["successes"][t]) == 1.0, what + "successes"
        # observations are copies, scalings and quaternion products: exact arithmetic, bit for bit
        assert bit_equal_fraction(to_np(b["obs"]), g["obs"][t]) == 1.0, what + "obs"
        if asym:
            assert bit_equal_fraction(to_np(b["states"]), g["states"][t]) == 1.0, what + "states"
        assert_close(to_np(b["rew"]), g["rew"][t], what=what + "rew")
        assert_close(to_np(b["consecutive_successes"]), g["consecutive_successes"][t], what=what + "cons")


# ---- HumanoidAMP observation columns (humanoid_amp_base.py:542-576, dof_to_obs :510-540) -----------------
AMP_DOF_OFFSETS = [0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28]
# (name, first column, one past the last, how the natural scale of the group is found)
#   "exact": copies of inputs, bit for bit;  "unit": tan-norm / unit-range columns, scale 1;
#   "row": rotated input vectors (root velocities, key-body offsets): scale = the largest vector norm of the row
AMP_OBS_GROUPS = [("root_h", 0, 1, "exact"), ("root_rot tan-norm", 1, 7, "unit"), ("local root vel", 7, 10, "row"),
                  ("local root ang vel", 10, 13, "row"), ("dof_obs", 13, 65, "unit"), ("dof_vel", 65, 93, "exact"),
                  ("key body pos", 93, 105, "row")]


def amp_joint_columns(kind):
    """Per joint of DOF_OFFSETS the columns it owns: kind "dof_pos" -> in the 28-wide joint vector, "obs" -> in the
    105-wide observation (three-DOF joints own six tan-norm columns of dof_obs, one-DOF joints one)."""
    cols, o = [], 13
    for a, b in zip(AMP_DOF_OFFSETS[:-1], AMP_DOF_OFFSETS[1:]):
        if kind == "dof_pos":
            cols.append(list(range(a, b)))
        else:
            w = 6 if b - a == 3 else 1
            cols.append(list(range(o, o + w)))
            o += w
    return cols


def amp_obs_errors(x, ref, copies_exact=True, root_quat=None):
    """Per column group of an [n,105] AMP observation: max of |x - ref| / max(|ref|, natural scale), as an array
    over the entries (so a caller may exempt entries) plus the group table.  copies_exact=False holds the copied
    columns to the tolerance too (scale 1): for motion tables that were not taken from the recording but rebuilt by
    the product's loader on this machine's host CPU, whose torch acos / sin may differ in the last bit.
    root_quat [n,4]: the root quaternions the observation was built from; a row whose quaternion is far from unit
    length (poisoned state) has its root tan-norm columns judged at the scale of that quaternion's largest finite
    component (squared): they are products of its components with heading-frame factors that may be as small as
    1e-3, so the size of the outputs says nothing about the size of their rounding errors."""
    x = np.asarray(x, np.float64).reshape(-1, 105)
    ref = np.asarray(ref, np.float64).reshape(-1, 105)
    err = np.zeros_like(ref)
    same = (x == ref) | (np.isnan(x) & np.isnan(ref))  # equal values (infinities included) and NaN where the reference has NaN
    with np.errstate(invalid="ignore", over="ignore"):
        for name, a, b, kind in AMP_OBS_GROUPS:
            d = np.abs(x[:, a:b] - ref[:, a:b])
            if kind == "exact" and copies_exact:
                err[:, a:b] = np.where(same[:, a:b], 0.0, np.inf)
                continue
            scale = 1.0
            if kind == "unit":
                # tan-norm columns are products of quaternion components: of size 1 for the unit quaternions of a
                # simulator, of the size of |q|^2 for anything else (a poisoned state), which is then their scale
                v = np.where(np.isfinite(ref[:, a:b]), np.abs(ref[:, a:b]), 0.0)
                scale = np.maximum(1.0, v.max(-1, keepdims=True))
                if root_quat is not None and a == 1:
                    q = np.asarray(root_quat, np.float64).reshape(-1, 4)
                    q = np.where(np.isfinite(q), np.abs(q), 0.0).max(-1, keepdims=True)
                    scale = np.maximum(scale, q * q)
            if kind == "row":
                v = np.where(np.isfinite(ref[:, a:b]), ref[:, a:b], 0.0).reshape(len(ref), -1, 3)
                scale = np.sqrt((v * v).sum(-1)).max(-1, keepdims=True)
            e = d / np.maximum(np.abs(ref[:, a:b]), scale)
            err[:, a:b] = np.where(same[:, a:b], 0.0, np.where(np.isfinite(e), e, np.inf))
    return err


def assert_amp_obs_close(x, ref, what="", rtol=RTOL, exempt=None, quiet=False, copies_exact=True, root_quat=None):
    """The 1e-5 bar per column group, each group at its own natural scale; prints the largest error of every group
    (in units of the tolerance' denominator).  `exempt`: boolean [n,105] of entries left out (knife-edge joints)."""
    err = amp_obs_errors(x, ref, copies_exact, root_quat)
    if exempt is not None:
        err = np.where(exempt.reshape(err.shape), 0.0, err)
    worst = {name: (float(err[:, a:b].max()) if err.size else 0.0) for name, a, b, _ in AMP_OBS_GROUPS}
    if not quiet:
        print("%s max error per group (tolerance %.0e): %s" % (what, rtol, ", ".join("%s %.2e" % kv for kv in worst.items())))
    bad = {k: v for k, v in worst.items() if not v <= rtol}
    assert not bad, "%s: groups outside %.0e: %s" % (what, rtol, bad)
    return worst


def knife_edge_joints(got_dof_pos, want_dof_pos):
    """quat_to_angle_axis masks a rotation to zero when sin(theta/2) = sqrt(1 - w*w) <= 1e-5, which in fp32 means
    w == 1.0 exactly; one ulp below, the exp map is ~2 xyz (torch_jit_utils.py:423-444).  For an interpolated
    rotation within ~1e-2 rad of identity, which side w lands on hangs on the last bit of the library's acos /
    sin inside slerp; the reference has the same knife edge.  Returns (disagree [n,J] bool: the joint's rotation
    is masked - all its components exactly 0 - on exactly one side, masked_want [n,J])."""
    cols = amp_joint_columns("dof_pos")
    g = np.stack([(got_dof_pos[:, c] == 0).all(-1) for c in cols], -1)
    w = np.stack([(want_dof_pos[:, c] == 0).all(-1) for c in cols], -1)
    return g ^ w, w


def half_turn_joints(got_dof_pos, want_dof_pos, ulps=4):
    """exp_map_to_angle_axis keeps a rotation only if normalize_angle(|exp map|) > 1e-5 (no absolute value:
    torch_jit_utils.py:570-587), so a three-DOF joint turned by pi comes out of dof_to_obs as that half turn when the
    norm of its exp map rounds to pi or below, and as the identity when it rounds one ulp above (atan2 then gives
    -pi).  quat_to_exp_map produces exactly such norms for w ~ 0; which side one lands on hangs on the last bit of
    the library's acos.  Returns [n,J] bool: three-DOF joints whose exp-map norm is within `ulps` fp32 ulps of pi on
    either side of the comparison."""
    cols = amp_joint_columns("dof_pos")
    tol = ulps * float(np.spacing(np.float32(np.pi)))

    def near_pi(d):
        return np.stack([(np.abs(np.linalg.norm(d[:, c].astype(np.float64), axis=-1) - np.pi) <= tol) if len(c) == 3
                         else np.zeros(d.shape[0], bool) for c in cols], -1)
    return near_pi(got_dof_pos) | near_pi(want_dof_pos)


def assert_close_near_identity(x, ref, scale=1.0, rtol=RTOL, quantum=2e-2, what="", disagree=None, kind="dof_pos",
                               masked_want=None):
    """assert_close for quantities derived from quat_to_angle_axis(slerp(...)): every entry outside rtol must belong
    to a joint that `knife_edge_joints` found on the knife edge (masked on one side only), must stay below `quantum`
    rad, and there must not be more such joints than the checker itself masks on the same inputs (the population
    that sits on the edge).  The counts are printed.  `kind`: which array x is ("dof_pos" [n,28] or "obs" [n,105])."""
    if kind == "obs":
        bad = amp_obs_errors(x, ref) > rtol
    else:
        bad = ~close(x, ref, scale, rtol)
    n_dis = 0 if disagree is None else int(disagree.sum())
    n_masked = 0 if masked_want is None else int(masked_want.sum())
    print("%s: %d/%d entries outside %.0e; joints masked on one side only: %d; joints the checker masks: %d" %
          (what, int(bad.sum()), bad.size, rtol, n_dis, n_masked))
    if not bad.any():
        return
    assert disagree is not None, "%s: %d entries outside tolerance" % (what, bad.sum())
    allowed = np.zeros(bad.shape, bool)
    for j, cols in enumerate(amp_joint_columns(kind)):
        allowed[:, cols] |= disagree[:, j:j + 1]
    stray = bad & ~allowed
    assert not stray.any(), "%s: %d entries outside tolerance that are not on the identity knife edge; first %s" % (
        what, stray.sum(), tuple(np.argwhere(stray)[0]))
    d = np.abs(np.asarray(x, np.float64) - ref)[bad]
    assert (d < quantum).all(), "%s: max jump %g" % (what, d.max())
    assert n_dis <= max(8, n_masked), "%s: %d joints disagree, the checker masks only %d" % (what, n_dis, n_masked)


def motion_tabs_from_golden(g):
    """igx_motion_tables (numpy) from the reference's per-frame tables in motion.npz."""
    M = len(g["motion_lengths"])
    cat = lambda key, f: np.ascontiguousarray(np.concatenate([f(g["m%d_%s" % (i, key)]) for i in range(M)]),  # noqa: E731
                                               np.float32)
    kb = g["meta"]["key_body_ids"]
    nf = g["motion_num_frames"].astype(np.int64)
    return dict(
        motion_length=np.ascontiguousarray(g["motion_lengths"], np.float64),
        motion_dt=np.ascontiguousarray(g["motion_dt"], np.float64),
        motion_num_frames=nf,
        frame_offset=np.concatenate([[0], np.cumsum(nf)[:-1]]).astype(np.int64),
        root_pos=cat("global_translation", lambda a: a[:, 0]),
        root_rot=cat("global_rotation", lambda a: a[:, 0]),
        local_rot=cat("local_rotation", lambda a: a),
        root_vel=cat("global_velocity", lambda a: a[:, 0]),
        root_ang_vel=cat("global_angular_velocity", lambda a: a[:, 0]),
        key_pos=cat("global_translation", lambda a: a[:, kb]),
        dof_vel=cat("dof_vels", lambda a: a))


def check_motion_state(got, g, what=""):
    """got: dict of numpy arrays named like igx_motion_state_out."""
    # frame indices and the fp64 blend (rounded to fp32) are exact
    assert np.array_equal(got["frame_idx0"], g["frame_idx0"]), what + "frame_idx0"
    assert np.array_equal(got["frame_idx1"], g["frame_idx1"]), what + "frame_idx1"
    assert np.array_equal(got["blend"], g["blend"].astype(np.float32)), what + "blend"
    # gathers and lerps are plain IEEE arithmetic: exact
    for k in ("root_pos", "root_vel", "root_ang_vel", "dof_vel", "key_pos"):
        assert bit_equal_fraction(got[k].reshape(g[k].shape), g[k]) == 1.0, what + k
    # slerp / quaternion -> dof go through acos / sin / atan2: 1e-5 class
    assert_close(got["root_rot"], g["root_rot"], what=what + "root_rot")
    assert_close(got["dof_pos"], g["dof_pos"], scale=np.pi, what=what + "dof_pos")


def amp_params(meta):
    return _abi.AmpParams(num_amp_obs_steps=meta["K"], local_root_obs=int(meta["local_root_obs"]),
                          enable_early_termination=int(meta["enable_early_termination"]), num_contact_bodies=2,
                          contact_body_ids=[11, 14, 0, 0], key_body_ids=[5, 8, 11, 14],
                          termination_height=meta["termination_height"],
                          max_episode_length_m1=float(meta["max_episode_length"] - 1))


def amp_replay(g, fns, to_buf, to_np, zeros):
    """Replay amp.npz.  fns: dict with step(p, bufs, n), reset(p, bufs, env_ids, state),
    motion_state(ids, times) -> dict of backend buffers, motion_amp_obs(ids, times, lro, out, rows, stride, off)."""
    m = g["meta"]
    N, K, T = m["N"], m["K"], m["T"]
    p = amp_params(m)
    b = dict(root_states=zeros((N, 13), np.float32), dof_state=zeros((N, 28, 2), np.float32),
             rigid_body_state=zeros((N, 15, 13), np.float32), contact_forces=zeros((N, 15, 3), np.float32),
             obs=zeros((N, 105), np.float32), amp_obs=zeros((N, K, 105), np.float32), rew=zeros((N,), np.float32),
             reset=to_buf(np.ones(N, np.int64)), terminate=to_buf(np.ones(N, np.int64)),
             progress=zeros((N,), np.int64))

    def put(name, arr):
        dst = b[name]
        src = to_buf(np.ascontiguousarray(arr).reshape(tuple(dst.shape)))
        if hasattr(dst, "copy_"):
            dst.copy_(src)
        else:
            dst[...] = src

    for t in range(T):
        if t == 1:
            prog = to_np(b["progress"]).copy()
            prog[:4] = [296, 297, 298, 299]
            put("progress", prog)
        if t == m["reset_before_step"]:
            env_ids = g["reset_env_ids"]
            assert np.array_equal(np.nonzero(to_np(b["reset"]))[0], env_ids)
            ids, times = g["reset_motion_ids"], g["reset_motion_times"]
            state = fns["motion_state"](ids, times)
            fns["reset"](p, b, env_ids, state)
            # history slots from the motion at t - dt * (k + 1)   (humanoid_amp.py:279-294)
            hist_ids = np.tile(np.expand_dims(ids, -1), [1, K - 1]).flatten()
            hist_t = (np.expand_dims(times, -1) + (-m["dt"] * (np.arange(0, K - 1) + 1))).flatten()
            rows = (np.expand_dims(env_ids, -1) * K + 1 + np.arange(0, K - 1)).flatten().astype(np.int64)
            fns["motion_amp_obs"](hist_ids, hist_t, p.local_root_obs, b["amp_obs"], rows, 105, 0)
            for k in ("progress", "reset", "terminate"):
                assert np.array_equal(to_np(b[k]), g["reset_after_" + k]), "after reset: " + k
            root = to_np(b["root_states"])
            assert_close(root, g["reset_after_root"], what="after reset: root_states")
            assert_close(to_np(b["dof_state"]).reshape(-1, 2), g["reset_after_dof_state"], scale=np.pi,
                         what="after reset: dof_state")
            # continue from the reference's own state so the comparison of the derived
            # observations is not conditioned on library-level differences in slerp
            put("root_states", g["reset_after_root"])
            put("dof_state", g["reset_after_dof_state"])
            obs = to_np(b["obs"])
            mask = np.zeros(N, bool)
            mask[env_ids] = True
            assert_amp_obs_close(obs[mask], g["reset_after_obs"][mask], what="after reset: obs")
            assert_amp_obs_close(to_np(b["amp_obs"])[mask], g["reset_after_amp_obs"][mask], what="after reset: amp_obs")
            put("obs", g["reset_after_obs"])
            put("amp_obs", g["reset_after_amp_obs"])
        put("root_states", g["in_root"][t])
        put("dof_state", g["in_dof_state"][t])
        put("rigid_body_state", g["in_rigid_body_state"][t])
        put("contact_forces", g["in_contact_forces"][t])
        fns["step"](p, b, N)
        what = "step %d " % t
        for k in ("reset", "terminate", "progress"):
            assert np.array_equal(to_np(b[k]), g[k][t]), what + k
        assert np.array_equal(to_np(b["rew"]), g["rew"][t]), what + "rew"
        assert_amp_obs_close(to_np(b["obs"]), g["obs"][t], what=what + "obs")
        amp = to_np(b["amp_obs"]).reshape(N, K * 105)
        assert_amp_obs_close(amp[:, :105], g["amp_obs"][t][:, :105], what=what + "amp_obs current", quiet=True)
        if t != m["reset_before_step"] and t > 0:
            # history = previous step's current slot, bit for bit
            assert bit_equal_fraction(amp[:, 105:210], prev_curr) == 1.0, what + "amp_obs history"
        prev_curr = amp[:, :105].copy()


def write_golden_motions(g, dirpath):
    """Write motion.npz's raw motions as plain-array .npz motion files plus the
    .yaml motion set naming them; returns the yaml path."""
    import os
    M = len(g["motion_lengths"])
    lines = ["motions:"]
    for i in range(M):
        np.savez(os.path.join(dirpath, "m%d.npz" % i), rotation=g["m%d_rotation" % i],
                 root_translation=g["m%d_root_translation" % i], global_velocity=g["m%d_global_velocity" % i],
                 global_angular_velocity=g["m%d_global_angular_velocity" % i], parent_indices=g["parent_indices"],
                 local_translation=g["local_translation"], fps=np.float64(g["motion_fps"][i]), is_local=np.bool_(True))
        lines.append('  - file: "m%d.npz"' % i)
        lines.append("    weight: %r" % float(g["meta"]["specs"][i][2]))
    path = os.path.join(dirpath, "set.yaml")
    with open(path, "w") as f:
        f.write("\n".join(lines) + "\n")
    return path


def ant_imitate_replay(g, step_fn, to_buf, to_np, zeros):
    """Replay ant_imitate.npz.  step_fn(params, bufs, arrays, meta) runs one post_physics_step on the
    backend buffers (bufs: igx_loco_buffers names; arrays: motion / target_pose_idx / pose_diff)."""
    m = g["meta"]
    N, T = m["N"], m["T"]
    p = loco_params(m, g)
    progress, pot = to_buf(g["progress0"]), to_buf(g["potentials0"])
    randomize = zeros((N,), np.int64)
    arrays = dict(motion=to_buf(g["motion"]), target_pose_idx=to_buf(g["target_pose_idx0"]),
                  pose_diff=zeros((N,), np.float64))
    init_root, init_dof = to_buf(g["initial_root_states"]), to_buf(g["initial_dof_pos"])
    nonangle = [c for c in range(60) if c not in LOCO_ANGLE_COLS]
    for t in range(T):
        ds = to_buf(g["in_dof_state"][t])
        bufs = dict(root_states=to_buf(g["in_root"][t]), dof_state=ds, sensors=to_buf(g["in_sensors"][t]),
                    actions=to_buf(g["in_actions"][t]), initial_root_states=init_root, initial_dof_pos=init_dof,
                    rand=to_buf(g["rand"][t]), obs=zeros((N, 60), np.float32), potentials=pot,
                    prev_potentials=zeros((N,), np.float32), up_vec=zeros((N, 3), np.float32),
                    heading_vec=zeros((N, 3), np.float32), rew=zeros((N,), np.float32),
                    reset=to_buf(g["in_reset"][t]), progress=progress, randomize=randomize)
        qv = step_fn(p, bufs, arrays, m)
        what = "step %d " % t
        assert np.array_equal(to_np(arrays["pose_diff"]).view(np.uint64), g["pose_diff"][t].view(np.uint64)), what + "pose_diff"
        assert max(m["distance_floor"], float(qv)) == float(g["threshold"][t]), what + "threshold"
        for k, ref in (("reset", "reset"), ("progress", "progress"), ("randomize", "randomize")):
            assert np.array_equal(to_np(bufs[k]), g[ref][t]), what + k
        assert np.array_equal(to_np(arrays["target_pose_idx"]), g["target_pose_idx"][t]), what + "target_pose_idx"
        obs = to_np(bufs["obs"])
        assert bit_equal_fraction(obs[:, nonangle], g["obs"][t][:, nonangle]) == 1.0, what + "obs"
        assert_angle_close(obs[:, list(LOCO_ANGLE_COLS)], g["obs"][t][:, list(LOCO_ANGLE_COLS)], what=what + "angles")
        assert_close(to_np(bufs["rew"]), g["rew"][t], what=what + "rew")  # exp() in fp64, rounded to fp32
        assert bit_equal_fraction(to_np(ds), g["out_dof_state"][t]) == 1.0, what + "dof_state"
        assert bit_equal_fraction(to_np(pot), g["potentials"][t]) == 1.0, what + "potentials"


def him_params(meta, g, seed=0, rng_offset=0):
    return _abi.HumanoidImitateParams(
        dt=meta["dt"], dof_vel_scale=meta["dof_vel_scale"], up_weight=meta["up_weight"],
        actions_cost_scale=meta["actions_cost_scale"], termination_height=meta["termination_height"],
        termination_radius_sq=float(np.float32(meta["termination_radius"] ** 2)),
        max_episode_length_m1=float(meta["max_episode_length"] - 1), targets=[1000.0, 0.0, 0.0],
        dof_limits_lower=g["dof_limits_lower"], dof_limits_upper=g["dof_limits_upper"], seed=seed, rng_offset=rng_offset)


def him_replay(g, step_fn, to_buf, to_np, zeros):
    """Replay humanoid_imitate.npz through step_fn(params, bufs, n)."""
    m = g["meta"]
    N, T, D = m["N"], m["T"], m["num_dof"]
    p = him_params(m, g)
    reset, progress = to_buf(g["reset0"]), to_buf(g["progress0"])
    pot, prev = to_buf(g["potentials0"]), to_buf(g["potentials0"])
    randomize = zeros((N,), np.int64)
    frame = to_buf(np.ascontiguousarray(g["motion"][m["pose_frame"]]))
    init_root, init_dof = to_buf(g["initial_root_states"]), to_buf(g["initial_dof_pos"])
    for t in range(T):
        ds = to_buf(g["in_dof_state"][t].reshape(N, D, 2))
        bufs = dict(root_states=to_buf(g["in_root"][t]), dof_state=ds,
                    rigid_body_states=to_buf(g["in_rigid_body_states"][t].reshape(N, 22, 13)),
                    actions=to_buf(g["in_actions"][t]), initial_root_states=init_root, initial_dof_pos=init_dof,
                    rand=to_buf(g["rand"][t]), motion_frame=frame, obs=zeros((N, 205), np.float32), potentials=pot,
                    prev_potentials=prev, rew=zeros((N,), np.float32), reset=reset, progress=progress,
                    randomize=randomize)
        step_fn(p, bufs, N)
        what = "step %d " % t
        for k in ("reset", "progress", "randomize"):
            assert np.array_equal(to_np(bufs[k]), g[k][t]), what + k
        # observations are copies and elementwise IEEE arithmetic: exact
        assert bit_equal_fraction(to_np(bufs["obs"]), g["obs"][t]) == 1.0, what + "obs"
        assert bit_equal_fraction(to_np(ds).reshape(-1, 2), g["out_dof_state"][t]) == 1.0, what + "dof_state"
        assert bit_equal_fraction(to_np(pot), g["potentials"][t]) == 1.0, what + "potentials"
        assert bit_equal_fraction(to_np(prev), g["prev_potentials"][t]) == 1.0, what + "prev_potentials"
        assert_close(to_np(bufs["rew"]), g["rew"][t], what=what + "rew")  # sums of 63 / 66 terms, exp


# ---- episode statistics: the recording of the reference's agent loop + observer (observer.npz) ------
class RecordingWriter:
    def __init__(self):
        self.rows = []

    def add_scalar(self, tag, value, step):
        self.rows.append((tag, float(value), float(step)))


def observer_replay(g, observer, stats_step, to_tensor, to_np, sum_rtol=0.0):
    """Feed the recorded steps to `observer` (an isaacgym_amd RLGPUAlgoObserver, after_init not yet called)
    and to `stats_step(rew, dones) -> (n_done, sum_reward, sum_length, current_rewards, current_lengths)`;
    compare with what the reference's CommonAgent.play_steps + RLGPUAlgoObserver left behind: the
    per-environment accumulators bit for bit after every step, the values game_rewards / game_lengths were
    given, and every writer.add_scalar row of every epoch."""
    import types
    import torch
    m = g["meta"]
    N, Hz, EPOCHS, G = m["N"], m["H"], m["EPOCHS"], m["G"]
    algo = types.SimpleNamespace(games_to_track=G, device=to_tensor(np.zeros(1, np.float32)).device, writer=RecordingWriter())
    observer.after_init(algo)
    t = 0
    for epoch in range(EPOCHS):
        for _ in range(Hz):
            dones = to_tensor(g["dones"][t])
            infos = {"terminate": to_tensor(np.zeros(N, np.float32)), "consecutive_successes": to_tensor(g["cons"][t]),
                     "episode": {"level": to_tensor(g["ep_level"][t]), "bonus": to_tensor(g["ep_bonus"][t])},
                     "episode_cumulative": {"successes": to_tensor(g["cum_successes"][t]),
                                            "energy": to_tensor(g["cum_energy"][t])},
                     "scalars": {"lr": 0.25 * (epoch + 1)}, "tensor_info": to_tensor(np.zeros(N, np.float32))}
            n, sr, sl, cr, cl = stats_step(to_tensor(g["rew"][t]), dones)
            observer.process_infos_flags(infos, dones)
            what = "step %d " % t
            assert n == int(g["n_done"][t]), what + "n_done"
            # the finished episodes' rewards: exact values, summed here in double
            # (the device adds them in another order: sum_rtol)
            assert abs(sr - float(g["sum_reward"][t])) <= sum_rtol * max(1.0, abs(float(g["sum_reward"][t]))), what + "sum_reward"
            assert sl == float(g["sum_length"][t]), what + "sum_length"
            for key in ("successes", "energy"):
                got = to_np(observer.episode_cumulative[key])
                assert np.array_equal(got.view(np.uint32), g["acc_" + key][t].view(np.uint32)), what + "accumulator " + key
            t += 1
        assert np.array_equal(to_np(cr).view(np.uint32), g["current_rewards"][epoch].view(np.uint32)), "current_rewards"
        assert np.array_equal(to_np(cl).view(np.uint32), g["current_lengths"][epoch].view(np.uint32)), "current_lengths"
        win, _ = observer.windows()
        for key in ("successes", "energy"):
            ref = np.asarray(m["windows"][epoch][key], np.float64)
            assert np.array_equal(win[key], ref), "window %s epoch %d" % (key, epoch)  # same values, same (deque) order
        algo.writer.rows = []
        observer.after_print_stats(frame=(epoch + 1) * N * Hz, epoch_num=epoch + 1, total_time=1.5 * (epoch + 1))
        ref_rows = [tuple(r) for r in m["writes"][epoch]]
        assert [r[0] for r in algo.writer.rows] == [r[0] for r in ref_rows], "tags epoch %d" % epoch
        for (tag, v, s), (_, rv, rs) in zip(algo.writer.rows, ref_rows):
            assert s == rs, tag
            assert v == rv or abs(v - rv) <= 1e-6 * max(1.0, abs(rv)), (tag, v, rv)  # float32 means of the same values


def oracle_window_quantile(capacity=4096):
    """ops.WindowQuantile with its five kernels replaced by the CPU restatement (host tensors): the protocol
    code under test is the product's, the arithmetic the checker's."""
    import torch
    from isaacgym_amd import ops
    from oracle import oracle

    class OracleWindowQuantile(ops.WindowQuantile):
        def _alloc(self, n, dtype):
            return torch.zeros(n, dtype=dtype)

        def begin(self, q, n_total, reset):
            oracle.window_begin(self.workspace.numpy(), q, n_total, reset)

        def histogram(self, x):
            oracle.window_histogram(x.numpy(), self.workspace.numpy())

        def locate(self):
            oracle.window_locate(self.workspace.numpy(), self.capacity, self.decision.numpy())

        def collect(self, x, num_slots):
            oracle.window_collect(x.numpy(), self.workspace.numpy(), self.slots.numpy(), num_slots)

        def finish(self, gathered, num_ranks, num_slots, out):
            v = oracle.window_finish(self.workspace.numpy(), gathered.numpy(), num_ranks, num_slots)
            if out is not None:
                out.fill_(v)

    return OracleWindowQuantile("cpu", capacity)


class LocalShards:
    """The collectives of a sharded job run inside one process: the shards' objects take turns, `all_reduce` sums
    the tensors the shards handed in, `all_gather` stacks them.  Drives N WindowQuantile objects in lock step."""

    def __init__(self, objs):
        self.objs = objs

    def run(self, shards, q, reset=False, outs=None):
        import threading
        import torch
        n_total = sum(int(s.numel()) for s in shards)
        world = len(self.objs)
        barrier = threading.Barrier(world)
        box, res, err = [None] * world, [None] * world, []

        def exchange(rank, t, combine):
            box[rank] = t
            barrier.wait()
            out = combine([b for b in box])
            barrier.wait()
            return out

        def work(rank):
            try:
                def all_reduce(t):
                    t.copy_(exchange(rank, t.clone(), lambda ts: torch.stack(ts).sum(0).to(t.dtype)))

                def all_gather(t):
                    return exchange(rank, t.clone(), lambda ts: torch.stack(ts))
                res[rank] = self.objs[rank].run(shards[rank], q, n_total, all_reduce, all_gather,
                                                out=None if outs is None else outs[rank], reset=reset)
            except BaseException as e:  # noqa: BLE001 - re-raised in the caller
                err.append(e)
                barrier.abort()

        threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        if err:
            raise err[0]
        return res


# ---- angles_edge.npz: torso rotations whose roll / yaw atan2 numerators are zero, signed zero, subnormal
# or rounding noise of both signs, recorded from the reference's get_euler_xyz / compute_*_observations ----
def angles_edge_check(g, euler_fn, ant_step, humanoid_step, to_buf, to_np):
    """euler_fn(q [n,4] numpy) -> (roll, pitch, yaw) numpy; *_step(params, bufs, n) on backend buffers.
    Everything is compared strictly (no modulo 2 pi); returns {what: number of reference values at 2 pi}."""
    from isaacgym_amd import synthetic
    root = g["root"]
    n = root.shape[0]
    seen = {}
    got = euler_fn(np.ascontiguousarray(root[:, 3:7]))
    for name, x in zip(("euler_roll", "euler_pitch", "euler_yaw"), got):
        assert_angle_close(np.asarray(x).reshape(-1), g[name], what="angles_edge " + name)
        seen[name] = int((g[name] > 6.28).sum())
    for tag in ("", "_tilt"):
        for task, spec, step in (("ant", synthetic.ANT, ant_step), ("hum", synthetic.HUMANOID, humanoid_step)):
            D, S, hum = spec["num_dof"], spec["num_sensor_floats"], spec["humanoid"]
            p = synthetic.loco_params(spec)
            p.set("inv_start_rot", g["inv_start_rot" + tag].tolist())
            p.set("dof_limits_lower", [-1.0] * D)
            p.set("dof_limits_upper", [1.0] * D)
            z = lambda *shape, dt=np.float32: to_buf(np.zeros(shape, dt))  # noqa: E731
            bufs = dict(root_states=to_buf(root), dof_state=z(n, D, 2), sensors=z(n, S), actions=z(n, D),
                        initial_root_states=z(n, 13), initial_dof_pos=z(n, D), obs=z(n, spec["num_obs"]),
                        potentials=z(n), prev_potentials=z(n), up_vec=z(n, 3), heading_vec=z(n, 3), rew=z(n),
                        reset=z(n, dt=np.int64), progress=z(n, dt=np.int64), randomize=z(n, dt=np.int64))
            if hum:
                bufs["dof_force"] = z(n, D)
            step(p, bufs, n)
            obs, ref = to_np(bufs["obs"])[:, :12], g["%s_obs_head%s" % (task, tag)]
            what = "angles_edge %s%s " % (task, tag)
            nonangle = [c for c in range(12) if c not in LOCO_ANGLE_COLS]
            assert bit_equal_fraction(obs[:, nonangle], ref[:, nonangle]) == 1.0, what + "non-angle columns"
            flips = assert_angle_close(obs[:, list(LOCO_ANGLE_COLS)], ref[:, list(LOCO_ANGLE_COLS)], what=what + "yaw / roll / angle_to_target",
                                       branch_cut=loco_branch_cut(hum))
            seen[what + "at 2 pi"] = int((ref[:, 7:9] > 6.28).sum())
            seen[what + "+-pi flips"] = flips
    return seen


def edge_quats(n, seed):
    """[n,4] float32 rotations concentrated on the remainder wrap of get_euler_xyz: a quarter each with the roll
    numerator 2(wx+yz) and the yaw numerator 2(wz+xy) solved to 0 / +-1e-9 .. +-1e-6 in fp64 (after rounding to
    fp32 the sign is rounding noise), a quarter pure pitch with x and z from {+-0, subnormals, +-1e-30 .. +-1e-7},
    a quarter generic."""
    rs = np.random.default_rng(seed)
    deltas = np.array([0.0, 1e-9, -1e-9, 1e-8, -1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 3e-10, -3e-10])
    tiny = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-40, -1e-40, 1e-38, -1e-38, 1e-30, -1e-30, 1e-20, -1e-20, 1e-12, -1e-12,
                     1e-9, -1e-9, 1e-7, -1e-7])
    m = n // 4
    parts = []
    for kind in (0, 1):
        a, b = rs.normal(size=m), rs.normal(size=m)
        w = rs.uniform(0.2, 1.0, size=m) * rs.choice([-1.0, 1.0], size=m)
        c = (rs.choice(deltas, size=m) - a * b) / w
        q = np.stack([c, a, b, w], -1) if kind == 0 else np.stack([a, b, c, w], -1)
        parts.append(q / np.linalg.norm(q, axis=-1, keepdims=True))
    pitch = rs.uniform(-np.pi, np.pi, size=m)
    parts.append(np.stack([rs.choice(tiny, size=m), np.sin(pitch / 2), rs.choice(tiny, size=m), np.cos(pitch / 2)], -1))
    q = rs.normal(size=(n - 3 * m, 4))
    parts.append(q / np.linalg.norm(q, axis=-1, keepdims=True))
    return np.ascontiguousarray(np.concatenate(parts).astype(np.float32)[rs.permutation(n)])
