"""ctypes mirror of include/igx.h (struct layouts and constants only).

No library is loaded here; isaacgym_amd._lib loads libigx.so.  Field order and
types must match the header exactly.
"""
import ctypes as C

IGX_MAX_DOF = 32

IGX_OK = 0
IGX_E_NULL = -1
IGX_E_SIZE = -2
IGX_E_PARAM = -3
IGX_E_ALIGN = -4
IGX_E_NODEVICE = -5

f32 = C.c_float
i32 = C.c_int32
i64 = C.c_int64
u64 = C.c_uint64
ptr = C.c_void_p


class _Struct(C.Structure):
    def __init__(self, **kw):
        super().__init__()
        for k, v in kw.items():
            self.set(k, v)

    def set(self, k, v):
        if isinstance(v, (int, float)):      # the common case, straight into ctypes
            setattr(self, k, v)
            return
        cur = getattr(self, k)
        if isinstance(cur, C.Array):
            v = v.tolist() if hasattr(v, "tolist") else list(v)
            if len(v) > len(cur):
                raise ValueError("%s: %d values for %d slots" % (k, len(v), len(cur)))
            cur[0:len(v)] = v
        else:
            setattr(self, k, v)


class CartpoleParams(_Struct):
    _fields_ = [("reset_dist", f32), ("max_episode_length_m1", f32),
                ("reward_variant", i32), ("clip_obs", f32),
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr), ("env_offset", i64)]


class CartpoleBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("rand", ptr), ("obs", ptr), ("rew", ptr),
                ("reset", ptr), ("progress", ptr), ("obs_clamped", ptr), ("time_outs", ptr)]


class LocoParams(_Struct):
    _fields_ = [("num_dof", i32), ("num_sensor_floats", i32),
                ("dt", f32), ("dof_vel_scale", f32), ("contact_force_scale", f32),
                ("angular_velocity_scale", f32),
                ("up_weight", f32), ("heading_weight", f32),
                ("actions_cost_scale", f32), ("energy_cost_scale", f32),
                ("joints_at_limit_cost_scale", f32),
                ("termination_height", f32), ("death_cost", f32),
                ("max_episode_length_m1", f32),
                ("targets", f32 * 3), ("inv_start_rot", f32 * 4),
                ("basis_vec0", f32 * 3), ("basis_vec1", f32 * 3),
                ("dof_limits_lower", f32 * IGX_MAX_DOF),
                ("dof_limits_upper", f32 * IGX_MAX_DOF),
                ("motor_efforts", f32 * IGX_MAX_DOF),
                ("max_motor_effort", f32), ("clip_obs", f32), ("reserved", i32),
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr), ("env_offset", i64)]


class LocoBuffers(_Struct):
    _fields_ = [("root_states", ptr), ("dof_state", ptr), ("dof_force", ptr),
                ("sensors", ptr), ("actions", ptr),
                ("initial_root_states", ptr), ("initial_dof_pos", ptr), ("rand", ptr),
                ("obs", ptr), ("potentials", ptr), ("prev_potentials", ptr),
                ("up_vec", ptr), ("heading_vec", ptr), ("rew", ptr),
                ("reset", ptr), ("progress", ptr), ("randomize", ptr),
                ("true_objective", ptr), ("obs_clamped", ptr), ("time_outs", ptr)]

IGX_ALLEGRO_DOF = 16
IGX_ALLEGRO_RAND_PER_ENV = 45
ALLEGRO_OBS_TYPES = {"full_no_vel": 0, "full": 1, "full_state": 2}
ALLEGRO_NUM_OBS = {0: 50, 1: 72, 2: 88}


class AllegroParams(_Struct):
    _fields_ = [("obs_type", i32), ("max_consecutive_successes", i32), ("ignore_z_rot", i32),
                ("use_relative_control", i32),
                ("dist_reward_scale", f32), ("rot_reward_scale", f32), ("rot_eps", f32),
                ("action_penalty_scale", f32),
                ("success_tolerance", f32), ("success_threshold_c", f32),
                ("reach_goal_bonus", f32), ("fall_dist", f32), ("fall_penalty", f32),
                ("max_episode_length_m1", f32),
                ("av_factor", f32), ("one_minus_av_factor", f32),
                ("vel_obs_scale", f32), ("force_torque_obs_scale", f32),
                ("reset_position_noise", f32), ("reset_dof_pos_noise", f32), ("reset_dof_vel_noise", f32),
                ("act_moving_average", f32), ("one_minus_act_moving_average", f32),
                ("dof_speed_scale_dt", f32),
                ("goal_displacement", f32 * 3),
                ("dof_lower", f32 * IGX_ALLEGRO_DOF), ("dof_upper", f32 * IGX_ALLEGRO_DOF),
                ("clip_obs", f32), ("clip_actions", f32), ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr),
                ("env_offset", i64)]


class AllegroBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("dof_force", ptr), ("root_state_tensor", ptr),
                ("object_indices", ptr), ("goal_object_indices", ptr),
                ("goal_states", ptr), ("goal_init_state", ptr), ("object_init_state", ptr),
                ("actions", ptr), ("prev_targets", ptr), ("cur_targets", ptr), ("rand", ptr),
                ("obs", ptr), ("states", ptr), ("rew", ptr),
                ("reset", ptr), ("reset_goal", ptr), ("progress", ptr), ("randomize", ptr),
                ("successes", ptr), ("consecutive_successes", ptr), ("workspace", ptr),
                ("obs_clamped", ptr), ("states_clamped", ptr), ("time_outs", ptr), ("shard_record", ptr), ("actions_out", ptr)]

IGX_SHADOW_DOF, IGX_SHADOW_ACTIONS, IGX_SHADOW_FINGERTIPS, IGX_SHADOW_RAND_PER_ENV = 24, 20, 5, 61
IGX_SHADOW_OBS = {"openai": 0, "full_no_vel": 1, "full": 2, "full_state": 3}
IGX_SHADOW_NUM_OBS = {"openai": 42, "full_no_vel": 77, "full": 157, "full_state": 211}


class ShadowParams(_Struct):
    _fields_ = [("obs_type", i32), ("max_consecutive_successes", i32), ("ignore_z_rot", i32),
                ("use_relative_control", i32),
                ("dist_reward_scale", f32), ("rot_reward_scale", f32), ("rot_eps", f32),
                ("action_penalty_scale", f32),
                ("success_tolerance", f32), ("success_threshold_c", f32),
                ("reach_goal_bonus", f32), ("fall_dist", f32), ("fall_penalty", f32),
                ("max_episode_length_m1", f32),
                ("av_factor", f32), ("one_minus_av_factor", f32),
                ("vel_obs_scale", f32), ("force_torque_obs_scale", f32),
                ("reset_position_noise", f32), ("reset_dof_pos_noise", f32), ("reset_dof_vel_noise", f32),
                ("act_moving_average", f32), ("one_minus_act_moving_average", f32),
                ("dof_speed_scale_dt", f32),
                ("goal_displacement", f32 * 3), ("num_bodies", i32),
                ("dof_lower", f32 * IGX_SHADOW_DOF), ("dof_upper", f32 * IGX_SHADOW_DOF),
                ("actuated_dof_indices", i32 * IGX_SHADOW_ACTIONS), ("fingertip_handles", i32 * IGX_SHADOW_FINGERTIPS),
                ("clip_obs", f32), ("clip_actions", f32), ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr),
                ("env_offset", i64)]


class ShadowBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("dof_force", ptr), ("vec_sensor_tensor", ptr), ("rigid_body_states", ptr),
                ("root_state_tensor", ptr), ("object_indices", ptr), ("goal_object_indices", ptr),
                ("goal_states", ptr), ("goal_init_state", ptr), ("object_init_state", ptr),
                ("actions", ptr), ("prev_targets", ptr), ("cur_targets", ptr), ("rand", ptr),
                ("obs", ptr), ("states", ptr), ("rew", ptr),
                ("reset", ptr), ("reset_goal", ptr), ("progress", ptr), ("randomize", ptr),
                ("successes", ptr), ("consecutive_successes", ptr), ("workspace", ptr),
                ("obs_clamped", ptr), ("states_clamped", ptr), ("time_outs", ptr), ("shard_record", ptr), ("actions_out", ptr)]


IGX_HAND_FORCE_RAND_PER_ENV = 5


class HandForces(_Struct):
    _fields_ = [("rb_forces", ptr), ("random_force_prob", ptr), ("reset", ptr), ("rand", ptr),
                ("num_bodies", i32), ("object_rb_handle", i32), ("object_rb_mass", f32), ("force_scale", f32),
                ("decay", f32), ("log_prob_lo", f32), ("log_prob_hi", f32), ("reserved", i32),
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr), ("env_offset", i64)]


IGX_AMP_NUM_BODIES = 15
IGX_AMP_NUM_DOF = 28
IGX_AMP_NUM_KEY_BODIES = 4
IGX_AMP_OBS = 105


class MotionTables(_Struct):
    _fields_ = [("num_motions", i32), ("reserved", i32),
                ("motion_length", ptr), ("motion_dt", ptr), ("motion_num_frames", ptr), ("frame_offset", ptr),
                ("root_pos", ptr), ("root_rot", ptr), ("local_rot", ptr), ("root_vel", ptr),
                ("root_ang_vel", ptr), ("key_pos", ptr), ("dof_vel", ptr)]


class MotionSampler(_Struct):
    _fields_ = [("cdf", ptr), ("random_time", i32), ("ref_init_prob", f32), ("seed", u64), ("rng_offset", u64),
                ("rng_offset_dev", ptr), ("env_offset", i64)]


class MotionStateOut(_Struct):
    _fields_ = [("root_pos", ptr), ("root_rot", ptr), ("dof_pos", ptr), ("root_vel", ptr),
                ("root_ang_vel", ptr), ("dof_vel", ptr), ("key_pos", ptr),
                ("frame_idx0", ptr), ("frame_idx1", ptr), ("blend", ptr)]


class AmpParams(_Struct):
    _fields_ = [("num_amp_obs_steps", i32), ("local_root_obs", i32), ("enable_early_termination", i32),
                ("num_contact_bodies", i32), ("contact_body_ids", i32 * 4), ("key_body_ids", i32 * 4),
                ("termination_height", f32), ("max_episode_length_m1", f32), ("clip_obs", f32), ("reserved", i32)]


class AmpBuffers(_Struct):
    _fields_ = [("root_states", ptr), ("dof_state", ptr), ("rigid_body_state", ptr), ("contact_forces", ptr),
                ("obs", ptr), ("amp_obs", ptr), ("rew", ptr), ("reset", ptr), ("terminate", ptr),
                ("progress", ptr), ("obs_clamped", ptr), ("time_outs", ptr)]


IGX_NOISE_ADDITIVE, IGX_NOISE_SCALING = 0, 1
IGX_NOISE_GAUSSIAN, IGX_NOISE_UNIFORM = 0, 1
IGX_ACT_FORCE, IGX_ACT_PD_TARGET, IGX_ACT_FIRST_DOF = 0, 1, 2


class Noise(_Struct):
    _fields_ = [("corr", ptr), ("fresh", ptr),
                ("corr_scale", f32), ("corr_shift", f32), ("scale", f32), ("shift", f32),
                ("operation", i32), ("distribution", i32), ("seed", u64), ("offset", u64), ("offset_dev", ptr)]


class ActionStage(_Struct):
    _fields_ = [("actions", ptr), ("actions_out", ptr), ("targets", ptr), ("gain", ptr), ("offset", ptr),
                ("noise", Noise), ("clip_actions", f32), ("power_scale", f32),
                ("mode", i32), ("num_actions", i32), ("num_dof", i32), ("reserved", i32)]


class VecTaskTail(_Struct):
    _fields_ = [("obs_buf", ptr), ("obs_out", ptr), ("obs_count", i64),
                ("states_buf", ptr), ("states_out", ptr), ("states_count", i64),
                ("progress", ptr), ("reset", ptr), ("time_outs", ptr),
                ("max_episode_length_m1", C.c_double), ("clip_obs", f32), ("reserved", i32),
                ("obs_noised", ptr), ("obs_noise", Noise)]


IGX_HIM_DOF, IGX_HIM_BODIES, IGX_HIM_OBS = 63, 22, 205


class HumanoidImitateParams(_Struct):
    _fields_ = [("dt", f32), ("dof_vel_scale", f32), ("up_weight", f32), ("actions_cost_scale", f32),
                ("termination_height", f32), ("termination_radius_sq", f32), ("max_episode_length_m1", f32),
                ("reserved", i32), ("targets", f32 * 3), ("clip_obs", f32),
                ("dof_limits_lower", f32 * (IGX_HIM_DOF + 1)), ("dof_limits_upper", f32 * (IGX_HIM_DOF + 1)),
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr), ("env_offset", i64)]


class HumanoidImitateBuffers(_Struct):
    _fields_ = [("root_states", ptr), ("dof_state", ptr), ("rigid_body_states", ptr), ("actions", ptr),
                ("initial_root_states", ptr), ("initial_dof_pos", ptr), ("rand", ptr), ("motion_frame", ptr),
                ("obs", ptr), ("potentials", ptr), ("prev_potentials", ptr), ("rew", ptr),
                ("reset", ptr), ("progress", ptr), ("randomize", ptr), ("obs_clamped", ptr), ("time_outs", ptr)]


class ImitateArgs(_Struct):
    _fields_ = [("motion", ptr), ("num_frames", i64), ("target_pose_idx", ptr), ("pose_diff", ptr), ("workspace", ptr),
                ("quantile", C.c_double), ("distance_floor", C.c_double), ("death_cost", C.c_double), ("n_total", i64)]


IGX_QUANTILE_PASSES = 6


def IGX_QUANTILE_HIST_OFFSET(p):
    return 512 + p * (2 * 2048 * 8)


IGX_QUANTILE_WORKSPACE_BYTES = IGX_QUANTILE_HIST_OFFSET(IGX_QUANTILE_PASSES)
IGX_QUANTILE_WINDOW_BINS = 16384
IGX_QUANTILE_WINDOW_HIST_WORDS = IGX_QUANTILE_WINDOW_BINS + 32
IGX_QUANTILE_WINDOW_HIST_OFFSET = 128
IGX_QUANTILE_WINDOW_WORKSPACE_BYTES = IGX_QUANTILE_WINDOW_HIST_OFFSET + 4 * IGX_QUANTILE_WINDOW_HIST_WORDS
IGX_QUANTILE_WINDOW_MAX_CANDIDATES = 4096
IGX_QUANTILE_WINDOW_MAX_ROUNDS = 12
IGX_QUANTILE_WINDOW_MAX_RANKS = 1024
IGX_QW_COLLECT, IGX_QW_REFINED, IGX_QW_VALUE, IGX_QW_GIVE_UP = 0, 1, 2, 4


class EpisodeBuffers(_Struct):
    _fields_ = [("rew", ptr), ("done", ptr), ("current_rewards", ptr), ("current_lengths", ptr), ("workspace", ptr)]


IGX_EPISODE_WINDOW_MAX_KEYS = 16
IGX_EPISODE_WINDOW_HEADER_BYTES = 32


def IGX_EPISODE_WINDOW_WORKSPACE_BYTES(n):
    return 32 + 16 * (((int(n) + 1023) // 1024 + 3) // 4)


class EpisodeWindow(_Struct):
    _fields_ = [("done", ptr), ("num_keys", i32), ("reserved", i32), ("games_to_track", i64),
                ("values", ptr * IGX_EPISODE_WINDOW_MAX_KEYS), ("cumulative", ptr * IGX_EPISODE_WINDOW_MAX_KEYS),
                ("window", ptr), ("workspace", ptr)]


def make_buffers(cls, arrays, ptr_of):
    """Build a *Buffers struct from a name->array mapping.

    ptr_of(array) returns the raw address; missing / None entries become NULL.
    Unknown names raise so that a typo cannot silently drop a buffer.
    """
    names = {n for n, _ in cls._fields_}
    for k in arrays:
        if k not in names:
            raise KeyError("%s has no buffer named %r" % (cls.__name__, k))
    b = cls()
    for n, ctype in cls._fields_:
        if ctype is not ptr:
            if n in arrays:
                setattr(b, n, arrays[n])
            continue
        a = arrays.get(n)
        setattr(b, n, None if a is None else ptr_of(a))
    return b

# igx_allegro_post_physics_static: the library-owned reduction scratch
IGX_ALLEGRO_STATIC_MAX_ENVS = 1 << 21
IGX_ALLEGRO_STATIC_SCRATCH_BYTES = 16 * ((1 << 21) // 16 + 1)


# ---- how large every buffer of a step call must be ------------------------------------------
# element counts as include/igx.h documents them: ("env", k) = k elements per environment,
# ("fixed", k) = k elements whatever the batch, ("bytes", f) = f(n_envs) bytes of opaque scratch,
# ("rows", k) = rows of k elements addressed through an index tensor (at least one row per environment)
def IGX_ALLEGRO_WORKSPACE_BYTES(n):
    return 16 * ((int(n) + 15) // 16 + 1)


def IGX_EPISODE_WORKSPACE_BYTES(n):
    return 32 * ((int(n) + 255) // 256 + 2)


def _env(k):
    return ("env", int(k))


_ONE = _env(1)


def _cartpole_widths(p):
    return dict(dof_state=_env(4), rand=_env(4), obs=_env(4), rew=_ONE, reset=_ONE, progress=_ONE,
                obs_clamped=_env(4), time_outs=_ONE)


def _loco_widths(p):
    d, s = int(p.num_dof), int(p.num_sensor_floats)
    hum = d == 21  # the 21-dof Humanoid also observes its joint forces
    nobs = 12 + (3 if hum else 2) * d + s + d
    return dict(root_states=_env(13), dof_state=_env(2 * d), dof_force=_env(d), sensors=_env(s), actions=_env(d),
                initial_root_states=_env(13), initial_dof_pos=_env(d), rand=_env(2 * d), obs=_env(nobs),
                potentials=_ONE, prev_potentials=_ONE, up_vec=_env(3), heading_vec=_env(3), rew=_ONE, reset=_ONE,
                progress=_ONE, randomize=_ONE, true_objective=_ONE, obs_clamped=_env(nobs), time_outs=_ONE)


def _hand_widths(dof, actions, rand, nobs, nstates, extra):
    w = dict(dof_state=_env(2 * dof), dof_force=_env(dof), root_state_tensor=("rows", 13), object_indices=_ONE,
             goal_object_indices=_ONE, goal_states=_env(13), goal_init_state=_env(13), object_init_state=_env(13),
             actions=_env(actions), prev_targets=_env(dof), cur_targets=_env(dof), rand=_env(rand), obs=_env(nobs),
             states=_env(nstates), rew=_ONE, reset=_ONE, reset_goal=_ONE, progress=_ONE, randomize=_ONE,
             successes=_ONE, consecutive_successes=("fixed", 1), workspace=("bytes", IGX_ALLEGRO_WORKSPACE_BYTES),
             obs_clamped=_env(nobs), states_clamped=_env(nstates), time_outs=_ONE, shard_record=("bytes", lambda n: 16), actions_out=_env(actions))
    w.update(extra)
    return w


def _allegro_widths(p):
    return _hand_widths(IGX_ALLEGRO_DOF, IGX_ALLEGRO_DOF, IGX_ALLEGRO_RAND_PER_ENV, ALLEGRO_NUM_OBS[int(p.obs_type)], 88, {})


_SHADOW_NUM_OBS_BY_TYPE = {0: 42, 1: 77, 2: 157, 3: 211}


def _shadow_widths(p):
    return _hand_widths(IGX_SHADOW_DOF, IGX_SHADOW_ACTIONS, IGX_SHADOW_RAND_PER_ENV,
                        _SHADOW_NUM_OBS_BY_TYPE[int(p.obs_type)], 211,
                        dict(vec_sensor_tensor=_env(30), rigid_body_states=_env(13 * max(int(p.num_bodies), 0))))


def _amp_widths(p):
    k = max(int(p.num_amp_obs_steps), 1)
    return dict(root_states=_env(13), dof_state=_env(2 * IGX_AMP_NUM_DOF), rigid_body_state=_env(13 * IGX_AMP_NUM_BODIES),
                contact_forces=_env(3 * IGX_AMP_NUM_BODIES), obs=_env(IGX_AMP_OBS), amp_obs=_env(k * IGX_AMP_OBS), rew=_ONE,
                reset=_ONE, terminate=_ONE, progress=_ONE, obs_clamped=_env(IGX_AMP_OBS), time_outs=_ONE)


def _him_widths(p):
    return dict(root_states=_env(13), dof_state=_env(2 * IGX_HIM_DOF), rigid_body_states=_env(13 * IGX_HIM_BODIES),
                actions=_env(IGX_HIM_DOF), initial_root_states=_env(13), initial_dof_pos=_env(IGX_HIM_DOF),
                rand=_env(2 * IGX_HIM_DOF), motion_frame=("fixed", IGX_HIM_DOF + 13 * IGX_HIM_BODIES), obs=_env(IGX_HIM_OBS),
                potentials=_ONE, prev_potentials=_ONE, rew=_ONE, reset=_ONE, progress=_ONE, randomize=_ONE,
                obs_clamped=_env(IGX_HIM_OBS), time_outs=_ONE)


def _episode_widths(p):
    return dict(rew=_ONE, done=_ONE, current_rewards=_ONE, current_lengths=_ONE,
                workspace=("bytes", IGX_EPISODE_WORKSPACE_BYTES))


BUFFER_WIDTHS = {CartpoleBuffers: _cartpole_widths, LocoBuffers: _loco_widths, AllegroBuffers: _allegro_widths,
                 ShadowBuffers: _shadow_widths, AmpBuffers: _amp_widths, HumanoidImitateBuffers: _him_widths,
                 EpisodeBuffers: _episode_widths}


def required_elements(cls, params, n_envs):
    """field -> (minimum number of elements, or of bytes for opaque scratch, is_bytes) for a call on
    n_envs environments; None for a struct without a table."""
    fn = BUFFER_WIDTHS.get(cls)
    if fn is None:
        return None
    out = {}
    n = int(n_envs)
    for name, (kind, k) in fn(params).items():
        if kind == "env" or kind == "rows":
            out[name] = (k * n, False)
        elif kind == "fixed":
            out[name] = (k, False)
        else:
            out[name] = (k(n), True)
    return out
