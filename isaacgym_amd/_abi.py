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
                ("reward_variant", i32), ("reserved", i32),
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr)]


class CartpoleBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("rand", ptr), ("obs", ptr), ("rew", ptr),
                ("reset", ptr), ("progress", ptr)]


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
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr)]


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
                ("clip_obs", f32), ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr)]


class AllegroBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("dof_force", ptr), ("root_state_tensor", ptr),
                ("object_indices", ptr), ("goal_object_indices", ptr),
                ("goal_states", ptr), ("goal_init_state", ptr), ("object_init_state", ptr),
                ("actions", ptr), ("prev_targets", ptr), ("cur_targets", ptr), ("rand", ptr),
                ("obs", ptr), ("states", ptr), ("rew", ptr),
                ("reset", ptr), ("reset_goal", ptr), ("progress", ptr), ("randomize", ptr),
                ("successes", ptr), ("consecutive_successes", ptr), ("workspace", ptr),
                ("obs_clamped", ptr), ("states_clamped", ptr), ("time_outs", ptr)]

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
                ("clip_obs", f32), ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr)]


class ShadowBuffers(_Struct):
    _fields_ = [("dof_state", ptr), ("dof_force", ptr), ("vec_sensor_tensor", ptr), ("rigid_body_states", ptr),
                ("root_state_tensor", ptr), ("object_indices", ptr), ("goal_object_indices", ptr),
                ("goal_states", ptr), ("goal_init_state", ptr), ("object_init_state", ptr),
                ("actions", ptr), ("prev_targets", ptr), ("cur_targets", ptr), ("rand", ptr),
                ("obs", ptr), ("states", ptr), ("rew", ptr),
                ("reset", ptr), ("reset_goal", ptr), ("progress", ptr), ("randomize", ptr),
                ("successes", ptr), ("consecutive_successes", ptr), ("workspace", ptr),
                ("obs_clamped", ptr), ("states_clamped", ptr), ("time_outs", ptr)]


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
                ("rng_offset_dev", ptr)]


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
                ("seed", u64), ("rng_offset", u64), ("rng_offset_dev", ptr)]


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


class EpisodeBuffers(_Struct):
    _fields_ = [("rew", ptr), ("done", ptr), ("current_rewards", ptr), ("current_lengths", ptr), ("workspace", ptr)]


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
