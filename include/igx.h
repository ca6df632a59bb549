/*
 * igx.h - C ABI of libigx.so: fused per-step observation / reward / reset
 * kernels for vectorised RL tasks on AMD MI355X (gfx950).
 *
 * Every entry point replaces a group of module-level @torch.jit.script
 * functions plus the in-place bookkeeping of one task class of the reference
 * (IsaacGymEnvs fork, paths relative to IsaacGymEnvs/isaacgymenvs/).  The
 * reference has no FFI for this path (it is Python calling ATen); the
 * signatures below are what a ctypes binding on the reference side binds.
 *
 * Conventions
 *  - All pointers are DEVICE pointers on the current HIP device unless a field
 *    says "host".  The library borrows them for the duration of the (async)
 *    launch and never allocates or frees caller memory.
 *  - Work is enqueued on the caller's stream (hipStream_t passed as void*,
 *    NULL = the default stream).  No entry point synchronises the device.
 *  - Layouts are the reference's: root / rigid-body state rows are 13 floats
 *    (pos 3, quat xyzw 4, lin vel 3, ang vel 3); dof_state is [N, D, 2]
 *    interleaved (pos, vel); flag buffers are int64.
 *  - Return value: 0 = ok, negative = IGX_E_* argument error (nothing was
 *    launched), positive = hipError_t of the failed launch.
 *    igx_last_error_string() describes the last error of the calling thread.
 *  - Threads: entry points may be called from several host threads at once, each on its own stream and
 *    buffers; the library keeps no state between calls apart from that per-thread error text, tuning
 *    constants read once from the environment, and the device address of the one file-scope scratch array
 *    (igx_allegro_post_physics_static, whose calls on one device share it and must not overlap).
 */
#ifndef IGX_H
#define IGX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IGX_OK 0
#define IGX_E_NULL (-1)    /* a required pointer is NULL */
#define IGX_E_SIZE (-2)    /* a count is negative or too large */
#define IGX_E_PARAM (-3)   /* a parameter is outside what the kernels support */
#define IGX_E_ALIGN (-4)   /* a pointer is not aligned as documented */
#define IGX_E_NODEVICE (-5) /* no usable HIP device */

#define IGX_MAX_DOF 32

typedef void *igx_stream_t; /* hipStream_t */

/* Library identification and error text (no reference counterpart).
 * igx_version() = 100 * major + 10 * minor, 120 for this header: a binding that needs igx_motion_fetch_amp_obs,
 * igx_motion_amp_obs_history or igx_extract_positions_from_rigid_body_states checks for >= 120. */
int igx_version(void);
const char *igx_last_error_string(void);

/* ------------------------------------------------------------------------
 * Cartpole: one fused post_physics_step.
 * Replaces tasks/cartpole.py:170-178 (post_physics_step), :148-162
 * (reset_idx), :135-146 (compute_observations) and :185-201
 * (compute_cartpole_reward).
 * Order per env: progress += 1; if reset != 0 { dof_pos = 0.9*(u-0.5),
 * dof_vel = 0.5*(u-0.5), reset = progress = 0 }; obs = dof_state row;
 * reward / reset decision.
 * ---------------------------------------------------------------------- */
typedef struct igx_cartpole_params {
    float reset_dist;             /* env.resetDist, 3.0 */
    float max_episode_length_m1;  /* max_episode_length - 1 = 499 */
    int32_t reward_variant;       /* 0: tasks/cartpole.py:185-201; 1: tasks/cartpole_sample.py:316-335
                                     (1 - 0.001 a^2 - 0.001|cart_vel| - 0.001|pole_vel|) */
    float clip_obs;               /* env.clipObservations (5.0), used only when obs_clamped is given */
    uint64_t seed;                /* in-kernel Philox key when rand == NULL */
    uint64_t rng_offset;          /* step counter mixed into the Philox counter */
    const uint64_t *rng_offset_dev; /* optional device-resident counter added to rng_offset when the kernel draws:
                                     a recorded HIP graph replays the same argument values, the counter (advanced
                                     on the stream between replays) gives every replay fresh draws; NULL = 0 */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_cartpole_params;

typedef struct igx_cartpole_buffers {
    float *dof_state;    /* [N,2,2] in/out (rewritten only for reset envs) */
    const float *rand;   /* [N,4] uniforms in [0,1): pos0,pos1,vel0,vel1; NULL = Philox */
    float *obs;          /* [N,4] out */
    float *rew;          /* [N] out */
    int64_t *reset;      /* [N] in/out */
    int64_t *progress;   /* [N] in/out */
    /* optional: the tail of VecTask.step (vec_task.py:404, :412) in the same launch; NULL skips */
    float *obs_clamped;  /* [N,4] clamp(obs, -clip_obs, clip_obs); must not alias obs */
    uint8_t *time_outs;  /* [N] bool (progress >= max_episode_length - 1) & (reset != 0) after the step */
} igx_cartpole_buffers;

/* Cartpole.reset_idx (tasks/cartpole.py:148-162) for an explicit list of m
 * environment ids: the same draws and arithmetic as the flagged resets inside
 * igx_cartpole_step (rand, when given, is indexed by environment id). */
int igx_cartpole_reset(const igx_cartpole_params *p, const igx_cartpole_buffers *b, const int64_t *env_ids,
                       int64_t m, igx_stream_t stream);

int igx_cartpole_step(const igx_cartpole_params *p, const igx_cartpole_buffers *b,
                      int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * Ant / Humanoid locomotion: one fused post_physics_step.
 * Ant:      tasks/ant.py:376-386 (post_physics_step), :339-366 (reset_idx),
 *           :463-497 (compute_ant_observations), :414-460 (compute_ant_reward),
 *           :332-337 (compute_true_objective).
 * Humanoid: tasks/humanoid.py:398-407, :360-390, :489-524, :434-486.
 * Shared helpers: utils/torch_jit_utils.py:41-62 quat_mul, :65-67 normalize,
 * :80-103 quat_rotate(_inverse), :175-195 get_euler_xyz, :239-241 unscale,
 * :248-277 compute_heading_and_up / compute_rot, :126-128 normalize_angle.
 * ---------------------------------------------------------------------- */
typedef struct igx_loco_params {
    int32_t num_dof;            /* 8 (Ant) or 21 (Humanoid) */
    int32_t num_sensor_floats;  /* 24 (Ant: 4 sensors x 6) or 12 (Humanoid) */
    float dt;                   /* sim.dt 0.0166 */
    float dof_vel_scale;        /* 0.2 Ant, 0.1 Humanoid */
    float contact_force_scale;  /* 0.1 Ant, 0.01 Humanoid */
    float angular_velocity_scale; /* Humanoid 0.25 (Ant: unused) */
    float up_weight, heading_weight;
    float actions_cost_scale, energy_cost_scale, joints_at_limit_cost_scale;
    float termination_height, death_cost;
    float max_episode_length_m1; /* episodeLength - 1 */
    float targets[3];            /* [1000,0,0] */
    float inv_start_rot[4];      /* conj(start_rotation), xyzw */
    float basis_vec0[3];         /* heading basis [1,0,0] */
    float basis_vec1[3];         /* up basis [0,0,1] */
    float dof_limits_lower[IGX_MAX_DOF];
    float dof_limits_upper[IGX_MAX_DOF];
    float motor_efforts[IGX_MAX_DOF]; /* Humanoid only */
    float max_motor_effort;           /* Humanoid only */
    float clip_obs;                   /* env.clipObservations (+inf copies); read only with obs_clamped */
    int32_t reserved;
    uint64_t seed;
    uint64_t rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_loco_params;

typedef struct igx_loco_buffers {
    const float *root_states;         /* [N,13] */
    float *dof_state;                 /* [N,D,2] in/out (rewritten only for reset envs) */
    const float *dof_force;           /* [N,D] Humanoid; NULL for Ant */
    const float *sensors;             /* [N,num_sensor_floats] */
    const float *actions;             /* [N,D] */
    const float *initial_root_states; /* [N,13] read for reset envs only */
    const float *initial_dof_pos;     /* [N,D]  read for reset envs only */
    const float *rand;                /* [N,2D] uniforms in [0,1): D position draws then D velocity draws; NULL = Philox */
    float *obs;                       /* [N,60] Ant / [N,108] Humanoid, out */
    float *potentials;                /* [N] in/out */
    float *prev_potentials;           /* [N] out */
    float *up_vec;                    /* [N,3] out */
    float *heading_vec;               /* [N,3] out */
    float *rew;                       /* [N] out */
    int64_t *reset;                   /* [N] in/out */
    int64_t *progress;                /* [N] in/out */
    int64_t *randomize;               /* [N] in/out, += 1; may be NULL */
    float *true_objective;            /* [N] out (Ant extras['true_objective']); may be NULL */
    /* The tail of VecTask.step (tasks/base/vec_task.py:404, :412) folded into the same launch - both
     * optional, and not for the imitation variants, whose finishing kernel still changes the flags: */
    float *obs_clamped;               /* [N,num_obs] clamp(obs, -clip_obs, clip_obs), what step() returns; may be NULL */
    uint8_t *time_outs;               /* [N] bool (progress >= max_episode_length - 1) & (reset != 0) after the step; may be NULL */
} igx_loco_buffers;

/* Ant.reset_idx (tasks/ant.py:339-366) / Humanoid.reset_idx (tasks/humanoid.py:360-390)
 * for an explicit list of m environment ids: joints re-drawn around
 * initial_dof_pos and clamped to the limits, potentials = prev_potentials from
 * initial_root_states, progress = reset = 0.  Same draws as the flagged resets
 * inside the step (rand, when given, is indexed by environment id); root_states
 * is not touched, as in the reference. */
int igx_ant_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m,
                  igx_stream_t stream);
int igx_humanoid_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m,
                       igx_stream_t stream);

int igx_ant_step(const igx_loco_params *p, const igx_loco_buffers *b,
                 int64_t n_envs, igx_stream_t stream);
int igx_humanoid_step(const igx_loco_params *p, const igx_loco_buffers *b,
                      int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * AntImitate (tasks/ant_imitate.py): the Ant step with the pose-imitation
 * reward (:447-555, JUST_MEASURE_POSE_SIMILARITY branch) in place of the
 * locomotion reward.  post_physics_step (:405-419): target_pose_idx += 1,
 * progress / randomize += 1, flagged resets, observations (as Ant), then
 *   pose_diff = sum_j |obs[12+j] - unscale(motion[target_pose_idx][j])|   (fp64:
 *               the motion comes from np.loadtxt and the arithmetic promotes)
 *   reward = exp(-pose_diff); death_cost below the termination height;
 *   threshold = max(distance_floor, quantile(pose_diff over ALL envs, q));
 *   pose_diff > threshold: reset = 1, reward = death_cost, target_pose_idx = 0;
 *   target_pose_idx >= num_frames - 3: reward = -3 * death_cost, target_pose_idx = 0.
 * The global quantile is an exact radix select (both neighbouring order
 * statistics, torch.quantile's linear interpolation).
 * ---------------------------------------------------------------------- */
typedef struct igx_imitate_args {
    const double *motion;       /* [num_frames, 8] fp64 joint targets */
    int64_t num_frames;
    int64_t *target_pose_idx;   /* [N] in/out */
    double *pose_diff;          /* [N] out */
    void *workspace;            /* IGX_QUANTILE_WORKSPACE_BYTES, 16-byte aligned, any contents */
    double quantile;            /* 0.99 */
    double distance_floor;      /* 1.8 */
    double death_cost;          /* env.deathCost as the double the reference multiplies with */
    int64_t n_total;            /* environments over all shards (== n_envs on one GPU) */
} igx_imitate_args;

/* workspace: eight 64-byte states, then per pass two histograms of 2048 uint64 */
#define IGX_QUANTILE_PASSES 6
#define IGX_QUANTILE_HIST_OFFSET(pass) (512 + (pass) * (2 * 2048 * 8))
#define IGX_QUANTILE_WORKSPACE_BYTES IGX_QUANTILE_HIST_OFFSET(IGX_QUANTILE_PASSES)

/* everything in one call (one GPU) */
int igx_ant_imitate_step(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                         int64_t n_envs, igx_stream_t stream);
/* the pieces, for environment-sharded runs: observe; igx_quantile_begin; then for pass = 0..5
 * { igx_quantile_histogram; all-reduce(sum) over the ranks of the 2 x 2048 uint64 at workspace +
 * IGX_QUANTILE_HIST_OFFSET(pass) }; igx_quantile_finish; igx_ant_imitate_finish.
 * igx_quantile_begin zeroes the histograms and sets the two ranks from q and n_total; the
 * histogram kernel of pass p first picks the bins of pass p - 1 from its (all-reduced) histogram;
 * igx_quantile_finish picks the last digit and interpolates: the value goes to the workspace
 * (where igx_ant_imitate_finish reads it) and to *out unless out is NULL.
 * The optional tail outputs of igx_loco_buffers: obs_clamped is written by the observe call, time_outs by
 * the finish call (it depends on the reset flags the threshold sets). */
int igx_ant_imitate_observe(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                            int64_t n_envs, igx_stream_t stream);
int igx_ant_imitate_finish(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                           int64_t n_envs, igx_stream_t stream);
int igx_quantile_begin(void *workspace, double q, int64_t n_total, igx_stream_t stream);
int igx_quantile_histogram(const double *x, int64_t n, void *workspace, int32_t pass, igx_stream_t stream);
int igx_quantile_finish(void *workspace, double *out, igx_stream_t stream);
/* torch.quantile(x, q) (linear interpolation) of n doubles -> *out (device); NaN if any element is NaN, as torch */
int igx_quantile_f64(const double *x, int64_t n, double q, double *out, void *workspace, igx_stream_t stream);
/* The same value in fewer launches when it is asked for again and again on data that drifts (what igx_ant_imitate_step
 * does every step): the window select of igx_quantile_window_* below on one GPU with no host decision in between - two
 * histogram rounds are enqueued (the second returns at once when the first located a bin of at most 4 096 keys), then
 * collect, sort and interpolate; when that does not resolve the value (a first call on narrow data, more than 4 096
 * copies of the value) the last kernel runs the whole select in ONE workgroup: exact, but about 0.5 ms per 2^20 values.
 * The window lives in `workspace` (same size as above) between calls; any contents are a valid start.  The bits are
 * those of igx_quantile_f64.  n <= 4096 and n >= 2^31 take igx_quantile_f64. */
int igx_quantile_f64_windowed(const double *x, int64_t n, double q, double *out, void *workspace, igx_stream_t stream);

/* The same quantile over the shards of a multi-GPU job in two collectives instead of six
 * (torch.quantile over all environments, ant_imitate.py:527-543).  Every rank histograms its shard over
 * IGX_QUANTILE_WINDOW_BINS bins of a window of the key space that all ranks share (bin 0 = below the window,
 * the last bin = above it; the window starts as the whole key space and, after every call, is centred on the
 * value just found, so that on slowly drifting data one histogram is enough):
 *   igx_quantile_window_begin;
 *   repeat { igx_quantile_window_histogram; all-reduce(sum) of the IGX_QUANTILE_WINDOW_HIST_WORDS uint32 at
 *            workspace + IGX_QUANTILE_WINDOW_HIST_OFFSET; igx_quantile_window_locate; read the 4 x int64
 *            decision {status, count, bin, round} } while status == IGX_QW_REFINED   (the bin that holds
 *            the wanted rank held more than `capacity` elements: the window shrank to that bin);
 *   status IGX_QW_COLLECT (or IGX_QW_VALUE: the bin is one single value, count = 0):
 *            igx_quantile_window_collect writes {local count, smallest key above the bin, the shard's keys in
 *            the bin} into `slots` (count + 2 uint64); all-gather the slots of all ranks;
 *            igx_quantile_window_finish sorts the gathered keys, picks the two order statistics, interpolates
 *            as igx_quantile_finish does and re-centres the window;
 *   status IGX_QW_GIVE_UP (more than IGX_QUANTILE_WINDOW_MAX_ROUNDS histograms): use the six-pass select.
 * The result has the bits of igx_quantile_f64 over the concatenated shards.  n_total < 2^31.
 * `reset_window` != 0 (or a workspace that igx_quantile_window_begin has not seen) starts from the whole
 * key space; all ranks must pass the same value. */
#define IGX_QUANTILE_WINDOW_BINS 16384
#define IGX_QUANTILE_WINDOW_HIST_WORDS (IGX_QUANTILE_WINDOW_BINS + 32)  /* [BINS] = number of NaNs */
#define IGX_QUANTILE_WINDOW_HIST_OFFSET 128
#define IGX_QUANTILE_WINDOW_WORKSPACE_BYTES (IGX_QUANTILE_WINDOW_HIST_OFFSET + 4 * IGX_QUANTILE_WINDOW_HIST_WORDS)
#define IGX_QUANTILE_WINDOW_MAX_CANDIDATES 4096
#define IGX_QUANTILE_WINDOW_MAX_ROUNDS 12
#define IGX_QUANTILE_WINDOW_MAX_RANKS 1024
#define IGX_QW_COLLECT 0
#define IGX_QW_REFINED 1
#define IGX_QW_VALUE 2
#define IGX_QW_GIVE_UP 4
int igx_quantile_window_begin(void *workspace, double q, int64_t n_total, int32_t reset_window, igx_stream_t stream);
int igx_quantile_window_histogram(const double *x, int64_t n, void *workspace, igx_stream_t stream);
/* capacity: 1..IGX_QUANTILE_WINDOW_MAX_CANDIDATES (all ranks the same); decision: 4 x int64 on the device */
int igx_quantile_window_locate(void *workspace, int32_t capacity, int64_t *decision, igx_stream_t stream);
/* slots: num_slots uint64 on the device, num_slots >= 2 + the decision's count */
int igx_quantile_window_collect(const double *x, int64_t n, const void *workspace, uint64_t *slots, int64_t num_slots,
                                igx_stream_t stream);
/* gathered: [num_ranks][num_slots] uint64, rank order; *out (device, may be NULL) receives the value, which
 * also stays in the workspace; to hand it to igx_ant_imitate_finish pass out = (double *)quantile_workspace + 5 */
int igx_quantile_window_finish(void *workspace, const uint64_t *gathered, int32_t num_ranks, int64_t num_slots,
                               double *out, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * HumanoidImitate (tasks/humanoid_imitate.py): 63-dof / 22-body humanoid.
 * post_physics_step (:634-643): progress / randomize += 1, flagged resets
 * (reset_idx :579-605, as Humanoid), observations (:795-831)
 *   obs[205] = [unscale(dof_pos) 63 | dof_vel * dofVelocityScale 63 | root row 13 | body positions 22 x 3],
 * reward (:672-758) against one frame of the motion file (fp64 where the
 * fp64 motion table enters):
 *   exp(-0.2 sum|body_pos - target_pos|) + exp(-0.3 sum|obs[0:63] - unscale(target_pose)|)
 *   + exp(-0.05 sum|obs[63:126]|) + (root_z > termination_height ? up_weight : 0)
 *   + 0.1 exp(-|root_xy|) - actions_cost_scale * sum(a^2);
 * reset on root_z < termination_height, x^2 + y^2 > termination_radius^2, or time-out.
 * ---------------------------------------------------------------------- */
#define IGX_HIM_DOF 63
#define IGX_HIM_BODIES 22
#define IGX_HIM_OBS 205

typedef struct igx_humanoid_imitate_params {
    float dt;
    float dof_vel_scale;
    float up_weight;
    float actions_cost_scale;
    float termination_height;
    float termination_radius_sq;  /* float32(termination_radius ** 2) */
    float max_episode_length_m1;
    int32_t reserved;
    float targets[3];
    float clip_obs;               /* env.clipObservations (+inf copies); read only with obs_clamped */
    float dof_limits_lower[IGX_HIM_DOF + 1];
    float dof_limits_upper[IGX_HIM_DOF + 1];
    uint64_t seed;
    uint64_t rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_humanoid_imitate_params;

typedef struct igx_humanoid_imitate_buffers {
    const float *root_states;          /* [N,13] */
    float *dof_state;                  /* [N,63,2] */
    const float *rigid_body_states;    /* [N,22,13] */
    const float *actions;              /* [N,63] */
    const float *initial_root_states;  /* [N,13] */
    const float *initial_dof_pos;      /* [N,63] */
    const float *rand;                 /* [N,126] or NULL (in-kernel Philox) */
    const double *motion_frame;        /* [63 + 22*13] fp64: the frame to imitate (motion_to_imitate[41]) */
    float *obs;                        /* [N,205] */
    float *potentials;                 /* [N] written for reset envs only */
    float *prev_potentials;            /* [N] " */
    float *rew;                        /* [N] */
    int64_t *reset, *progress, *randomize; /* [N]; randomize may be NULL */
    /* optional: the tail of VecTask.step (tasks/base/vec_task.py:404, :412) in the same launch */
    float *obs_clamped;                /* [N,205] clamp(obs, -clip_obs, clip_obs) */
    uint8_t *time_outs;                /* [N] bool (progress >= episodeLength - 1) & (reset != 0) after the step */
} igx_humanoid_imitate_buffers;

int igx_humanoid_imitate_step(const igx_humanoid_imitate_params *p, const igx_humanoid_imitate_buffers *b,
                              int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * AllegroHand in-hand reorientation.
 * Two entry points, as the reference splits the work (tasks/allegro_hand.py):
 *  igx_allegro_pre_physics  = pre_physics_step :586-615 with reset_target_pose
 *      :506-522, reset_idx :524-584 and randomize_rotation(_pen) :722-732
 *      (resets happen BEFORE the physics step in this task);
 *  igx_allegro_post_physics = post_physics_step :627-632 with
 *      compute_observations :406-435, compute_full_state :462-504,
 *      compute_full_observations :437-460 and compute_hand_reward :663-719,
 *      including the cross-environment update of consecutive_successes.
 * Actors per env in root_state_tensor: hand, object, goal (rows addressed by
 * object_indices / goal_object_indices).  All 16 hand joints are actuated.
 * ---------------------------------------------------------------------- */
#define IGX_ALLEGRO_DOF 16
#define IGX_ALLEGRO_OBS_FULL_NO_VEL 0 /* 50 columns */
#define IGX_ALLEGRO_OBS_FULL 1        /* 72 columns */
#define IGX_ALLEGRO_OBS_FULL_STATE 2  /* 88 columns (default) */
#define IGX_ALLEGRO_RAND_PER_ENV 45   /* 4 goal-only draws, 37 env-reset draws, 4 goal draws inside the env reset */

typedef struct igx_allegro_params {
    int32_t obs_type;                  /* IGX_ALLEGRO_OBS_* */
    int32_t max_consecutive_successes; /* env.maxConsecutiveSuccesses, 0 */
    int32_t ignore_z_rot;              /* object_type == "pen" */
    int32_t use_relative_control;      /* env.useRelativeControl */
    float dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale;
    float success_tolerance;   /* already doubled for the pen */
    float success_threshold_c; /* largest c with |2*asin(c)| <= success_tolerance in the caller's libm:
                                  the success decision is taken on c = min(|quat_diff.xyz|, 1) */
    float reach_goal_bonus, fall_dist, fall_penalty;
    float max_episode_length_m1;
    float av_factor, one_minus_av_factor; /* float32(av), float32(1.0 - double(float32(av))) */
    float vel_obs_scale;          /* 0.2 */
    float force_torque_obs_scale; /* 10.0 */
    float reset_position_noise, reset_dof_pos_noise, reset_dof_vel_noise;
    float act_moving_average, one_minus_act_moving_average;
    float dof_speed_scale_dt;     /* dofSpeedScale * dt (relative control) */
    float goal_displacement[3];   /* (-0.2, -0.06, 0.12) */
    float dof_lower[IGX_ALLEGRO_DOF], dof_upper[IGX_ALLEGRO_DOF];
    float clip_obs;               /* env.clipObservations (+inf copies); read only with obs_clamped / states_clamped */
    float clip_actions;           /* pre_physics with actions_out: env.clipActions (+inf copies) */
    uint64_t seed, rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_allegro_params;

/* scratch for the cross-environment reduction: 16-byte records - the total, then one per tile of
 * at least 16 environments */
#define IGX_ALLEGRO_WORKSPACE_BYTES(n_envs) (16 * (((int64_t)(n_envs) + 15) / 16 + 1))

typedef struct igx_allegro_buffers {
    float *dof_state;                 /* [N,16,2] in (post) / in-out (pre: reset envs) */
    const float *dof_force;           /* [N,16]; needed for the 88-column layouts */
    float *root_state_tensor;         /* [3N,13]; pre: object / goal rows rewritten on reset */
    const int64_t *object_indices;    /* [N] rows of root_state_tensor */
    const int64_t *goal_object_indices; /* [N] */
    float *goal_states;               /* [N,13] */
    const float *goal_init_state;     /* [N,13] */
    const float *object_init_state;   /* [N,13] */
    const float *actions;             /* [N,16] */
    float *prev_targets, *cur_targets; /* [N,16] */
    const float *rand;                /* [N,45] uniforms in [0,1); NULL = Philox */
    float *obs;                       /* [N,50|72|88] out */
    float *states;                    /* [N,88] out, asymmetric observations; may be NULL */
    float *rew;                       /* [N] out */
    int64_t *reset, *reset_goal, *progress; /* [N] in/out */
    int64_t *randomize;               /* [N] += 1 in post; may be NULL */
    float *successes;                 /* [N] in/out */
    float *consecutive_successes;     /* [1] in/out */
    void *workspace;                  /* IGX_ALLEGRO_WORKSPACE_BYTES(n_envs) of device scratch, 16-byte aligned,
                                         contents irrelevant on entry (per-tile partial sums of the reduction) */
    /* post only, all optional: the tail of VecTask.step (tasks/base/vec_task.py:404, :412-416) in the same launch */
    float *obs_clamped;               /* [N,num_obs] clamp(obs, -clip_obs, clip_obs) */
    float *states_clamped;            /* [N,88] clamp(states, ...); needs states */
    uint8_t *time_outs;               /* [N] bool (progress >= episodeLength - 1) & (reset != 0) after the step */
    void *shard_record;               /* igx_*_post_physics_shard only, optional: the call's 16-byte record is ALSO written here
                                         (a slot of a caller-side queue of records, one per step) */
    float *actions_out;               /* pre_physics only, optional: the clamp of VecTask.step (tasks/base/vec_task.py:384) folded
                                         into this launch - `actions` then holds what step() was handed, clamp(actions,
                                         -clip_actions, clip_actions) is written here (the task's self.actions) and is what
                                         the targets are computed from.  NULL: `actions` is used as it is. */
} igx_allegro_buffers;

int igx_allegro_pre_physics(const igx_allegro_params *p, const igx_allegro_buffers *b,
                            int64_t n_envs, igx_stream_t stream);
int igx_allegro_post_physics(const igx_allegro_params *p, const igx_allegro_buffers *b,
                             int64_t n_envs, igx_stream_t stream);

/* igx_allegro_post_physics for callers that cannot allocate device memory and bring no workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No
 * counterpart in the reference.  b->workspace is ignored: the per-tile records of the reduction
 * live in a file-scope device array of IGX_ALLEGRO_STATIC_SCRATCH_BYTES inside the library, one
 * copy per GPU, held until the process ends.  n_envs above IGX_ALLEGRO_STATIC_MAX_ENVS is
 * IGX_E_SIZE.  The call allocates and frees nothing and gives the same results as
 * igx_allegro_post_physics.  Every call on one GPU uses that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; calls on one stream, or ordered by events, do not.
 * The array's device address is looked up once per device: by igx_allegro_static_prepare() (a set-up
 * call for the current device, no counterpart in the reference either; run it outside stream capture),
 * or else by the first igx_allegro_post_physics_static call on that device - which therefore must not be
 * made under capture (it is refused with IGX_E_PARAM if it is).  After that the entry point may be
 * captured into a HIP graph and replayed. */
#define IGX_ALLEGRO_STATIC_MAX_ENVS ((int64_t)1 << 21)
#define IGX_ALLEGRO_STATIC_SCRATCH_BYTES (16 * ((1 << 21) / 16 + 1)) /* = IGX_ALLEGRO_WORKSPACE_BYTES(max) */
int igx_allegro_static_prepare(void);
int igx_allegro_post_physics_static(const igx_allegro_params *p, const igx_allegro_buffers *b,
                                    int64_t n_envs, igx_stream_t stream);

/* Environment-sharded form of igx_allegro_post_physics (one shard per GPU).
 * compute_hand_reward's consecutive_successes (tasks/allegro_hand.py:714-719)
 * averages over ALL environments, so a shard stops after leaving its 16-byte
 * record {uint64 num_resets; double sum(successes * resets)} at b->workspace;
 * the caller gathers the records of all shards (rank order) and every rank
 * calls igx_allegro_finalize_shards, which adds them in index order and updates
 * consecutive_successes[0]. */
#define IGX_ALLEGRO_SHARD_RECORD_BYTES 16
int igx_allegro_post_physics_shard(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n_envs,
                                   igx_stream_t stream);
int igx_allegro_finalize_shards(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                int32_t num_shards, igx_stream_t stream);
/* The same for num_steps queued steps at once, so that the all-gather can leave the step's critical path:
 * every shard call leaves its record in slot t of a queue (b->shard_record), the queues of all shards are
 * gathered once per horizon (on a side stream), and consecutive_successes is advanced step by step, in step
 * order - bit for bit what num_steps calls of igx_allegro_finalize_shards give.
 * records: [num_shards][num_steps] 16-byte records (shard-major, as an all-gather of the queues lays them out). */
int igx_allegro_finalize_shards_queued(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                       int32_t num_shards, int32_t num_steps, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * ShadowHand in-hand reorientation (tasks/shadow_hand.py): the same task family as
 * AllegroHand - compute_hand_reward :747-800 is the same function - with a 24-joint hand
 * of which 20 joints are actuated (actuated_dof_indices), five fingertip bodies read out
 * of rigid_body_states (fingertip_handles), fingertip force-torque sensors, and four
 * observation layouts.
 *  igx_shadow_pre_physics  = pre_physics_step :670-698 with reset_target_pose :586-602,
 *      reset_idx :604-668 and randomize_rotation(_pen) :803-813 (forceScale 0: no random forces);
 *  igx_shadow_post_physics = post_physics_step :710-715 with compute_observations :437-471,
 *      compute_fingertip_observations :473-496, compute_full_observations :498-526,
 *      compute_full_state :528-584 and compute_hand_reward.
 * The workspace, the shard record and the finalize step are AllegroHand's.
 * ---------------------------------------------------------------------- */
#define IGX_SHADOW_DOF 24
#define IGX_SHADOW_ACTIONS 20
#define IGX_SHADOW_FINGERTIPS 5
#define IGX_SHADOW_OBS_OPENAI 0      /* 42 columns  (:481-485) */
#define IGX_SHADOW_OBS_FULL_NO_VEL 1 /* 77 columns  (:500-510) */
#define IGX_SHADOW_OBS_FULL 2        /* 157 columns (:512-526) */
#define IGX_SHADOW_OBS_FULL_STATE 3  /* 211 columns (:558-584), also the asymmetric states_buf */
#define IGX_SHADOW_RAND_PER_ENV 61   /* 4 goal-only draws, 53 env-reset draws, 4 goal draws inside the env reset */

typedef struct igx_shadow_params {
    int32_t obs_type;                  /* IGX_SHADOW_OBS_* */
    int32_t max_consecutive_successes;
    int32_t ignore_z_rot;              /* object_type == "pen" */
    int32_t use_relative_control;
    float dist_reward_scale, rot_reward_scale, rot_eps, action_penalty_scale;
    float success_tolerance;           /* already doubled for the pen */
    float success_threshold_c;         /* as in igx_allegro_params */
    float reach_goal_bonus, fall_dist, fall_penalty;
    float max_episode_length_m1;
    float av_factor, one_minus_av_factor;
    float vel_obs_scale;               /* 0.2 */
    float force_torque_obs_scale;      /* 10.0 */
    float reset_position_noise, reset_dof_pos_noise, reset_dof_vel_noise;
    float act_moving_average, one_minus_act_moving_average;
    float dof_speed_scale_dt;
    float goal_displacement[3];
    int32_t num_bodies;                /* rigid bodies per environment (rows of rigid_body_states per env) */
    float dof_lower[IGX_SHADOW_DOF], dof_upper[IGX_SHADOW_DOF];
    int32_t actuated_dof_indices[IGX_SHADOW_ACTIONS]; /* strictly increasing, in [0, 24) */
    int32_t fingertip_handles[IGX_SHADOW_FINGERTIPS]; /* body indices in [0, num_bodies) */
    float clip_obs;                    /* as in igx_allegro_params */
    float clip_actions;           /* pre_physics with actions_out: env.clipActions (+inf copies) */
    uint64_t seed, rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_shadow_params;

typedef struct igx_shadow_buffers {
    float *dof_state;                 /* [N,24,2] */
    const float *dof_force;           /* [N,24]; full_state / asymmetric */
    const float *vec_sensor_tensor;   /* [N,30]; full_state / asymmetric */
    const float *rigid_body_states;   /* [N,num_bodies,13] */
    float *root_state_tensor;         /* [3N,13] */
    const int64_t *object_indices;    /* [N] */
    const int64_t *goal_object_indices; /* [N] */
    float *goal_states;               /* [N,13] */
    const float *goal_init_state;     /* [N,13] */
    const float *object_init_state;   /* [N,13] */
    const float *actions;             /* [N,20] */
    float *prev_targets, *cur_targets; /* [N,24] */
    const float *rand;                /* [N,61] uniforms in [0,1); NULL = Philox */
    float *obs;                       /* [N,42|77|157|211] out */
    float *states;                    /* [N,211] out, asymmetric observations; may be NULL */
    float *rew;                       /* [N] out */
    int64_t *reset, *reset_goal, *progress; /* [N] in/out */
    int64_t *randomize;               /* [N] += 1 in post; may be NULL */
    float *successes;                 /* [N] in/out */
    float *consecutive_successes;     /* [1] in/out */
    void *workspace;                  /* IGX_ALLEGRO_WORKSPACE_BYTES(n_envs), 16-byte aligned, any contents */
    float *obs_clamped;               /* as in igx_allegro_buffers */
    float *states_clamped;            /* [N,211] */
    uint8_t *time_outs;
    void *shard_record;               /* igx_*_post_physics_shard only, optional: the call's 16-byte record is ALSO written here
                                         (a slot of a caller-side queue of records, one per step) */
    float *actions_out;               /* pre_physics only, optional: the clamp of VecTask.step (tasks/base/vec_task.py:384) folded
                                         into this launch - `actions` then holds what step() was handed, clamp(actions,
                                         -clip_actions, clip_actions) is written here (the task's self.actions) and is what
                                         the targets are computed from.  NULL: `actions` is used as it is. */
} igx_shadow_buffers;

int igx_shadow_pre_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n_envs, igx_stream_t stream);
int igx_shadow_post_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n_envs, igx_stream_t stream);
/* environment-sharded form, as igx_allegro_post_physics_shard / igx_allegro_finalize_shards */
int igx_shadow_post_physics_shard(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n_envs,
                                  igx_stream_t stream);
int igx_shadow_finalize_shards(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records,
                               int32_t num_shards, igx_stream_t stream);
int igx_shadow_finalize_shards_queued(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records,
                                      int32_t num_shards, int32_t num_steps, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * Random disturbance forces on the manipulated object of AllegroHand / ShadowHand: everything
 * pre_physics_step does to rb_forces and random_force_prob around
 * apply_rigid_body_force_tensors (tasks/allegro_hand.py:532, :558-559, :617-625;
 * tasks/shadow_hand.py:616, :642-643, :700-708).  Per environment, in the reference's order:
 *   reset[e] != 0:  rb_forces[e, :, :] = 0;
 *                   random_force_prob[e] = exp((log_prob_lo - log_prob_hi) * u_prob + log_prob_hi)
 *   rb_forces[e, :, :] *= decay
 *   u_sel < random_force_prob[e]:  rb_forces[e, object_rb_handle, :] = (z * object_rb_mass) * force_scale
 * Call it BEFORE igx_allegro_pre_physics / igx_shadow_pre_physics of the same step, which clear the
 * reset flags it reads (nothing else is shared between the two calls).  The object is one rigid body,
 * as in the reference's assets.  rand, when given, holds per environment
 * (u_prob, u_sel, z0, z1, z2): two uniforms in [0,1) and three standard normals - torch's own
 * draws scattered to their environments reproduce the reference bit for bit (random_force_prob to
 * the last bit of exp); NULL = the library's Philox4x32-10 stream with Box-Muller normals.
 * ---------------------------------------------------------------------- */
#define IGX_HAND_FORCE_RAND_PER_ENV 5
typedef struct igx_hand_forces {
    float *rb_forces;            /* [N, num_bodies, 3] in/out */
    float *random_force_prob;    /* [N] in/out */
    const int64_t *reset;        /* [N] reset_buf as pre_physics_step finds it */
    const float *rand;           /* [N,5] or NULL */
    int32_t num_bodies;          /* rigid bodies per environment */
    int32_t object_rb_handle;    /* the object's body index in [0, num_bodies) */
    float object_rb_mass;
    float force_scale;           /* env.forceScale */
    float decay;                 /* float32 pow(env.forceDecay, dt / env.forceDecayInterval) */
    float log_prob_lo, log_prob_hi; /* float32 log of env.forceProbRange[0], [1] */
    int32_t reserved;
    uint64_t seed, rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_hand_forces;

int igx_hand_random_forces(const igx_hand_forces *f, int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * HumanoidAMP and the AMP motion library.  The character is the reference's
 * 15-body / 28-dof AMP humanoid: DOF_BODY_IDS, DOF_OFFSETS and the four key
 * bodies are fixed as in tasks/amp/humanoid_amp_base.py:43-49.
 * ---------------------------------------------------------------------- */
#define IGX_AMP_NUM_BODIES 15
#define IGX_AMP_NUM_DOF 28
#define IGX_AMP_NUM_KEY_BODIES 4
#define IGX_AMP_OBS 105 /* 1 + 6 + 3 + 3 + 52 + 28 + 12 */

/* Device-resident per-frame tables of every loaded motion, concatenated over
 * motions (frame f of motion m is row frame_offset[m] + f).  Built once at
 * load time by the host (tasks/amp/utils_amp/motion_lib.py:155-206, :248-324
 * and the forward kinematics of poselib/skeleton/skeleton3d.py:431-472). */
typedef struct igx_motion_tables {
    int32_t num_motions;
    int32_t reserved;
    const double *motion_length;      /* [M] (F-1)/fps, fp64 as in the reference */
    const double *motion_dt;          /* [M] 1/fps */
    const int64_t *motion_num_frames; /* [M] */
    const int64_t *frame_offset;      /* [M] */
    const float *root_pos;            /* [sumF,3]   global_translation[:,0] */
    const float *root_rot;            /* [sumF,4]   global_rotation[:,0] */
    const float *local_rot;           /* [sumF,15,4] local_rotation */
    const float *root_vel;            /* [sumF,3]   global_root_velocity */
    const float *root_ang_vel;        /* [sumF,3]   global_root_angular_velocity */
    const float *key_pos;             /* [sumF,4,3] global_translation[:, key_body_ids] */
    const float *dof_vel;             /* [sumF,28]  dof_vels */
} igx_motion_tables;

typedef struct igx_motion_state_out { /* every pointer may be NULL to skip that output */
    float *root_pos;      /* [n,3] */
    float *root_rot;      /* [n,4] */
    float *dof_pos;       /* [n,28] */
    float *root_vel;      /* [n,3] */
    float *root_ang_vel;  /* [n,3] */
    float *dof_vel;       /* [n,28] */
    float *key_pos;       /* [n,4,3] */
    int64_t *frame_idx0;  /* [n] */
    int64_t *frame_idx1;  /* [n] */
    float *blend;         /* [n] */
} igx_motion_state_out;

/* MotionLib.get_motion_state (tasks/amp/utils_amp/motion_lib.py:83-153) with
 * _calc_frame_blend (:233-241, fp64) and _local_rotation_to_dof (:265-293).
 * A motion id outside [0, num_motions) is a hole: that sample's output rows are left untouched
 * (here and in igx_motion_amp_obs), so that a masked list can be processed without the host. */
int igx_motion_state(const igx_motion_tables *t, const int64_t *motion_ids, const double *motion_times,
                     int64_t n, const igx_motion_state_out *out, igx_stream_t stream);

/* Reference-state sampling for resets without a round trip to the host.  The reference finds the
 * flagged environments with nonzero() and samples on the host with numpy (tasks/humanoid_amp.py:220-246:
 * MotionLib.sample_motions = np.random.choice(m, p=weights), tasks/amp/utils_amp/motion_lib.py:61-65;
 * sample_time = uniform phase * motion length, :67-78).  Here, for every environment e:
 *   reset[e] != 0 and u2 < ref_init_prob (a reference-state reset, :220-246):
 *                  env_ids[e] = e, motion_ids[e] = first i with u0 < cdf[i], motion_times[e] = u1 * length
 *                  (random_time != 0, stateInit Random / Hybrid) or 0 (stateInit Start);
 *   reset[e] != 0 otherwise (a default-pose reset, :206-218; stateInit Default is ref_init_prob = 0,
 *                  Hybrid is hybridInitProb, :248-261): default_env_ids[e] = e;
 *   everything else is a hole: env_ids / motion_ids / default_env_ids = -1, motion_times = 0, which
 *                  igx_motion_state, igx_motion_amp_obs and igx_humanoid_amp_reset skip.
 * u0, u1, u2 come from the library's Philox4x32-10 stream keyed by (seed, rng_offset, e): the distributions
 * are the reference's, the stream is not numpy's / torch's.  default_env_ids may be NULL when
 * ref_init_prob >= 1. */
typedef struct igx_motion_sampler {
    const double *cdf;              /* [num_motions] cumulative weights / their total (device memory) */
    int32_t random_time;
    float ref_init_prob;            /* 1: every reset takes a reference state */
    uint64_t seed, rng_offset;
    const uint64_t *rng_offset_dev; /* as in igx_cartpole_params */
    int64_t env_offset;             /* index of this call's first environment in the whole job (0 on one GPU): added to the
                                       environment index in the Philox counter only, so that a shard draws what the
                                       unsharded run draws for the same environments */
} igx_motion_sampler;
int igx_motion_sample_flagged(const igx_motion_tables *t, const igx_motion_sampler *s, const int64_t *reset, int64_t n,
                              int64_t *env_ids, int64_t *motion_ids, double *motion_times, int64_t *default_env_ids,
                              igx_stream_t stream);

/* build_amp_observations (tasks/humanoid_amp.py:337-371) == compute_humanoid_observations
 * (tasks/amp/humanoid_amp_base.py:542-576) with dof_to_obs (:510-540).
 * root_states [n,13], dof_pos [n,28], dof_vel [n,28], key_body_pos [n,4,3] -> obs [n,105]. */
int igx_amp_observations(const float *root_states, const float *dof_pos, const float *dof_vel,
                         const float *key_body_pos, int32_t local_root_obs, float *obs, int64_t n,
                         igx_stream_t stream);

/* get_motion_state + build_amp_observations fused (HumanoidAMP.fetch_amp_obs_demo,
 * tasks/humanoid_amp.py:146-171, and _init_amp_obs_ref :279-294): sample i is written
 * to out + (out_rows ? out_rows[i] : i) * out_row_stride + out_col_offset (floats). */
int igx_motion_amp_obs(const igx_motion_tables *t, const int64_t *motion_ids, const double *motion_times,
                       int64_t n, int32_t local_root_obs, float *out, const int64_t *out_rows,
                       int64_t out_row_stride, int64_t out_col_offset, igx_stream_t stream);

/* HumanoidAMP.fetch_amp_obs_demo (tasks/humanoid_amp.py:146-171) after its two draws, in one launch: for drawn
 * sample i (motion_ids[i], motion_times[i], i < n) and history step k < num_steps the observation of the motion at
 * motion_times[i] + (-step_dt * k) (fp64, the operations of :156-159: np.tile of the ids, times0 + (-dt * arange))
 * is written to out[(i * num_steps + k) * 105 ...]: out is the [n, num_steps * 105] batch the learner receives.
 * Equals igx_motion_amp_obs on the tiled lists bit for bit; the host sends n ids and times instead of n * num_steps. */
int igx_motion_amp_obs_steps(const igx_motion_tables *t, const int64_t *motion_ids, const double *motion_times, int64_t n,
                             int32_t num_steps, double step_dt, int32_t local_root_obs, float *out, igx_stream_t stream);

/* The AMP history of environments that were reset to a reference motion (HumanoidAMP._init_amp_obs_ref,
 * tasks/humanoid_amp.py:279-294): for list entry i < n and history step k, first_step <= k < num_steps, the observation
 * of motion_ids[i] at motion_times[i] + (-step_dt * k) is written to row (env_ids ? env_ids[i] : i) * num_steps + k of
 * amp_obs [num_envs, num_steps, 105].  The reference fills steps 1 ... num_steps - 1 (first_step = 1; step 0 is the
 * environment's own observation).  Entries with a motion id outside [0, num_motions) are holes and are skipped (the
 * masked lists of igx_motion_sample_flagged).  Equals igx_motion_amp_obs on the tiled lists with explicit rows, bit
 * for bit, without the tiling on the host. */
int igx_motion_amp_obs_history(const igx_motion_tables *t, const int64_t *motion_ids, const double *motion_times,
                               const int64_t *env_ids, int64_t n, int32_t num_steps, int32_t first_step, double step_dt,
                               int32_t local_root_obs, float *amp_obs, igx_stream_t stream);

/* The two draws of fetch_amp_obs_demo on the device (MotionLib.sample_motions / sample_time,
 * tasks/amp/utils_amp/motion_lib.py:61-78): motion_ids[i] = first j with u0 < cdf[j], motion_times[i] = u1 * length
 * (0 with random_time == 0), u0 / u1 from the Philox stream keyed by (seed, rng_offset, i + env_offset) as in
 * igx_motion_sample_flagged with every flag set; ref_init_prob is ignored.  The distributions are the
 * reference's, the stream is not numpy's. */
int igx_motion_sample(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int64_t *motion_ids,
                      double *motion_times, igx_stream_t stream);

/* fetch_amp_obs_demo (tasks/humanoid_amp.py:146-171) whole, in one launch: the draws of igx_motion_sample and the
 * batch of igx_motion_amp_obs_steps on them.  out: [n, num_steps * 105]; motion_ids / motion_times (each may be NULL)
 * receive the n draws.  Bit for bit what the two calls give one after the other with the same sampler. */
int igx_motion_fetch_amp_obs(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int32_t num_steps,
                             double step_dt, int32_t local_root_obs, float *out, int64_t *motion_ids, double *motion_times,
                             igx_stream_t stream);

typedef struct igx_amp_params {
    int32_t num_amp_obs_steps;        /* env.numAMPObsSteps >= 2 */
    int32_t local_root_obs;           /* env.localRootObs */
    int32_t enable_early_termination; /* env.enableEarlyTermination */
    int32_t num_contact_bodies;       /* <= 4 */
    int32_t contact_body_ids[4];      /* feet: 11, 14 */
    int32_t key_body_ids[4];          /* 5, 8, 11, 14 */
    float termination_height;         /* 0.5 */
    float max_episode_length_m1;      /* episodeLength - 1 */
    float clip_obs;                   /* env.clipObservations (+inf copies); read only with obs_clamped */
    int32_t reserved;
} igx_amp_params;

typedef struct igx_amp_buffers {
    float *root_states;             /* [N,13] */
    float *dof_state;               /* [N,28,2] */
    const float *rigid_body_state;  /* [N,15,13] */
    const float *contact_forces;    /* [N,15,3] */
    float *obs;                     /* [N,105] */
    float *amp_obs;                 /* [N,K,105] newest first */
    float *rew;                     /* [N] */
    int64_t *reset, *terminate, *progress; /* [N] */
    /* step only, optional: the tail of VecTask.step (tasks/base/vec_task.py:404, :412) in the same launch */
    float *obs_clamped;             /* [N,105] clamp(obs, -clip_obs, clip_obs) */
    uint8_t *time_outs;             /* [N] bool (progress >= episodeLength - 1) & (reset != 0) after the step */
} igx_amp_buffers;

/* HumanoidAMP.post_physics_step (tasks/humanoid_amp.py:125-134 over
 * tasks/amp/humanoid_amp_base.py:424-438): progress += 1, observations,
 * reward (== 1), compute_humanoid_reset (:584-609), AMP history shift
 * (:312-319) and current AMP observation (:321-330). */
int igx_humanoid_amp_step(const igx_amp_params *p, const igx_amp_buffers *b, int64_t n_envs, igx_stream_t stream);

/* reset_idx for env_ids [m] (tasks/humanoid_amp.py:184-204, :296-310 _set_env_state,
 * tasks/amp/humanoid_amp_base.py:165-169): scatter the given per-env state rows
 * (root_pos [m,3], root_rot [m,4], dof_pos [m,28], root_vel [m,3], root_ang_vel [m,3],
 * dof_vel [m,28]) into root_states / dof_state, zero progress / reset / terminate,
 * then recompute obs[env_ids] and amp_obs[env_ids, 0] from the (not yet refreshed)
 * rigid_body_state, and copy it into every history slot (the _init_amp_obs_default
 * behaviour; reference-state resets overwrite the history afterwards with
 * igx_motion_amp_obs).  A negative entry of env_ids is a hole and is skipped. */
int igx_humanoid_amp_reset(const igx_amp_params *p, const igx_amp_buffers *b, const int64_t *env_ids, int64_t m,
                           const igx_motion_state_out *state, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * Observation / action noise of the domain-randomisation block
 * (tasks/base/vec_task.py:693-703 gaussian, :720-729 uniform noise_lambda):
 *   noise = ((corr * corr_scale + corr_shift) + fresh * scale) + shift
 *   out   = x + noise (additive)  or  x * noise (scaling)
 * one fp32 rounding per operation.  gaussian: corr_scale = var_corr, corr_shift =
 * mu_corr, scale = var, shift = mu; uniform: corr_scale = hi_corr - lo_corr,
 * corr_shift = lo_corr, scale = hi - lo, shift = lo (differences taken in double,
 * then rounded to fp32, as the reference's Python floats are).  `corr` is the
 * draw the reference keeps in params['corr'] from the first call on (always
 * standard normal, :697 / :724); `fresh` is this call's randn_like / rand_like.
 * With fresh == NULL the kernel draws it itself (Philox4x32-10 keyed by seed,
 * counter = (element index / 4, offset); Box-Muller for the gaussian), which
 * saves writing and re-reading one tensor per step; the stream then differs
 * from torch's, the distribution does not.  corr == NULL switches the noise off.
 * ---------------------------------------------------------------------- */
#define IGX_NOISE_ADDITIVE 0
#define IGX_NOISE_SCALING 1
#define IGX_NOISE_GAUSSIAN 0
#define IGX_NOISE_UNIFORM 1
typedef struct igx_noise {
    const float *corr;     /* [count] or NULL = no noise */
    const float *fresh;    /* [count] or NULL = drawn in the kernel */
    float corr_scale, corr_shift, scale, shift;
    int32_t operation;     /* IGX_NOISE_ADDITIVE / IGX_NOISE_SCALING */
    int32_t distribution;  /* IGX_NOISE_GAUSSIAN / IGX_NOISE_UNIFORM; read only when fresh == NULL */
    uint64_t seed;
    uint64_t offset;       /* advance by one per call */
    const uint64_t *offset_dev; /* optional device-resident counter added to offset (for replayed graphs); NULL = 0 */
} igx_noise;

/* out = noise_lambda(x) on count floats; out may alias x. */
int igx_noise_apply(const igx_noise *nz, const float *x, float *out, int64_t count, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * What happens to the actions between VecTask.step's entry and the physics step:
 * action noise (tasks/base/vec_task.py:381-382), torch.clamp(+-clip_actions)
 * (:384), the copy into self.actions and the task's scaling into joint forces
 * or PD targets, one launch.
 *   IGX_ACT_FORCE     targets = (a * gain[j]) * power_scale
 *                     (tasks/ant.py:369-370 joint_gears, tasks/humanoid.py:393-394 and
 *                     tasks/amp/humanoid_amp_base.py:418 motor_efforts)
 *   IGX_ACT_PD_TARGET targets = offset[j] + gain[j] * a
 *                     (tasks/amp/humanoid_amp_base.py:413, :467-469 _action_to_pd_targets)
 *   IGX_ACT_FIRST_DOF targets[i, 0] = a[i] * power_scale, targets[i, 1:] = 0; num_actions 1
 *                     (tasks/cartpole.py:164-166, power_scale = max_push_effort)
 * ---------------------------------------------------------------------- */
#define IGX_ACT_FORCE 0
#define IGX_ACT_PD_TARGET 1
#define IGX_ACT_FIRST_DOF 2
typedef struct igx_action_stage {
    const float *actions;      /* [N,num_actions] as handed to step() */
    float *actions_out;        /* [N,num_actions] self.actions (after noise and clamp); NULL skips */
    float *targets;            /* [N,num_actions]; IGX_ACT_FIRST_DOF: [N,num_dof] */
    const float *gain;         /* [num_actions]; unused by IGX_ACT_FIRST_DOF */
    const float *offset;       /* [num_actions], IGX_ACT_PD_TARGET only */
    igx_noise noise;           /* over the N*num_actions actions */
    float clip_actions;        /* env.clipActions; +inf copies */
    float power_scale;
    int32_t mode;
    int32_t num_actions;
    int32_t num_dof;           /* IGX_ACT_FIRST_DOF: dofs per env */
    int32_t reserved;
} igx_action_stage;

int igx_action_stage_step(const igx_action_stage *s, int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * The part of VecTask.step every task shares (tasks/base/vec_task.py:401-418).
 * ---------------------------------------------------------------------- */
typedef struct igx_vec_task_tail {
    const float *obs_buf;      /* [N,num_obs] */
    float *obs_out;            /* clamp(obs_buf, -clip_obs, clip_obs) (:412); must not alias obs_buf */
    int64_t obs_count;         /* N * num_obs floats */
    const float *states_buf;   /* [N,num_states] or NULL */
    float *states_out;         /* get_state() (:355, :415-416) */
    int64_t states_count;      /* N * num_states floats, 0 without asymmetric observations */
    const int64_t *progress;   /* [N] */
    const int64_t *reset;      /* [N] */
    uint8_t *time_outs;        /* [N] bool: (progress >= max_episode_length - 1) & (reset != 0) (:404); NULL skips */
    double max_episode_length_m1;
    float clip_obs;            /* env.clipObservations; +inf copies */
    int32_t reserved;
    float *obs_noised;         /* [N,num_obs] obs_buf after observation noise (:407-408), may be obs_buf
                                  itself (the reference rebinds self.obs_buf to it); NULL keeps no copy */
    igx_noise obs_noise;       /* applied before the clamp; corr == NULL: none */
} igx_vec_task_tail;

int igx_vec_task_tail_step(const igx_vec_task_tail *t, int64_t n_envs, igx_stream_t stream);

/* torch.clamp(x, lo, hi) on count floats (the action clamp, tasks/base/vec_task.py:384); NaN stays NaN. */
int igx_clamp(const float *x, float lo, float hi, float *out, int64_t count, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * Trajectory recorder: the fork's sampling mode (tasks/cartpole_sample.py:99-107,
 * :119-121, :289-293) captures selected observation columns of every
 * environment at every step with obs.clone().cpu() and an O(T^2) torch.cat.
 * Here a capture is a gather kernel on the producer's stream into a device
 * staging slot, and a copy into pinned host memory on the recorder's own side
 * stream, ordered by events.  The copy is a kernel of a few workgroups writing
 * through the host mapping: the runtime's device-to-host copy is a blit kernel
 * that fills the GPU for the whole transfer and holds the producer's kernels up.  A slot batches consecutive steps up to about
 * 128 MiB (at most 64 steps) per copy; the producer waits only when the ring of slots is full, i.e. when
 * it produces faster than the host link drains.
 * Host layout: [max_steps][num_cols][n_rows] fp32 (step-major), pinned.
 * With cols == NULL (and num_cols == 0) a capture takes the whole buffer as it is - host layout
 * [max_steps][n_rows][row_width]: the batches of HumanoidAMP.fetch_amp_obs_demo (tasks/humanoid_amp.py:146-171,
 * [num_samples, numAMPObsSteps * 105]) streamed to the host as they are sampled.
 * ---------------------------------------------------------------------- */
typedef struct igx_recorder igx_recorder;

/* Allocates the staging ring (ring_slots slots of 1..64 steps of num_cols x n_rows floats on the
 * current device) and the pinned host buffer.  cols: column indices into a row of row_width (at most 1024),
 * or NULL with num_cols == 0 for whole rows. */
int igx_recorder_create(igx_recorder **out, int64_t n_rows, int32_t row_width, const int32_t *cols,
                        int32_t num_cols, int32_t ring_slots, int64_t max_steps);
/* Capture src [n_rows,row_width] (device) as the next step; asynchronous.
 * IGX_E_SIZE once max_steps captures have been taken. */
/* (capture and sync order the copy only against the stream of the LAST capture: a producer that changes
 * streams must order them itself; the recorder's device must be the current device of every call) */
int igx_recorder_capture(igx_recorder *r, const float *src, igx_stream_t producer_stream);
/* Block the host until every capture so far has landed in host memory (a partly filled batch is
 * copied now, ordered behind the stream of the last capture). */
int igx_recorder_sync(igx_recorder *r);
int64_t igx_recorder_steps(const igx_recorder *r);
const float *igx_recorder_host_data(const igx_recorder *r);
int igx_recorder_destroy(igx_recorder *r);

/* ------------------------------------------------------------------------
 * Episode statistics (learning/common_agent.py:284-296 game_rewards /
 * game_lengths; utils/rlgames_utils.py:151-174): per step
 *   current_rewards += rew; current_lengths += 1;
 *   done envs contribute (1, current_rewards, current_lengths) to the totals;
 *   current_rewards *= 1 - done; current_lengths *= 1 - done.
 * workspace: IGX_EPISODE_WORKSPACE_BYTES(n_envs), 16-byte aligned; its first
 * record {uint64 n_done; double sum_reward; double sum_length; uint64 steps}
 * is the running total since the caller last zeroed it (zero it before the first
 * call), the second the totals of the last call.  Shards of a multi-GPU job
 * all-gather their first records and add them.
 * ---------------------------------------------------------------------- */
#define IGX_EPISODE_RECORD_BYTES 32
#define IGX_EPISODE_WORKSPACE_BYTES(n_envs) (32 * (((int64_t)(n_envs) + 255) / 256 + 2))
typedef struct igx_episode_buffers {
    const float *rew;          /* [N] */
    const int64_t *done;       /* [N] the reset flags returned by step() */
    float *current_rewards;    /* [N] in/out */
    float *current_lengths;    /* [N] in/out */
    void *workspace;
} igx_episode_buffers;

int igx_episode_stats_step(const igx_episode_buffers *b, int64_t n_envs, igx_stream_t stream);

/* The episode_cumulative part of RLGPUAlgoObserver.process_infos (utils/rlgames_utils.py:159-174):
 * for every key of infos['episode_cumulative']
 *     cumulative[key] += value
 *     for done_idx in done_indices (ascending):  window[key].append(cumulative[key][done_idx]);
 *                                                cumulative[key][done_idx] = 0
 * where window[key] is the reference's deque(maxlen=games_to_track): here a ring of games_to_track
 * floats per key; finished episode number i (counted from the last time the caller zeroed the
 * workspace) sits at ring position i % games_to_track, so the valid entries, oldest first, are
 * positions (total - min(total, G)) .. (total - 1), each modulo G.  No nonzero(), no .item():
 * after_print_stats (:200-206) reads the ring and the header once per epoch.
 * workspace: IGX_EPISODE_WINDOW_WORKSPACE_BYTES(n_envs), 16-byte aligned, zeroed once by the caller;
 * it starts with {uint64 finished_total; uint64 base; uint64 last_count; uint64 pad}. */
#define IGX_EPISODE_WINDOW_MAX_KEYS 16
#define IGX_EPISODE_WINDOW_HEADER_BYTES 32
#define IGX_EPISODE_WINDOW_WORKSPACE_BYTES(n_envs) (32 + 16 * ((((int64_t)(n_envs) + 1023) / 1024 + 3) / 4))
typedef struct igx_episode_window {
    const int64_t *done;        /* [N] the reset flags returned by step() */
    int32_t num_keys;
    int32_t reserved;
    int64_t games_to_track;     /* G: ring length per key */
    const float *values[IGX_EPISODE_WINDOW_MAX_KEYS];   /* [N] per key: this step's infos['episode_cumulative'][key] */
    float *cumulative[IGX_EPISODE_WINDOW_MAX_KEYS];     /* [N] per key, in/out */
    float *window;              /* [num_keys, G] rings */
    void *workspace;
} igx_episode_window;

int igx_episode_window_step(const igx_episode_window *w, int64_t n_envs, igx_stream_t stream);

/* ------------------------------------------------------------------------
 * torch_jit_utils math, one batched call per reference function
 * (utils/torch_jit_utils.py).  Quaternions are [n,4] xyzw, vectors [n,3],
 * all fp32 contiguous; `out` may alias an input of the same shape.
 * ---------------------------------------------------------------------- */
/* :41-62 quat_mul(a, b) */
int igx_quat_mul(const float *a, const float *b, float *out, int64_t n, igx_stream_t stream);
/* :106-110 quat_conjugate(a) */
int igx_quat_conjugate(const float *a, float *out, int64_t n, igx_stream_t stream);
/* :80-90 quat_rotate(q, v) (also :411-421 my_quat_rotate) */
int igx_quat_rotate(const float *q, const float *v, float *out, int64_t n, igx_stream_t stream);
/* :93-103 quat_rotate_inverse(q, v) */
int igx_quat_rotate_inverse(const float *q, const float *v, float *out, int64_t n, igx_stream_t stream);
/* :595-628 slerp(q0, q1, t); t is [n] (one blend per quaternion, may be outside [0,1]) */
int igx_slerp(const float *q0, const float *q1, const float *t, float *out, int64_t n, igx_stream_t stream);
/* :234-236 scale, :239-241 unscale, :228-230 tensor_clamp: x is [n,d]; lower / upper are [d] */
int igx_scale(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
              igx_stream_t stream);
int igx_unscale(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
                igx_stream_t stream);
int igx_tensor_clamp(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
                     igx_stream_t stream);
/* :118-123 quat_from_angle_axis(angle [n], axis [n,3]) */
int igx_quat_from_angle_axis(const float *angle, const float *axis, float *out, int64_t n, igx_stream_t stream);
/* :126-128 normalize_angle(x [n]) */
int igx_normalize_angle(const float *x, float *out, int64_t n, igx_stream_t stream);
/* :454-461 quat_to_exp_map(q) -> [n,3] */
int igx_quat_to_exp_map(const float *q, float *out, int64_t n, igx_stream_t stream);
/* :589-593 exp_map_to_quat(e [n,3]) -> [n,4] */
int igx_exp_map_to_quat(const float *e, float *out, int64_t n, igx_stream_t stream);
/* :548-561 quat_to_tan_norm(q) -> [n,6] */
int igx_quat_to_tan_norm(const float *q, float *out, int64_t n, igx_stream_t stream);
/* :656-667 calc_heading_quat_inv(q) -> [n,4] */
int igx_calc_heading_quat_inv(const float *q, float *out, int64_t n, igx_stream_t stream);
/* :65-67 normalize(x [n,d]): x / clamp(||x||, 1e-9); d = 3 or 4 */
int igx_normalize(const float *x, float *out, int64_t n, int32_t d, igx_stream_t stream);
/* :113-115 quat_unit(a) = normalize of [n,4] */
int igx_quat_unit(const float *a, float *out, int64_t n, igx_stream_t stream);
/* :70-77 quat_apply(a [n,4], b [n,3]) -> [n,3]; tf_vector (:142-144) is the same function */
int igx_quat_apply(const float *a, const float *b, float *out, int64_t n, igx_stream_t stream);
/* :131-134 tf_inverse(q, t) -> (quat_conjugate(q) [n,4], -quat_apply(q_inv, t) [n,3]) */
int igx_tf_inverse(const float *q, const float *t, float *q_out, float *t_out, int64_t n, igx_stream_t stream);
/* :137-139 tf_apply(q, t, v) = quat_apply(q, v) + t */
int igx_tf_apply(const float *q, const float *t, const float *v, float *out, int64_t n, igx_stream_t stream);
/* :147-149 tf_combine(q1, t1, q2, t2) -> (quat_mul(q1, q2), quat_apply(q1, t2) + t1) */
int igx_tf_combine(const float *q1, const float *t1, const float *q2, const float *t2, float *q_out, float *t_out,
                   int64_t n, igx_stream_t stream);
/* :168-172 copysign(a, b [n]) = |a| * sign(b)  (0 where b is +-0 or NaN, as torch.sign) */
int igx_copysign(float a, const float *b, float *out, int64_t n, igx_stream_t stream);
/* :175-195 get_euler_xyz(q) -> roll, pitch, yaw [n] each, in [0, 2 pi) */
int igx_get_euler_xyz(const float *q, float *roll, float *pitch, float *yaw, int64_t n, igx_stream_t stream);
/* :198-212 quat_from_euler_xyz(roll, pitch, yaw [n]) -> [n,4] */
int igx_quat_from_euler_xyz(const float *roll, const float *pitch, const float *yaw, float *out, int64_t n,
                            igx_stream_t stream);
/* :563-568 euler_xyz_to_exp_map(roll, pitch, yaw [n]) -> [n,3] */
int igx_euler_xyz_to_exp_map(const float *roll, const float *pitch, const float *yaw, float *out, int64_t n,
                             igx_stream_t stream);
/* :215-218 torch_rand_float(lower, upper, shape): (upper - lower) * u + lower over `count` elements.
 * uniforms: the torch.rand draws in [0,1) to reproduce a torch sequence, or NULL for the library's
 * Philox4x32-10 stream keyed by (seed, rng_offset, element). */
int igx_torch_rand_float(double lower, double upper, const float *uniforms, uint64_t seed, uint64_t rng_offset,
                         float *out, int64_t count, igx_stream_t stream);
/* :248-263 compute_heading_and_up(torso_rotation [n,4], inv_start_rot [n,4], to_target [n,3], vec0 [n,3],
 * vec1 [n,3], up_idx) -> torso_quat [n,4], up_proj [n], heading_proj [n], up_vec [n,3], heading_vec [n,3] */
int igx_compute_heading_and_up(const float *torso_rotation, const float *inv_start_rot, const float *to_target,
                               const float *vec0, const float *vec1, int32_t up_idx, float *torso_quat,
                               float *up_proj, float *heading_proj, float *up_vec, float *heading_vec, int64_t n,
                               igx_stream_t stream);
/* :266-277 compute_rot(torso_quat [n,4], velocity, ang_velocity, targets, torso_positions [n,3] each)
 * -> vel_loc [n,3], angvel_loc [n,3], roll, pitch, yaw, angle_to_target [n] each */
int igx_compute_rot(const float *torso_quat, const float *velocity, const float *ang_velocity, const float *targets,
                    const float *torso_positions, float *vel_loc, float *angvel_loc, float *roll, float *pitch,
                    float *yaw, float *angle_to_target, int64_t n, igx_stream_t stream);
/* :280-285 quat_axis(q, axis in 0..2) = quat_rotate(q, unit vector of that axis) */
int igx_quat_axis(const float *q, int32_t axis, float *out, int64_t n, igx_stream_t stream);
/* :292-310 scale_transform, :313-331 unscale_transform, :333-348 saturate: x [n,d] against lower / upper [d] */
int igx_scale_transform(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
                        igx_stream_t stream);
int igx_unscale_transform(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
                          igx_stream_t stream);
int igx_saturate(const float *x, const float *lower, const float *upper, float *out, int64_t n, int32_t d,
                 igx_stream_t stream);
/* :354-373 quat_diff_rad(a, b) -> [n] */
int igx_quat_diff_rad(const float *a, const float *b, float *out, int64_t n, igx_stream_t stream);
/* :376-395 local_to_world_space(pos_offset_local [n,3], pose_global [n,7] = pos 3 + quat 4) -> [n,3] */
int igx_local_to_world_space(const float *pos_offset_local, const float *pose_global, float *out, int64_t n,
                             igx_stream_t stream);
/* :423-444 quat_to_angle_axis(q) -> angle [n], axis [n,3] */
int igx_quat_to_angle_axis(const float *q, float *angle, float *axis, int64_t n, igx_stream_t stream);
/* :446-452 angle_axis_to_exp_map(angle [n], axis [n,3]) -> [n,3] */
int igx_angle_axis_to_exp_map(const float *angle, const float *axis, float *out, int64_t n, igx_stream_t stream);
/* :570-587 exp_map_to_angle_axis(exp_map [n,3]) -> angle [n], axis [n,3] */
int igx_exp_map_to_angle_axis(const float *exp_map, float *angle, float *axis, int64_t n, igx_stream_t stream);
/* :630-641 calc_heading(q) -> [n]; :643-654 calc_heading_quat(q) -> [n,4] */
int igx_calc_heading(const float *q, float *out, int64_t n, igx_stream_t stream);
int igx_calc_heading_quat(const float *q, float *out, int64_t n, igx_stream_t stream);
/* :670-681 extract_positions_from_rigid_body_states(tensor [n, >= 13 * num_bodies]) -> [n, 3 * num_bodies]: columns
 * 13 i + {0, 1, 2} of every row, i < num_bodies (the reference fixes num_bodies = 22, HumanoidImitate's body count:
 * tasks/humanoid_imitate.py:708, :811).  row_stride = floats between consecutive rows of `states`. */
int igx_extract_positions_from_rigid_body_states(const float *states, int64_t row_stride, int32_t num_bodies, float *out,
                                                 int64_t n, igx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IGX_H */
