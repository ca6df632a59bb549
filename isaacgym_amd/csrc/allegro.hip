// AllegroHand in-hand reorientation: pre_physics_step (goal / env resets and
// the action -> joint-target path) and post_physics_step (observations,
// compute_hand_reward, cross-environment consecutive_successes update).
//
// Both kernels give a wavefront a tile of consecutive environments.
// post: the tile's dof_state / dof_force / actions slices stream in with
//   16-byte non-temporal loads and land in their final columns of the LDS image
//   of the observation rows; lane i takes its object's root row (addressed by
//   object_indices) and its goal pose, writes the 24 (18) pose columns, and
//   evaluates the reward from its own row; rows leave as whole lines.  The two
//   global sums of compute_hand_reward are reduced per wave into a record per
//   tile and added in a fixed order by a one-block kernel, which also folds them
//   into consecutive_successes.
// pre: flagged environments are re-initialised one at a time by the whole wave
//   (lane j = joint j), then the action path runs over the tile's [64,16]
//   slab of targets with coalesced 16-byte accesses.
//
// Reference: IsaacGymEnvs/isaacgymenvs/tasks/allegro_hand.py:586-615, :506-584,
// :722-732 (pre) and :627-632, :406-504, :663-719 (post).
#include <atomic>

#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_hand.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int AD = IGX_ALLEGRO_DOF;

template <int OBS>
struct AllegroLayout {
    static constexpr bool VEL = OBS != IGX_ALLEGRO_OBS_FULL_NO_VEL;
    static constexpr bool FORCE = OBS == IGX_ALLEGRO_OBS_FULL_STATE;
    static constexpr int C_POS = 0;
    static constexpr int C_VEL = AD;                                  // if VEL
    static constexpr int C_FORCE = 2 * AD;                            // if FORCE
    static constexpr int C_POSE = AD * (1 + (VEL ? 1 : 0) + (FORCE ? 1 : 0));
    static constexpr int POSE_W = VEL ? 24 : 18;                      // obj pose 7 (+ vel 6) + goal pose 7 + quat diff 4
    static constexpr int C_ACT = C_POSE + POSE_W;
    static constexpr int NOBS = C_ACT + AD;                           // 50 / 72 / 88
};

struct f2pair { float up, uv; };

// j-th of this environment's 45 uniforms, mapped to torch_rand_float(-1, 1)
__device__ __forceinline__ float allegro_draw(const igx_allegro_params &p, const igx_allegro_buffers &b, int64_t e, int j)
{
    float u;
    if (b.rand) {
        u = b.rand[e * IGX_ALLEGRO_RAND_PER_ENV + j];
    } else {
        float q[4];
        philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, j >> 2, q);
        const int a = j & 3;
        u = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
    }
    return 2.0f * u + (-1.0f);
}

__global__ __launch_bounds__(WAVE) void allegro_pre_kernel(const igx_allegro_params p, const igx_allegro_buffers b,
                                                           int64_t n)
{
    __shared__ __align__(16) float s_prev[WAVE * AD];
    __shared__ float s_lim[2 * AD];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nvalid = (n - base >= WAVE) ? WAVE : (int)(n - base);
    const bool live = lane < nvalid;
    const int64_t e = base + lane;
    if (lane < 2 * AD) s_lim[lane] = lane < AD ? p.dof_lower[lane] : p.dof_upper[lane - AD];

    int64_t rst = 0, rg = 0;
    if (live) {
        rst = b.reset[e];
        rg = b.reset_goal[e];
    }
    // the tile's slab of previous targets -> LDS (the reset below may overwrite rows)
    const int count = nvalid * AD;
    v4f act[AD / 4];  // the actions are requested in the same round trip
    if (nvalid == WAVE) {
        v4f r[AD / 4];
        load_chunks<WAVE * AD / 4>(b.prev_targets + base * AD, r, lane);
        load_chunks<WAVE * AD / 4>(b.actions + base * AD, act, lane);
        if (b.actions_out) {  // VecTask.step's clamp, folded in: self.actions receives the clamped actions
            const float hi = p.clip_actions, lo = -p.clip_actions;
#pragma unroll
            for (int it = 0; it < AD / 4; ++it) {
                act[it].x = clamp1(act[it].x, lo, hi); act[it].y = clamp1(act[it].y, lo, hi);
                act[it].z = clamp1(act[it].z, lo, hi); act[it].w = clamp1(act[it].w, lo, hi);
                st_stream(reinterpret_cast<v4f *>(b.actions_out + base * AD) + lane + it * WAVE, act[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < AD / 4; ++it) *reinterpret_cast<v4f *>(s_prev + 4 * (lane + it * WAVE)) = r[it];
    } else {
        for (int i = lane; i < count; i += WAVE) s_prev[i] = b.prev_targets[base * AD + i];
    }
    wave_sync();

    // ---- resets: goal-only resets first, then full environment resets, in the order the
    // reference applies them.  The per-environment part (goal pose, object actor) is done by
    // the environment's own lane, so a tile in which every environment resets (the first step:
    // reset_buf starts at ones) costs the same as a tile with one reset.
    if (live && rg != 0) reset_target_pose(p, b, e, allegro_draw(p, b, e, 0), allegro_draw(p, b, e, 1));
    if (live && rst != 0) {  // target pose again (inside reset_idx), then the object actor
        reset_target_pose(p, b, e, allegro_draw(p, b, e, 41), allegro_draw(p, b, e, 42));
        reset_object_row(p, b, e, allegro_draw(p, b, e, 4), allegro_draw(p, b, e, 5), allegro_draw(p, b, e, 6),
                         allegro_draw(p, b, e, 7), allegro_draw(p, b, e, 8));
    }
    // hand joints: four flagged environments per round, lane = (slot, joint)
    uint64_t env_mask = __ballot(live && rst != 0);
    while (env_mask) {
        int srcs[WAVE / AD];
#pragma unroll
        for (int s = 0; s < WAVE / AD; ++s) {
            srcs[s] = env_mask ? __builtin_ctzll(env_mask) : -1;
            env_mask &= env_mask - 1;  // 0 stays 0
        }
        const int slot = lane / AD;
        int src = srcs[0];
#pragma unroll
        for (int s = 1; s < WAVE / AD; ++s) src = slot == s ? srcs[s] : src;
        if (src >= 0) {
            const int64_t ee = base + src;
            const int j = lane % AD;
            const float rp = allegro_draw(p, b, ee, 4 + 5 + j);
            const float rv = allegro_draw(p, b, ee, 4 + 5 + AD + j);
            const float delta_max = s_lim[AD + j] - 0.0f, delta_min = s_lim[j] - 0.0f;
            const float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rp + 1.0f);
            const float pos = 0.0f + p.reset_dof_pos_noise * rand_delta;
            const float vel = 0.0f + p.reset_dof_vel_noise * rv;
            reinterpret_cast<v2f *>(b.dof_state + ee * (2 * AD))[j] = v2f{pos, vel};
            s_prev[src * AD + j] = pos;  // prev_targets = cur_targets = pos, consumed by the action path below
        }
    }
    wave_sync();

    // ---- action path over the tile's slab (chunk k: env k/4, joints 4(k%4)..)
    auto target = [&](float a, float prev, float lo, float hi) { return joint_target(p, a, prev, lo, hi); };
    if (nvalid == WAVE) {
#pragma unroll
        for (int it = 0; it < AD / 4; ++it) {
            const int k = lane + it * WAVE;
            const int j0 = 4 * (k & 3);
            const v4f pv = *reinterpret_cast<const v4f *>(s_prev + 4 * k);
            v4f c;
            c.x = target(act[it].x, pv.x, s_lim[j0 + 0], s_lim[AD + j0 + 0]);
            c.y = target(act[it].y, pv.y, s_lim[j0 + 1], s_lim[AD + j0 + 1]);
            c.z = target(act[it].z, pv.z, s_lim[j0 + 2], s_lim[AD + j0 + 2]);
            c.w = target(act[it].w, pv.w, s_lim[j0 + 3], s_lim[AD + j0 + 3]);
            st_stream(reinterpret_cast<v4f *>(b.cur_targets + base * AD) + k, c);
            st_stream(reinterpret_cast<v4f *>(b.prev_targets + base * AD) + k, c);
        }
    } else {
        for (int i = lane; i < count; i += WAVE) {
            const int j = i & (AD - 1);
            float a = b.actions[base * AD + i];
            if (b.actions_out) {
                a = clamp1(a, -p.clip_actions, p.clip_actions);
                b.actions_out[base * AD + i] = a;
            }
            const float c = target(a, s_prev[i], s_lim[j], s_lim[AD + j]);
            b.cur_targets[base * AD + i] = c;
            b.prev_targets[base * AD + i] = c;
        }
    }
}

// T environments per tile (one wavefront each)
template <int OBS, bool REWARD, int T>
__global__ __launch_bounds__(WAVE) void allegro_post_kernel(const igx_allegro_params p, const igx_allegro_buffers b,
                                                            float *__restrict__ obs_out, int64_t n)
{
    using L = AllegroLayout<OBS>;
    constexpr int NOBS = L::NOBS;
    static_assert(T == 32 || T == 64, "tile rows");
    __shared__ __align__(16) float s_tile[T * NOBS];
    __shared__ __align__(16) float s_lim[2 * AD];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * T;
    const int nvalid = (n - base >= T) ? T : (int)(n - base);
    const bool full = nvalid == T;
    const bool live = lane < nvalid;
    const int64_t e = base + lane;
    if (lane < 2 * AD) s_lim[lane] = lane < AD ? p.dof_lower[lane] : p.dof_upper[lane - AD];

    // ---- loads
    HandFlags fl{0, 0, 0, 0.0f};
    int64_t rnd = 0;
    float obj[13], goal[7];
    v4f r_dof[T * AD * 2 / 4 / WAVE], r_force[T * AD / 4 / WAVE], r_act[T * AD / 4 / WAVE];
    v4f r_goal[(T * 13 / 4 + WAVE - 1) / WAVE];
    if (live && REWARD) {
        fl.prog = ld_narrow(b.progress + e);
        fl.rst = ld_narrow(b.reset + e);
        fl.rg = ld_narrow(b.reset_goal + e);
        fl.succ = ld_narrow(b.successes + e);
        if (b.randomize) rnd = ld_narrow(b.randomize + e);
    }
    RowWindow<T, T * NOBS> object_rows;
    object_rows.load(b.root_state_tensor, b.object_indices, e, live, full, lane, obj);
    if (full) {
        load_chunks<T * AD * 2 / 4>(b.dof_state + base * (2 * AD), r_dof, lane);
        if (L::FORCE) load_chunks<T * AD / 4>(b.dof_force + base * AD, r_force, lane);
        load_chunks<T * AD / 4>(b.actions + base * AD, r_act, lane);
        load_chunks<T * 13 / 4>(b.goal_states + base * 13, r_goal, lane);
    }

    // ---- goal rows, then the object rows of a windowed tile: transposed through the (still empty) tile
    park_rows<13, T>(r_goal, b.goal_states + base * 13, s_tile, nvalid, lane);
    wave_sync();
    if (live) {
#pragma unroll
        for (int i = 0; i < 7; ++i) goal[i] = s_tile[lane * 13 + i];
    }
    wave_sync();
    object_rows.pick(s_tile, live, lane, obj);

    // ---- stage pass-through columns
    if (full) {
        if constexpr (L::VEL) {
            stage_dof_state<AD, NOBS, L::C_POS, L::C_VEL, T>(r_dof, s_tile, s_lim, p.vel_obs_scale, lane);
        } else {  // positions only
#pragma unroll
            for (int it = 0; it < T * AD * 2 / 4 / WAVE; ++it) {
                const int pr = 2 * (lane + it * WAVE);
                const int env = pr / AD, j = pr - env * AD;
                s_tile[env * NOBS + L::C_POS + j] = unscale(r_dof[it].x, s_lim[j], s_lim[AD + j]);
                s_tile[env * NOBS + L::C_POS + j + 1] = unscale(r_dof[it].z, s_lim[j + 1], s_lim[AD + j + 1]);
            }
        }
        if constexpr (L::FORCE) stage_rows<AD, L::C_FORCE, NOBS, true, T>(r_force, s_tile, p.force_torque_obs_scale, lane);
        stage_rows<AD, L::C_ACT, NOBS, false, T>(r_act, s_tile, 1.0f, lane);
    } else {
        for (int i = lane; i < nvalid * AD; i += WAVE) {
            const int env = i / AD, j = i - env * AD;
            const v2f pv = reinterpret_cast<const v2f *>(b.dof_state)[base * AD + i];
            s_tile[env * NOBS + L::C_POS + j] = unscale(pv.x, s_lim[j], s_lim[AD + j]);
            if (L::VEL) s_tile[env * NOBS + L::C_VEL + j] = p.vel_obs_scale * pv.y;
            if (L::FORCE) s_tile[env * NOBS + L::C_FORCE + j] = p.force_torque_obs_scale * b.dof_force[base * AD + i];
            s_tile[env * NOBS + L::C_ACT + j] = b.actions[base * AD + i];
        }
    }
    wave_sync();

    // ---- per-env: pose columns and reward
    float fin = 0.0f;
    fl.rst = live ? fl.rst : 0;
    if (live) {
        float *row = s_tile + lane * NOBS;
        const f4 orot{obj[3], obj[4], obj[5], obj[6]};
        const f4 grot{goal[3], goal[4], goal[5], goal[6]};
        const f4 qd = quat_mul(orot, quat_conjugate(grot));
        float *c = row + L::C_POSE;
#pragma unroll
        for (int i = 0; i < 7; ++i) c[i] = obj[i];
        int k = 7;
        if (L::VEL) {
            c[7] = obj[7]; c[8] = obj[8]; c[9] = obj[9];
            c[10] = p.vel_obs_scale * obj[10]; c[11] = p.vel_obs_scale * obj[11]; c[12] = p.vel_obs_scale * obj[12];
            k = 13;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) c[k + i] = goal[i];
        c[k + 7] = qd.x; c[k + 8] = qd.y; c[k + 9] = qd.z; c[k + 10] = qd.w;

        if (REWARD) {  // :663-719 compute_hand_reward
            if (b.randomize) st_narrow(b.randomize + e, rnd + 1);
            float act[AD], sq[AD];
            read_cols<AD, L::C_ACT, NOBS>(row, act);
#pragma unroll
            for (int j = 0; j < AD; ++j) sq[j] = act[j] * act[j];
            const float reward = hand_reward(p, obj, goal, qd, row_sum<AD>(sq), fl, fin);
            st_narrow(b.rew + e, reward);
            st_narrow(b.reset + e, fl.rst);
            st_narrow(b.reset_goal + e, fl.rg);
            st_narrow(b.progress + e, fl.prog);
            st_narrow(b.successes + e, fl.succ);
            if (b.time_outs) b.time_outs[e] = ((float)fl.prog >= p.max_episode_length_m1 && fl.rst != 0) ? 1 : 0;
        }
    }
    wave_sync();
    drain_rows<NOBS, T>(obs_out + base * NOBS, s_tile, nvalid, lane);
    // the clamped copies VecTask.step hands to the learner (this launch writes obs, or only states)
    float *clamped = REWARD ? b.obs_clamped : b.states_clamped;
    if (clamped) drain_rows_clamped<NOBS, T>(clamped + base * NOBS, s_tile, nvalid, lane, p.clip_obs);
    if (REWARD && OBS == IGX_ALLEGRO_OBS_FULL_STATE && b.states) {  // asymmetric observations: same 88 columns
        drain_rows<NOBS, T>(b.states + base * NOBS, s_tile, nvalid, lane);
        if (b.states_clamped) drain_rows_clamped<NOBS, T>(b.states_clamped + base * NOBS, s_tile, nvalid, lane, p.clip_obs);
    }
    if (REWARD) store_tile_record(b.workspace, fin, live ? fl.rst : 0, lane);
}

static int check_allegro(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n, bool post,
                         const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n);
    if (p->obs_type < 0 || p->obs_type > 2) return fail(IGX_E_PARAM, "%s: unknown obs_type %d", who, p->obs_type);
    if (n == 0) return IGX_OK;
#define REQ(field, vec)                                                                   \
    do {                                                                                  \
        if (!b->field) return fail(IGX_E_NULL, "%s: %s is NULL", who, #field);            \
        if (vec && !aligned16(b->field))                                                  \
            return fail(IGX_E_ALIGN, "%s: %s must be 16-byte aligned", who, #field);      \
    } while (0)
    REQ(dof_state, true);
    REQ(root_state_tensor, false);
    REQ(object_indices, false);
    REQ(goal_states, true);
    REQ(actions, true);
    REQ(reset, false);
    REQ(reset_goal, false);
    REQ(progress, false);
    REQ(successes, false);
    if (post) {
        REQ(obs, true);
        REQ(rew, false);
        REQ(consecutive_successes, false);
        REQ(workspace, true);
        if (p->obs_type == IGX_ALLEGRO_OBS_FULL_STATE || b->states) REQ(dof_force, true);
        if (b->states && !aligned16(b->states)) return fail(IGX_E_ALIGN, "%s: states must be 16-byte aligned", who);
        if (b->obs_clamped || b->states_clamped) {
            if (!(p->clip_obs >= 0.0f)) return fail(IGX_E_PARAM, "%s: clip_obs must be >= 0 when a clamped copy is asked for", who);
            if (b->states_clamped && !b->states) return fail(IGX_E_NULL, "%s: states_clamped needs states", who);
            if (b->obs_clamped == b->obs || (b->states_clamped && b->states_clamped == b->states))
                return fail(IGX_E_PARAM, "%s: a clamped copy must not alias its source", who);
        }
    } else {
        REQ(goal_object_indices, false);
        REQ(goal_init_state, false);
        REQ(object_init_state, false);
        REQ(prev_targets, true);
        REQ(cur_targets, true);
        if (b->actions_out) {
            if (!aligned16(b->actions_out)) return fail(IGX_E_ALIGN, "%s: actions_out must be 16-byte aligned", who);
            if (!(p->clip_actions >= 0.0f)) return fail(IGX_E_PARAM, "%s: clip_actions must be >= 0 with actions_out", who);
        }
    }
#undef REQ
    return IGX_OK;
}

}  // namespace igx

using namespace igx;

extern "C" int igx_allegro_pre_physics(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n,
                                       igx_stream_t stream)
{
    int rc = check_allegro(p, b, n, false, "igx_allegro_pre_physics");
    if (rc != IGX_OK || n == 0) return rc;
    const int64_t tiles = (n + WAVE - 1) / WAVE;
    hipLaunchKernelGGL(allegro_pre_kernel, dim3((unsigned)tiles), dim3(WAVE), 0, static_cast<hipStream_t>(stream), *p,
                       *b, n);
    return launch_status("igx_allegro_pre_physics");
}

template <int OBS, bool REWARD, int T>
static void launch_post(const igx_allegro_params *p, const igx_allegro_buffers *b, float *obs_out, int64_t n, int reserve,
                        hipStream_t s)
{
    const unsigned tiles = (unsigned)((n + T - 1) / T);
    hipLaunchKernelGGL((allegro_post_kernel<OBS, REWARD, T>), dim3(tiles), dim3(WAVE), (size_t)reserve, s, *p, *b, obs_out, n);
}

template <int T>
static int allegro_post_launch_t(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n, int reserve,
                                 bool finalize, hipStream_t s)
{
    switch (p->obs_type) {
    case IGX_ALLEGRO_OBS_FULL_NO_VEL: launch_post<IGX_ALLEGRO_OBS_FULL_NO_VEL, true, T>(p, b, b->obs, n, reserve, s); break;
    case IGX_ALLEGRO_OBS_FULL: launch_post<IGX_ALLEGRO_OBS_FULL, true, T>(p, b, b->obs, n, reserve, s); break;
    default: launch_post<IGX_ALLEGRO_OBS_FULL_STATE, true, T>(p, b, b->obs, n, reserve, s); break;
    }
    int rc = launch_status("igx_allegro_post_physics");
    if (rc != IGX_OK) return rc;
    if (b->states && p->obs_type != IGX_ALLEGRO_OBS_FULL_STATE) {  // asymmetric critic state beside a smaller policy obs
        launch_post<IGX_ALLEGRO_OBS_FULL_STATE, false, T>(p, b, b->states, n, reserve, s);
        rc = launch_status("igx_allegro_post_physics(states)");
        if (rc != IGX_OK) return rc;
    }
    const int64_t tiles = (n + T - 1) / T;
    if (finalize)
        hipLaunchKernelGGL((hand_reduce_kernel<igx_allegro_params, igx_allegro_buffers, true>), dim3(1), dim3(1024), 0, s, *p,
                           *b, tiles);
    else
        hipLaunchKernelGGL((hand_reduce_kernel<igx_allegro_params, igx_allegro_buffers, false>), dim3(1), dim3(1024), 0, s, *p,
                           *b, tiles);
    return launch_status("igx_allegro_post_physics(reduce)");
}

// observation / reward launch; leaves the {num_resets, sum(successes * resets)} sums of these
// n environments in record 0 of b->workspace and, with `finalize`, applies them
static int allegro_post_launch(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n, bool finalize,
                               igx_stream_t stream)
{
    static const int reserve = tuning_int("IGX_ALLEGRO_LDS_RESERVE", 6000);  // unused LDS: 5 waves per CU stream fastest
    static const int tile = tuning_int("IGX_ALLEGRO_TILE", 64);           // environments per wavefront
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (tile == 32) return allegro_post_launch_t<32>(p, b, n, reserve, finalize, s);
    return allegro_post_launch_t<64>(p, b, n, reserve, finalize, s);
}

extern "C" int igx_allegro_post_physics(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n,
                                        igx_stream_t stream)
{
    int rc = check_allegro(p, b, n, true, "igx_allegro_post_physics");
    if (rc != IGX_OK || n == 0) return rc;
    return allegro_post_launch(p, b, n, true, stream);
}

// ---- the variant for callers that may not allocate and bring no workspace ------------------
// The per-tile records of the reduction live in this file-scope array instead of b->workspace.
// One copy per GPU, held until the process ends; its address on a device is looked up once with
// that device current, and only that address (never the array's name) is handed to kernels.
static __device__ __align__(256) unsigned char scratch[IGX_ALLEGRO_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];

// device address of `scratch` on the current device (no allocation; thread-safe: every thread that
// races here computes the same address)
int scratch_address(void **out, const char *who)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "%s: hipGetDevice: %s", who, hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices) return fail(IGX_E_NODEVICE, "%s: device index %d not supported", who, dev);
    void *ptr = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!ptr) {
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr)
            return fail(e != hipSuccess ? (int)e : IGX_E_NODEVICE, "%s: hipGetSymbolAddress(scratch): %s", who,
                        hipGetErrorString(e));
        g_scratch_ptr[dev].store(ptr, std::memory_order_release);
    }
    *out = ptr;
    return IGX_OK;
}
}  // namespace

// set-up call: resolves the array's address on the current device.  Meant to run once per device, outside
// any stream capture (the lookup may load the code object); the entry point below then only reads the cached
// address.  Calling it again is a table lookup.
extern "C" int igx_allegro_static_prepare(void)
{
    void *ptr = nullptr;
    return scratch_address(&ptr, "igx_allegro_static_prepare");
}

extern "C" int igx_allegro_post_physics_static(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n,
                                               igx_stream_t stream)
{
    const char *who = "igx_allegro_post_physics_static";
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n > 0) {
        // a first call per device that arrives while the stream is being captured must not start looking symbols
        // up (not a stream operation, and unsafe inside a capture): it is refused with the remedy named
        int dev = -1;
        if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDevices &&
            !g_scratch_ptr[dev].load(std::memory_order_acquire)) {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(static_cast<hipStream_t>(stream), &cs) == hipSuccess && cs != hipStreamCaptureStatusNone)
                return fail(IGX_E_PARAM, "%s: first call on device %d arrived under stream capture; call "
                            "igx_allegro_static_prepare() on this device before capturing", who, dev);
        }
    }
    if (n > IGX_ALLEGRO_STATIC_MAX_ENVS)
        return fail(IGX_E_SIZE, "%s: n_envs %lld exceeds IGX_ALLEGRO_STATIC_MAX_ENVS (%lld); use igx_allegro_post_physics "
                    "with a caller-provided workspace", who, (long long)n, (long long)IGX_ALLEGRO_STATIC_MAX_ENVS);
    igx_allegro_buffers bb = *b;
    bb.workspace = nullptr;
    if (n > 0) {
        int rc = scratch_address(&bb.workspace, who);
        if (rc != IGX_OK) return rc;
    }
    int rc = check_allegro(p, &bb, n, true, who);
    if (rc != IGX_OK || n == 0) return rc;
    // every tile writes its record and the reduction writes record 0 before anything reads them:
    // nothing of the previous call's contents survives into this one, so nothing is cleared
    return allegro_post_launch(p, &bb, n, true, stream);
}

extern "C" int igx_allegro_post_physics_shard(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n,
                                              igx_stream_t stream)
{
    int rc = check_allegro(p, b, n, true, "igx_allegro_post_physics_shard");
    if (rc != IGX_OK || n == 0) return rc;
    return allegro_post_launch(p, b, n, false, stream);
}

extern "C" int igx_allegro_finalize_shards(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                           int32_t num_shards, igx_stream_t stream)
{
    if (!p || !b) return fail(IGX_E_NULL, "igx_allegro_finalize_shards: params/buffers struct is NULL");
    if (num_shards < 1 || num_shards > 65536) return fail(IGX_E_SIZE, "igx_allegro_finalize_shards: num_shards %d", num_shards);
    if (!records || !b->consecutive_successes)
        return fail(IGX_E_NULL, "igx_allegro_finalize_shards: records / consecutive_successes is NULL");
    if (!aligned16(records)) return fail(IGX_E_ALIGN, "igx_allegro_finalize_shards: records must be 16-byte aligned");
    hipLaunchKernelGGL((hand_finalize_kernel<igx_allegro_params, igx_allegro_buffers>), dim3(1), dim3(1), 0,
                       static_cast<hipStream_t>(stream), *p, *b, static_cast<const HandRecord *>(records), (int)num_shards, 1);
    return launch_status("igx_allegro_finalize_shards");
}

extern "C" int igx_allegro_finalize_shards_queued(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                           int32_t num_shards, int32_t num_steps, igx_stream_t stream)
{
    if (!p || !b) return fail(IGX_E_NULL, "igx_allegro_finalize_shards_queued: params/buffers struct is NULL");
    if (num_shards < 1 || num_shards > 65536) return fail(IGX_E_SIZE, "igx_allegro_finalize_shards_queued: num_shards %d", num_shards);
    if (num_steps < 1 || num_steps > 65536) return fail(IGX_E_SIZE, "igx_allegro_finalize_shards_queued: num_steps %d", num_steps);
    if (!records || !b->consecutive_successes)
        return fail(IGX_E_NULL, "igx_allegro_finalize_shards_queued: records / consecutive_successes is NULL");
    if (!aligned16(records)) return fail(IGX_E_ALIGN, "igx_allegro_finalize_shards_queued: records must be 16-byte aligned");
    hipLaunchKernelGGL((hand_finalize_kernel<igx_allegro_params, igx_allegro_buffers>), dim3(1), dim3(1), 0,
                       static_cast<hipStream_t>(stream), *p, *b, static_cast<const HandRecord *>(records), (int)num_shards, (int)num_steps);
    return launch_status("igx_allegro_finalize_shards_queued");
}
