// ShadowHand in-hand reorientation: pre_physics_step (goal / env resets and the
// action -> joint-target path over the 20 actuated of 24 joints) and
// post_physics_step (four observation layouts, compute_hand_reward, the
// cross-environment consecutive_successes update).
//
// post: a wavefront owns a tile of 32 consecutive environments (211-wide rows: 27 KB of
//   LDS).  The tile's dof_state / dof_force / sensor / action / goal slices stream in
//   with 16-byte non-temporal loads and land in their observation columns in LDS; the
//   five fingertip bodies of each environment are gathered from rigid_body_states by
//   (environment, fingertip) pairs spread over the lanes; object rows come through the
//   streamed window (or a gather); lanes 0..31 write the pose columns and evaluate the
//   reward; rows leave as whole lines.
// pre: 64 environments per wavefront; the [64,24] slabs of prev / cur targets are
//   staged in LDS, flagged environments are re-initialised by the whole wave (lane j =
//   joint j), the action path scatters through actuated_dof_indices in LDS, and both
//   slabs leave with coalesced 16-byte stores.
//
// Reference: IsaacGymEnvs/isaacgymenvs/tasks/shadow_hand.py:670-698, :586-668, :803-813
// (pre) and :710-715, :437-584, :747-800 (post).
#include "igx_device.hpp"
#include "igx_hand.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int SD = IGX_SHADOW_DOF, SA = IGX_SHADOW_ACTIONS, SF = IGX_SHADOW_FINGERTIPS, SS = 6 * IGX_SHADOW_FINGERTIPS;
constexpr int ST = 32;  // environments per tile in the post kernel (16 measured the same)

template <int OBS>
struct ShadowLayout {
    static constexpr bool DOF = OBS != IGX_SHADOW_OBS_OPENAI;
    static constexpr bool VEL = OBS == IGX_SHADOW_OBS_FULL || OBS == IGX_SHADOW_OBS_FULL_STATE;
    static constexpr bool FORCE = OBS == IGX_SHADOW_OBS_FULL_STATE;  // joint forces and fingertip sensors
    static constexpr bool GOAL = DOF;                                // goal pose columns
    static constexpr int OBJ_W = !DOF ? 3 : (VEL ? 13 : 7);
    static constexpr int FT_W = VEL ? 13 : 3;                        // fingertip state or position
    static constexpr int C_POS = 0, C_VEL = SD, C_FORCE = 2 * SD;
    static constexpr int C_FT_OPENAI = 0;
    static constexpr int C_OBJ = !DOF ? SF * 3 : SD * (1 + (VEL ? 1 : 0) + (FORCE ? 1 : 0));
    static constexpr int C_GOAL = C_OBJ + OBJ_W;
    static constexpr int C_QD = C_GOAL + (GOAL ? 7 : 0);
    static constexpr int C_FT = !DOF ? 0 : C_QD + 4;
    static constexpr int C_SENS = C_FT + SF * FT_W;
    static constexpr int C_ACT = !DOF ? C_QD + 4 : C_SENS + (FORCE ? SS : 0);
    static constexpr int NOBS = C_ACT + SA;
};
static_assert(ShadowLayout<IGX_SHADOW_OBS_OPENAI>::NOBS == 42 && ShadowLayout<IGX_SHADOW_OBS_OPENAI>::C_ACT == 22, "openai");
static_assert(ShadowLayout<IGX_SHADOW_OBS_FULL_NO_VEL>::NOBS == 77 && ShadowLayout<IGX_SHADOW_OBS_FULL_NO_VEL>::C_FT == 42,
              "full_no_vel");
static_assert(ShadowLayout<IGX_SHADOW_OBS_FULL>::NOBS == 157 && ShadowLayout<IGX_SHADOW_OBS_FULL>::C_FT == 72, "full");
static_assert(ShadowLayout<IGX_SHADOW_OBS_FULL_STATE>::NOBS == 211 && ShadowLayout<IGX_SHADOW_OBS_FULL_STATE>::C_SENS == 161 &&
                  ShadowLayout<IGX_SHADOW_OBS_FULL_STATE>::C_ACT == 191,
              "full_state");

// j-th of this environment's 61 uniforms, mapped to torch_rand_float(-1, 1)
__device__ __forceinline__ float shadow_draw(const igx_shadow_params &p, const igx_shadow_buffers &b, int64_t e, int j)
{
    float u;
    if (b.rand) {
        u = b.rand[e * IGX_SHADOW_RAND_PER_ENV + j];
    } else {
        float q[4];
        philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, j >> 2, q);
        const int a = j & 3;
        u = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
    }
    return 2.0f * u + (-1.0f);
}

__global__ __launch_bounds__(WAVE) void shadow_pre_kernel(const igx_shadow_params p, const igx_shadow_buffers b, int64_t n)
{
    __shared__ __align__(16) float s_prev[WAVE * SD];
    __shared__ __align__(16) float s_cur[WAVE * SD];
    __shared__ float s_lim[2 * SD];
    __shared__ int s_idx[SA];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nvalid = (n - base >= WAVE) ? WAVE : (int)(n - base);
    const bool live = lane < nvalid;
    const int64_t e = base + lane;
    if (lane < 2 * SD) s_lim[lane] = lane < SD ? p.dof_lower[lane] : p.dof_upper[lane - SD];
    if (lane < SA) s_idx[lane] = p.actuated_dof_indices[lane];

    int64_t rst = 0, rg = 0;
    if (live) {
        rst = b.reset[e];
        rg = b.reset_goal[e];
    }
    // the tile's slabs of targets -> LDS; the actions are requested in the same round trip
    const int count = nvalid * SD;
    float r_act[SA];
    if (nvalid == WAVE) {
        v4f rp[SD / 4], rc[SD / 4];
        load_chunks<WAVE * SD / 4>(b.prev_targets + base * SD, rp, lane);
        load_chunks<WAVE * SD / 4>(b.cur_targets + base * SD, rc, lane);
#pragma unroll
        for (int it = 0; it < SA; ++it) r_act[it] = ld_narrow(b.actions + base * SA + lane + it * WAVE);
        if (b.actions_out) {  // VecTask.step's clamp, folded in: self.actions receives the clamped actions
#pragma unroll
            for (int it = 0; it < SA; ++it) {
                r_act[it] = clamp1(r_act[it], -p.clip_actions, p.clip_actions);
                st_narrow(b.actions_out + base * SA + lane + it * WAVE, r_act[it]);
            }
        }
#pragma unroll
        for (int it = 0; it < SD / 4; ++it) {
            *reinterpret_cast<v4f *>(s_prev + 4 * (lane + it * WAVE)) = rp[it];
            *reinterpret_cast<v4f *>(s_cur + 4 * (lane + it * WAVE)) = rc[it];
        }
    } else {
        for (int i = lane; i < count; i += WAVE) {
            s_prev[i] = b.prev_targets[base * SD + i];
            s_cur[i] = b.cur_targets[base * SD + i];
        }
    }
    wave_sync();

    // ---- resets (rare): goal-only resets first, then full environment resets, in the order
    // the reference applies them.  Draws of an environment: 0..3 reset_target_pose from
    // pre_physics_step, 4..56 the 53 floats of reset_idx, 57..60 its inner reset_target_pose.
    // The per-environment part (goal pose, object actor) is done by the environment's own lane:
    // a tile in which every environment resets costs the same as a tile with one reset.
    if (live && rg != 0) reset_target_pose(p, b, e, shadow_draw(p, b, e, 0), shadow_draw(p, b, e, 1));
    if (live && rst != 0) {
        reset_target_pose(p, b, e, shadow_draw(p, b, e, 57), shadow_draw(p, b, e, 58));
        reset_object_row(p, b, e, shadow_draw(p, b, e, 4), shadow_draw(p, b, e, 5), shadow_draw(p, b, e, 6),
                         shadow_draw(p, b, e, 7), shadow_draw(p, b, e, 8));
    }
    // hand joints (:646-655): two flagged environments per round, lane = (slot, joint)
    uint64_t env_mask = __ballot(live && rst != 0);
    while (env_mask) {
        const int src0 = __builtin_ctzll(env_mask);
        env_mask &= env_mask - 1;
        const int src1 = env_mask ? __builtin_ctzll(env_mask) : -1;
        env_mask &= env_mask - 1;  // 0 stays 0
        const int src = lane < SD ? src0 : (lane < 2 * SD ? src1 : -1);
        if (src >= 0) {
            const int64_t ee = base + src;
            const int j = lane < SD ? lane : lane - SD;
            const float rp = shadow_draw(p, b, ee, 4 + 5 + j);
            const float rv = shadow_draw(p, b, ee, 4 + 5 + SD + j);
            const float delta_max = s_lim[SD + j] - 0.0f, delta_min = s_lim[j] - 0.0f;
            const float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rp + 1.0f);
            const float pos = 0.0f + p.reset_dof_pos_noise * rand_delta;
            const float vel = 0.0f + p.reset_dof_vel_noise * rv;
            reinterpret_cast<v2f *>(b.dof_state + ee * (2 * SD))[j] = v2f{pos, vel};
            s_prev[src * SD + j] = pos;
            s_cur[src * SD + j] = pos;
        }
    }
    wave_sync();

    // ---- action path (:684-697): actuated joints only; the others keep their targets
    if (nvalid == WAVE) {
#pragma unroll
        for (int it = 0; it < SA; ++it) {
            const int i = lane + it * WAVE;
            const int env = i / SA, a = i - env * SA, j = s_idx[a];
            const float c = joint_target(p, r_act[it], s_prev[env * SD + j], s_lim[j], s_lim[SD + j]);
            s_cur[env * SD + j] = c;
            s_prev[env * SD + j] = c;
        }
    } else {
        for (int i = lane; i < nvalid * SA; i += WAVE) {
            const int env = i / SA, a = i - env * SA, j = s_idx[a];
            float act = b.actions[base * SA + i];
            if (b.actions_out) {
                act = clamp1(act, -p.clip_actions, p.clip_actions);
                b.actions_out[base * SA + i] = act;
            }
            const float c = joint_target(p, act, s_prev[env * SD + j], s_lim[j], s_lim[SD + j]);
            s_cur[env * SD + j] = c;
            s_prev[env * SD + j] = c;
        }
    }
    wave_sync();
    if (nvalid == WAVE) {
#pragma unroll
        for (int it = 0; it < SD / 4; ++it) {
            const int k = lane + it * WAVE;
            st_stream(reinterpret_cast<v4f *>(b.cur_targets + base * SD) + k, *reinterpret_cast<const v4f *>(s_cur + 4 * k));
            st_stream(reinterpret_cast<v4f *>(b.prev_targets + base * SD) + k, *reinterpret_cast<const v4f *>(s_prev + 4 * k));
        }
    } else {
        for (int i = lane; i < count; i += WAVE) {
            b.cur_targets[base * SD + i] = s_cur[i];
            b.prev_targets[base * SD + i] = s_prev[i];
        }
    }
}

template <int OBS, bool REWARD>
__global__ __launch_bounds__(WAVE) void shadow_post_kernel(const igx_shadow_params p, const igx_shadow_buffers b,
                                                           float *__restrict__ obs_out, int64_t n)
{
    using L = ShadowLayout<OBS>;
    constexpr int NOBS = L::NOBS, T = ST;
    __shared__ __align__(16) float s_tile[T * NOBS];
    __shared__ __align__(16) float s_lim[2 * SD];
    __shared__ int s_ft[SF];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * T;
    const int nvalid = (n - base >= T) ? T : (int)(n - base);
    const bool full = nvalid == T;
    const bool live = lane < nvalid;
    const int64_t e = base + lane;
    if (lane < 2 * SD) s_lim[lane] = lane < SD ? p.dof_lower[lane] : p.dof_upper[lane - SD];
    if (lane < SF) s_ft[lane] = p.fingertip_handles[lane];
    wave_sync();

    // ---- loads
    HandFlags fl{0, 0, 0, 0.0f};
    int64_t rnd = 0;
    float obj[13], goal[7];
    constexpr int DOF4 = T * SD * 2 / 4, FORCE4 = T * SD / 4, SENS4 = T * SS / 4, ACT4 = T * SA / 4, GOAL4 = T * 13 / 4;
    v4f r_dof[(DOF4 + WAVE - 1) / WAVE], r_force[(FORCE4 + WAVE - 1) / WAVE], r_sens[(SENS4 + WAVE - 1) / WAVE],
        r_act[(ACT4 + WAVE - 1) / WAVE], r_goal[(GOAL4 + WAVE - 1) / WAVE];
    if (live && REWARD) {
        fl.prog = ld_narrow(b.progress + e);
        fl.rst = ld_narrow(b.reset + e);
        fl.rg = ld_narrow(b.reset_goal + e);
        fl.succ = ld_narrow(b.successes + e);
        if (b.randomize) rnd = ld_narrow(b.randomize + e);
    }
    RowWindow<T, T * NOBS> object_rows;
    object_rows.load(b.root_state_tensor, b.object_indices, e, live, full, lane, obj);
    // fingertip bodies, row q = (environment q / 5, fingertip q % 5).  Whole 13-float states: four lanes
    // share a row (four 16-byte pieces, the last one overlapping), so that the lanes of one load instruction
    // fall into 16 rows instead of 64 - the gather is bound by the number of distinct lines an
    // instruction touches.  Positions only: one lane per row, one 16-byte load.
    constexpr int FT_ROWS_PER_IT = L::FT_W == 13 ? WAVE / 4 : WAVE;
    constexpr int FT_ITERS = (T * SF + FT_ROWS_PER_IT - 1) / FT_ROWS_PER_IT;
    v4f ft[FT_ITERS];
#pragma unroll
    for (int it = 0; it < FT_ITERS; ++it) {
        const int q = it * FT_ROWS_PER_IT + (L::FT_W == 13 ? lane >> 2 : lane), env = q / SF, f = q - env * SF;
        if (env < nvalid) {
            const float *row = b.rigid_body_states + ((base + env) * p.num_bodies + s_ft[f]) * 13;
            if constexpr (L::FT_W == 13) {
                const int part = lane & 3;  // pieces at floats 0, 4, 8 and 9 (the last one overlaps: its .w is float 12)
                ft[it] = ld_stream_u(row + (part < 3 ? 4 * part : 9));
            } else {
                ft[it] = ld_stream_u(row);
            }
        }
    }
    if (full) {
        if (L::DOF) load_chunks<DOF4>(b.dof_state + base * (2 * SD), r_dof, lane);
        if (L::FORCE) {
            load_chunks<FORCE4>(b.dof_force + base * SD, r_force, lane);
            load_chunks<SENS4>(b.vec_sensor_tensor + base * SS, r_sens, lane);
        }
        load_chunks<ACT4>(b.actions + base * SA, r_act, lane);
        load_chunks<GOAL4>(b.goal_states + base * 13, r_goal, lane);
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
            stage_dof_state<SD, NOBS, L::C_POS, L::C_VEL, T>(r_dof, s_tile, s_lim, p.vel_obs_scale, lane);
        } else if constexpr (L::DOF) {  // positions only
#pragma unroll
            for (int it = 0; it < DOF4 / WAVE; ++it) {
                const int pr = 2 * (lane + it * WAVE);
                const int env = pr / SD, j = pr - env * SD;
                s_tile[env * NOBS + L::C_POS + j] = unscale(r_dof[it].x, s_lim[j], s_lim[SD + j]);
                s_tile[env * NOBS + L::C_POS + j + 1] = unscale(r_dof[it].z, s_lim[j + 1], s_lim[SD + j + 1]);
            }
        }
        if constexpr (L::FORCE) {
            stage_rows<SD, L::C_FORCE, NOBS, true, T>(r_force, s_tile, p.force_torque_obs_scale, lane);
            stage_rows<SS, L::C_SENS, NOBS, true, T>(r_sens, s_tile, p.force_torque_obs_scale, lane);
        }
        stage_rows<SA, L::C_ACT, NOBS, false, T>(r_act, s_tile, 1.0f, lane);
    } else {
        for (int i = lane; i < nvalid * SD; i += WAVE) {
            const int env = i / SD, j = i - env * SD;
            if (L::DOF) {
                const v2f pv = reinterpret_cast<const v2f *>(b.dof_state)[base * SD + i];
                s_tile[env * NOBS + L::C_POS + j] = unscale(pv.x, s_lim[j], s_lim[SD + j]);
                if (L::VEL) s_tile[env * NOBS + L::C_VEL + j] = p.vel_obs_scale * pv.y;
            }
            if (L::FORCE) s_tile[env * NOBS + L::C_FORCE + j] = p.force_torque_obs_scale * b.dof_force[base * SD + i];
        }
        if (L::FORCE)
            for (int i = lane; i < nvalid * SS; i += WAVE) {
                const int env = i / SS, k = i - env * SS;
                s_tile[env * NOBS + L::C_SENS + k] = p.force_torque_obs_scale * b.vec_sensor_tensor[base * SS + i];
            }
        for (int i = lane; i < nvalid * SA; i += WAVE) {
            const int env = i / SA, k = i - env * SA;
            s_tile[env * NOBS + L::C_ACT + k] = b.actions[base * SA + i];
        }
    }
#pragma unroll
    for (int it = 0; it < FT_ITERS; ++it) {
        const int q = it * FT_ROWS_PER_IT + (L::FT_W == 13 ? lane >> 2 : lane), env = q / SF, f = q - env * SF;
        if (env < nvalid) {
            float *c = s_tile + env * NOBS + L::C_FT + f * L::FT_W;
            if constexpr (L::FT_W == 13) {
                const int part = lane & 3;
                if (part < 3) {
                    c[4 * part] = ft[it].x; c[4 * part + 1] = ft[it].y; c[4 * part + 2] = ft[it].z; c[4 * part + 3] = ft[it].w;
                } else {
                    c[12] = ft[it].w;
                }
            } else {
                c[0] = ft[it].x; c[1] = ft[it].y; c[2] = ft[it].z;
            }
        }
    }
    wave_sync();

    // ---- per-env: pose columns and reward
    float fin = 0.0f;
    if (live) {
        float *row = s_tile + lane * NOBS;
        const f4 orot{obj[3], obj[4], obj[5], obj[6]};
        const f4 grot{goal[3], goal[4], goal[5], goal[6]};
        const f4 qd = quat_mul(orot, quat_conjugate(grot));
        float *c = row + L::C_OBJ;
#pragma unroll
        for (int i = 0; i < (L::OBJ_W < 7 ? L::OBJ_W : 7); ++i) c[i] = obj[i];
        if (L::OBJ_W == 13) {
            c[7] = obj[7]; c[8] = obj[8]; c[9] = obj[9];
            c[10] = p.vel_obs_scale * obj[10]; c[11] = p.vel_obs_scale * obj[11]; c[12] = p.vel_obs_scale * obj[12];
        }
        if (L::GOAL) {
#pragma unroll
            for (int i = 0; i < 7; ++i) row[L::C_GOAL + i] = goal[i];
        }
        row[L::C_QD] = qd.x; row[L::C_QD + 1] = qd.y; row[L::C_QD + 2] = qd.z; row[L::C_QD + 3] = qd.w;

        if (REWARD) {  // :747-800 compute_hand_reward
            if (b.randomize) st_narrow(b.randomize + e, rnd + 1);
            float act[SA], sq[SA];
            read_cols<SA, L::C_ACT, NOBS>(row, act);
#pragma unroll
            for (int j = 0; j < SA; ++j) sq[j] = act[j] * act[j];
            const float reward = hand_reward(p, obj, goal, qd, row_sum<SA>(sq), fl, fin);
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
    if (REWARD && OBS == IGX_SHADOW_OBS_FULL_STATE && b.states) {  // asymmetric observations: same 211 columns
        drain_rows<NOBS, T>(b.states + base * NOBS, s_tile, nvalid, lane);
        if (b.states_clamped) drain_rows_clamped<NOBS, T>(b.states_clamped + base * NOBS, s_tile, nvalid, lane, p.clip_obs);
    }
    if (REWARD) store_tile_record(b.workspace, fin, live ? fl.rst : 0, lane);
}

static int check_shadow(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n, bool post, const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n);
    if (p->obs_type < 0 || p->obs_type > 3) return fail(IGX_E_PARAM, "%s: unknown obs_type %d", who, p->obs_type);
    if (p->num_bodies < 1) return fail(IGX_E_PARAM, "%s: num_bodies must be positive", who);
    for (int i = 0; i < SF; ++i)
        if (p->fingertip_handles[i] < 0 || p->fingertip_handles[i] >= p->num_bodies)
            return fail(IGX_E_PARAM, "%s: fingertip_handles[%d] = %d outside [0, %d)", who, i, p->fingertip_handles[i],
                        p->num_bodies);
    for (int i = 0; i < SA; ++i) {
        const int j = p->actuated_dof_indices[i];
        if (j < 0 || j >= SD || (i > 0 && j <= p->actuated_dof_indices[i - 1]))
            return fail(IGX_E_PARAM, "%s: actuated_dof_indices must be strictly increasing in [0, %d)", who, SD);
    }
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
        REQ(rigid_body_states, false);
        REQ(consecutive_successes, false);
        REQ(workspace, true);
        if (p->obs_type == IGX_SHADOW_OBS_FULL_STATE || b->states) {
            REQ(dof_force, true);
            REQ(vec_sensor_tensor, true);
        }
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
        if (b->actions_out && !(p->clip_actions >= 0.0f))
            return fail(IGX_E_PARAM, "%s: clip_actions must be >= 0 with actions_out", who);
    }
#undef REQ
    return IGX_OK;
}

template <int OBS, bool REWARD>
static void launch_shadow_post(const igx_shadow_params *p, const igx_shadow_buffers *b, float *obs_out, int64_t n, int reserve,
                               hipStream_t s)
{
    const unsigned tiles = (unsigned)((n + ST - 1) / ST);
    hipLaunchKernelGGL((shadow_post_kernel<OBS, REWARD>), dim3(tiles), dim3(WAVE), (size_t)reserve, s, *p, *b, obs_out, n);
}

// observation / reward launch; leaves the {num_resets, sum(successes * resets)} sums of these
// n environments in record 0 of b->workspace and, with `finalize`, applies them
static int shadow_post_launch(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n, bool finalize,
                              igx_stream_t stream)
{
    static const int reserve = tuning_int("IGX_SHADOW_LDS_RESERVE", 0);  // unused LDS: caps the waves per CU
    hipStream_t s = static_cast<hipStream_t>(stream);
    switch (p->obs_type) {
    case IGX_SHADOW_OBS_OPENAI: launch_shadow_post<IGX_SHADOW_OBS_OPENAI, true>(p, b, b->obs, n, reserve, s); break;
    case IGX_SHADOW_OBS_FULL_NO_VEL: launch_shadow_post<IGX_SHADOW_OBS_FULL_NO_VEL, true>(p, b, b->obs, n, reserve, s); break;
    case IGX_SHADOW_OBS_FULL: launch_shadow_post<IGX_SHADOW_OBS_FULL, true>(p, b, b->obs, n, reserve, s); break;
    default: launch_shadow_post<IGX_SHADOW_OBS_FULL_STATE, true>(p, b, b->obs, n, reserve, s); break;
    }
    int rc = launch_status("igx_shadow_post_physics");
    if (rc != IGX_OK) return rc;
    if (b->states && p->obs_type != IGX_SHADOW_OBS_FULL_STATE) {  // asymmetric critic state beside a smaller policy obs
        launch_shadow_post<IGX_SHADOW_OBS_FULL_STATE, false>(p, b, b->states, n, reserve, s);
        rc = launch_status("igx_shadow_post_physics(states)");
        if (rc != IGX_OK) return rc;
    }
    const int64_t tiles = (n + ST - 1) / ST;
    if (finalize)
        hipLaunchKernelGGL((hand_reduce_kernel<igx_shadow_params, igx_shadow_buffers, true>), dim3(1), dim3(1024), 0, s, *p, *b,
                           tiles);
    else
        hipLaunchKernelGGL((hand_reduce_kernel<igx_shadow_params, igx_shadow_buffers, false>), dim3(1), dim3(1024), 0, s, *p,
                           *b, tiles);
    return launch_status("igx_shadow_post_physics(reduce)");
}

}  // namespace igx

using namespace igx;

extern "C" int igx_shadow_pre_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n, igx_stream_t stream)
{
    int rc = check_shadow(p, b, n, false, "igx_shadow_pre_physics");
    if (rc != IGX_OK || n == 0) return rc;
    const int64_t tiles = (n + WAVE - 1) / WAVE;
    hipLaunchKernelGGL(shadow_pre_kernel, dim3((unsigned)tiles), dim3(WAVE), 0, static_cast<hipStream_t>(stream), *p, *b, n);
    return launch_status("igx_shadow_pre_physics");
}

extern "C" int igx_shadow_post_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n, igx_stream_t stream)
{
    int rc = check_shadow(p, b, n, true, "igx_shadow_post_physics");
    if (rc != IGX_OK || n == 0) return rc;
    return shadow_post_launch(p, b, n, true, stream);
}

extern "C" int igx_shadow_post_physics_shard(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n,
                                             igx_stream_t stream)
{
    int rc = check_shadow(p, b, n, true, "igx_shadow_post_physics_shard");
    if (rc != IGX_OK || n == 0) return rc;
    return shadow_post_launch(p, b, n, false, stream);
}

extern "C" int igx_shadow_finalize_shards(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records,
                                          int32_t num_shards, igx_stream_t stream)
{
    if (!p || !b) return fail(IGX_E_NULL, "igx_shadow_finalize_shards: params/buffers struct is NULL");
    if (num_shards < 1 || num_shards > 65536) return fail(IGX_E_SIZE, "igx_shadow_finalize_shards: num_shards %d", num_shards);
    if (!records || !b->consecutive_successes)
        return fail(IGX_E_NULL, "igx_shadow_finalize_shards: records / consecutive_successes is NULL");
    if (!aligned16(records)) return fail(IGX_E_ALIGN, "igx_shadow_finalize_shards: records must be 16-byte aligned");
    hipLaunchKernelGGL((hand_finalize_kernel<igx_shadow_params, igx_shadow_buffers>), dim3(1), dim3(1), 0,
                       static_cast<hipStream_t>(stream), *p, *b, static_cast<const HandRecord *>(records), (int)num_shards, 1);
    return launch_status("igx_shadow_finalize_shards");
}

extern "C" int igx_shadow_finalize_shards_queued(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records,
                                          int32_t num_shards, int32_t num_steps, igx_stream_t stream)
{
    if (!p || !b) return fail(IGX_E_NULL, "igx_shadow_finalize_shards_queued: params/buffers struct is NULL");
    if (num_shards < 1 || num_shards > 65536) return fail(IGX_E_SIZE, "igx_shadow_finalize_shards_queued: num_shards %d", num_shards);
    if (num_steps < 1 || num_steps > 65536) return fail(IGX_E_SIZE, "igx_shadow_finalize_shards_queued: num_steps %d", num_steps);
    if (!records || !b->consecutive_successes)
        return fail(IGX_E_NULL, "igx_shadow_finalize_shards_queued: records / consecutive_successes is NULL");
    if (!aligned16(records)) return fail(IGX_E_ALIGN, "igx_shadow_finalize_shards_queued: records must be 16-byte aligned");
    hipLaunchKernelGGL((hand_finalize_kernel<igx_shadow_params, igx_shadow_buffers>), dim3(1), dim3(1), 0,
                       static_cast<hipStream_t>(stream), *p, *b, static_cast<const HandRecord *>(records), (int)num_shards, (int)num_steps);
    return launch_status("igx_shadow_finalize_shards_queued");
}
