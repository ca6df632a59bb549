// Fused post_physics_step for the Ant and Humanoid locomotion tasks.
//
// One workgroup owns a tile of TILE consecutive environments.  The
// observation rows of the tile live in LDS while they are assembled:
//   phase 1  all lanes stream the tile's AoS inputs with coalesced 16-byte
//            loads and drop the pass-through / scaled columns (dof position and
//            velocity, dof force, sensors, actions) straight into the LDS rows;
//            root-state rows are parked in LDS unchanged;
//   phase 2  one lane per environment reads its root row, does the reset
//            bookkeeping and the quaternion math in registers, writes the 12
//            computed columns, and evaluates the reward from its own LDS row;
//   phase 3  all lanes write the finished rows to obs with coalesced 16-byte
//            stores.
// Every input byte is read once and every output byte written once.
//
// Reference: tasks/ant.py:376-386, :339-366, :463-497, :414-460 and
// tasks/humanoid.py:398-407, :360-390, :489-524, :434-486
// (paths relative to IsaacGymEnvs/isaacgymenvs/).
#include "igx_device.hpp"
#include "igx_host.hpp"

namespace igx {

template <int D, int S, bool HUM>
struct LocoLayout {
    static constexpr int C_POS = 12;
    static constexpr int C_VEL = 12 + D;
    static constexpr int C_FORCE = 12 + 2 * D;                 // Humanoid only
    static constexpr int C_SENS = HUM ? 12 + 3 * D : 12 + 2 * D;
    static constexpr int C_ACT = C_SENS + S;
    static constexpr int NOBS = C_ACT + D;
};


// Calls f(k) for k = tid, tid+TILE, ... < n4.  For a full tile the trip count
// is a compile-time constant so the loop unrolls and all loads issue together.
template <int TILE, bool FULL, int N4_FULL, class F>
__device__ __forceinline__ void for_each_chunk(int n4, int tid, F f)
{
    if constexpr (FULL) {
        constexpr int ITERS = (N4_FULL + TILE - 1) / TILE;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int k = tid + it * TILE;
            if ((it + 1) * TILE <= N4_FULL || k < N4_FULL) f(k);
        }
    } else {
        for (int k = tid; k < n4; k += TILE) f(k);
    }
}

// Stage rows of ROW floats (global, contiguous over the tile) into columns
// C0..C0+ROW-1 of the LDS observation rows, optionally scaled.
template <int ROW, int C0, int NOBS, int TILE, bool FULL, bool SCALE>
__device__ __forceinline__ void stage_rows(const float *__restrict__ g, float *s_tile, int nvalid, float scl, int tid)
{
    const int count = FULL ? TILE * ROW : nvalid * ROW;
    const int n4 = count >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    if constexpr (ROW % 4 == 0 && C0 % 4 == 0 && NOBS % 4 == 0) {
        constexpr int R4 = ROW / 4;
        for_each_chunk<TILE, FULL, TILE * ROW / 4>(n4, tid, [&](int k) {
            float4 v = g4[k];
            if (SCALE) { v.x *= scl; v.y *= scl; v.z *= scl; v.w *= scl; }
            int env = k / R4, part = k - env * R4;
            *reinterpret_cast<float4 *>(s_tile + env * NOBS + C0 + 4 * part) = v;
        });
    } else {
        for_each_chunk<TILE, FULL, TILE * ROW / 4>(n4, tid, [&](int k) {
            float4 v = g4[k];
            float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int flat = 4 * k + i;
                int env = flat / ROW, col = flat - env * ROW;
                s_tile[env * NOBS + C0 + col] = SCALE ? e[i] * scl : e[i];
            }
        });
        if (!FULL)
            for (int flat = (n4 << 2) + tid; flat < count; flat += TILE) {  // <4 trailing floats of a partial tile
            int env = flat / ROW, col = flat - env * ROW;
            float v = g[flat];
            s_tile[env * NOBS + C0 + col] = SCALE ? v * scl : v;
        }
    }
}

// Stage the interleaved (pos, vel) dof_state rows: position -> unscale ->
// column C_POS+j, velocity * dof_vel_scale -> column C_VEL+j.
template <int D, int NOBS, int C_POS, int C_VEL, int TILE, bool FULL>
__device__ __forceinline__ void stage_dof_state(const float *__restrict__ g, float *s_tile, const float *s_lim,
                                                int nvalid, float dof_vel_scale, int tid)
{
    const int count = FULL ? TILE * D * 2 : nvalid * D * 2;  // floats; always even
    const int n4 = count >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for_each_chunk<TILE, FULL, TILE * D * 2 / 4>(n4, tid, [&](int k) {
        float4 v = g4[k];
        float pos[2] = {v.x, v.z}, vel[2] = {v.y, v.w};
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            int pr = 2 * k + h;
            int env = pr / D, j = pr - env * D;
            float lo = s_lim[j], hi = s_lim[D + j];
            s_tile[env * NOBS + C_POS + j] = unscale(pos[h], lo, hi);
            s_tile[env * NOBS + C_VEL + j] = vel[h] * dof_vel_scale;
        }
    });
    if (!FULL) {
        int pr = (n4 << 1) + tid;  // at most one trailing (pos, vel) pair
        if (pr * 2 < count) {
            int env = pr / D, j = pr - env * D;
            float lo = s_lim[j], hi = s_lim[D + j];
            s_tile[env * NOBS + C_POS + j] = unscale(g[2 * pr], lo, hi);
            s_tile[env * NOBS + C_VEL + j] = g[2 * pr + 1] * dof_vel_scale;
        }
    }
}

// Linear global -> LDS copy (root rows).
template <int TILE, bool FULL, int ROW>
__device__ __forceinline__ void copy_in(const float *__restrict__ g, float *s, int nvalid, int tid)
{
    const int count = FULL ? TILE * ROW : nvalid * ROW;
    const int n4 = count >> 2;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    for_each_chunk<TILE, FULL, TILE * ROW / 4>(n4, tid, [&](int k) { *reinterpret_cast<float4 *>(s + 4 * k) = g4[k]; });
    if (!FULL)
        for (int k = (n4 << 2) + tid; k < count; k += TILE) s[k] = g[k];
}

// Linear LDS -> global copy (finished observation rows).
template <int TILE, bool FULL, int ROW>
__device__ __forceinline__ void copy_out(float *__restrict__ g, const float *s, int nvalid, int tid)
{
    const int count = FULL ? TILE * ROW : nvalid * ROW;
    const int n4 = count >> 2;
    float4 *g4 = reinterpret_cast<float4 *>(g);
    for_each_chunk<TILE, FULL, TILE * ROW / 4>(n4, tid, [&](int k) { g4[k] = *reinterpret_cast<const float4 *>(s + 4 * k); });
    if (!FULL)
        for (int k = (n4 << 2) + tid; k < count; k += TILE) g[k] = s[k];
}

// torch.sum(dim=-1) over a row of D floats as the ATen CPU kernel orders it:
// 8 vector lanes over the full blocks of 8, tail first, lanes left to right.
template <int D>
__device__ __forceinline__ float row_sum(const float (&x)[D])
{
    constexpr int NB = D / 8;
    if constexpr (NB == 0) {
        float r = x[0];
#pragma unroll
        for (int i = 1; i < D; ++i) r += x[i];
        return r;
    } else {
        float lane[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float a = x[l];
#pragma unroll
            for (int b = 1; b < NB; ++b) a += x[8 * b + l];
            lane[l] = a;
        }
        constexpr int T = 8 * NB;
        float r;
        if constexpr (T < D) {
            r = x[T];
#pragma unroll
            for (int i = T + 1; i < D; ++i) r += x[i];
            r += lane[0];
        } else {
            r = lane[0];
        }
#pragma unroll
        for (int l = 1; l < 8; ++l) r += lane[l];
        return r;
    }
}

template <int LEN, int C0, int NOBS>
__device__ __forceinline__ void read_cols(const float *row, float (&out)[LEN])
{
    if constexpr (LEN % 4 == 0 && C0 % 4 == 0 && NOBS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < LEN / 4; ++i) {
            float4 v = *reinterpret_cast<const float4 *>(row + C0 + 4 * i);
            out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LEN; ++i) out[i] = row[C0 + i];
    }
}

template <int D, int S, bool HUM, int TILE, bool FULL>
__device__ __forceinline__ void loco_tile(const igx_loco_params &p, const igx_loco_buffers &b, int64_t n, int64_t base,
                                          int nvalid, float *s_tile, float *s_root, float *s_lim)
{
    using L = LocoLayout<D, S, HUM>;
    constexpr int NOBS = L::NOBS;
    const int tid = threadIdx.x;

    // ---- phase 1: coalesced loads, pass-through columns into the LDS rows
    if (tid < 2 * D) s_lim[tid] = (tid < D) ? p.dof_limits_lower[tid] : p.dof_limits_upper[tid - D];

    const int64_t e = base + tid;
    const bool live = FULL || tid < nvalid;
    int64_t prog = 0, rst = 0, rnd = 0;
    float pot = 0.0f;
    if (live) {
        prog = b.progress[e];
        rst = b.reset[e];
        pot = b.potentials[e];
        if (b.randomize) rnd = b.randomize[e];
    }
    copy_in<TILE, FULL, 13>(b.root_states + base * 13, s_root, nvalid, tid);
    stage_rows<S, L::C_SENS, NOBS, TILE, FULL, true>(b.sensors + base * S, s_tile, nvalid, p.contact_force_scale, tid);
    stage_rows<D, L::C_ACT, NOBS, TILE, FULL, false>(b.actions + base * D, s_tile, nvalid, 1.0f, tid);
    if constexpr (HUM)
        stage_rows<D, L::C_FORCE, NOBS, TILE, FULL, true>(b.dof_force + base * D, s_tile, nvalid, p.contact_force_scale, tid);
    __syncthreads();  // s_lim visible
    stage_dof_state<D, NOBS, L::C_POS, L::C_VEL, TILE, FULL>(b.dof_state + base * D * 2, s_tile, s_lim, nvalid,
                                                            p.dof_vel_scale, tid);
    __syncthreads();

    // ---- phase 2: one environment per lane
    if (live) {
        float *row = s_tile + tid * NOBS;
        const float *rr = s_root + tid * 13;
        const f3 pos{rr[0], rr[1], rr[2]};
        const f4 rot{rr[3], rr[4], rr[5], rr[6]};
        const f3 lin{rr[7], rr[8], rr[9]};
        const f3 ang{rr[10], rr[11], rr[12]};

        prog += 1;
        if (b.randomize) b.randomize[e] = rnd + 1;

        if (rst != 0) {  // reset_idx: new joint state, potentials from the initial root position
            float *ds = b.dof_state + e * (2 * D);
            const float *idp = b.initial_dof_pos + e * D;
            // 2D uniforms: D position draws then D velocity draws
            constexpr int NU = 2 * D, NBLK = (NU + 3) / 4;
            float u[NBLK * 4];
            if (b.rand) {
#pragma unroll
                for (int j = 0; j < NU; ++j) u[j] = b.rand[e * NU + j];
            } else {
#pragma unroll
                for (int k = 0; k < NBLK; ++k) {
                    float q[4];
                    philox_uniform4(p.seed, p.rng_offset, e, k, q);
                    u[4 * k] = q[0]; u[4 * k + 1] = q[1]; u[4 * k + 2] = q[2]; u[4 * k + 3] = q[3];
                }
            }
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float dp = 0.4f * u[j] + (-0.2f);      // torch_rand_float(-0.2, 0.2)
                float dv = 0.2f * u[D + j] + (-0.1f);  // torch_rand_float(-0.1, 0.1)
                float lo = s_lim[j], hi = s_lim[D + j];
                float np_ = tensor_clamp(idp[j] + dp, lo, hi);
                reinterpret_cast<float2 *>(ds)[j] = make_float2(np_, dv);
                row[L::C_POS + j] = unscale(np_, lo, hi);
                row[L::C_VEL + j] = dv * p.dof_vel_scale;
            }
            const float *ir = b.initial_root_states + e * 13;
            f3 tt{p.targets[0] - ir[0], p.targets[1] - ir[1], 0.0f};
            pot = -norm3(tt) / p.dt;
            prog = 0;
            rst = 0;
        }

        // compute_*_observations
        const f3 to_target{p.targets[0] - pos.x, p.targets[1] - pos.y, 0.0f};
        const float prev_pot = pot;
        pot = -norm3(to_target) / p.dt;
        const f3 target_dirs = normalize3(to_target);
        const f4 isr{p.inv_start_rot[0], p.inv_start_rot[1], p.inv_start_rot[2], p.inv_start_rot[3]};
        const f4 tq = quat_mul(rot, isr);
        const f3 up_vec = quat_rotate<false>(tq, f3{p.basis_vec1[0], p.basis_vec1[1], p.basis_vec1[2]});
        const f3 heading_vec = quat_rotate<false>(tq, f3{p.basis_vec0[0], p.basis_vec0[1], p.basis_vec0[2]});
        const float up_proj = up_vec.z;
        const float heading_proj = dot3(heading_vec, target_dirs);
        const f3 vel_loc = quat_rotate<true>(tq, lin);
        f3 angvel_loc = quat_rotate<true>(tq, ang);
        float roll, yaw;
        euler_roll_yaw(tq, roll, yaw);
        float angle_to_target = atan2f(p.targets[2] - pos.z, p.targets[0] - pos.x) - yaw;
        if constexpr (HUM) {
            roll = normalize_angle(roll);
            yaw = normalize_angle(yaw);
            angle_to_target = normalize_angle(angle_to_target);
            angvel_loc.x *= p.angular_velocity_scale;
            angvel_loc.y *= p.angular_velocity_scale;
            angvel_loc.z *= p.angular_velocity_scale;
        }
        *reinterpret_cast<float4 *>(row + 0) = make_float4(pos.z, vel_loc.x, vel_loc.y, vel_loc.z);
        *reinterpret_cast<float4 *>(row + 4) = make_float4(angvel_loc.x, angvel_loc.y, angvel_loc.z, yaw);
        *reinterpret_cast<float4 *>(row + 8) = make_float4(roll, angle_to_target, up_proj, heading_proj);

        // compute_*_reward, from this environment's own observation row
        float act[D], dpos[D], dvel[D], tmp[D];
        read_cols<D, L::C_ACT, NOBS>(row, act);
        read_cols<D, L::C_POS, NOBS>(row, dpos);
        read_cols<D, L::C_VEL, NOBS>(row, dvel);
        const float heading_reward = (heading_proj > 0.8f) ? p.heading_weight : p.heading_weight * heading_proj / 0.8f;
        const float up_reward = (up_proj > 0.93f) ? 0.0f + p.up_weight : 0.0f;
#pragma unroll
        for (int j = 0; j < D; ++j) tmp[j] = act[j] * act[j];
        const float actions_cost = row_sum<D>(tmp);
        const float progress_reward = pot - prev_pot;
        float total;
        if constexpr (!HUM) {
#pragma unroll
            for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * dvel[j]);
            const float electricity_cost = row_sum<D>(tmp);
            int cnt = 0;
#pragma unroll
            for (int j = 0; j < D; ++j) cnt += (dpos[j] > 0.99f) ? 1 : 0;
            total = progress_reward + 0.5f + up_reward + heading_reward - p.actions_cost_scale * actions_cost -
                    p.energy_cost_scale * electricity_cost - (float)cnt * p.joints_at_limit_cost_scale;
        } else {
            float ratio[D];
#pragma unroll
            for (int j = 0; j < D; ++j) ratio[j] = p.motor_efforts[j] / p.max_motor_effort;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                float ab = fabsf(dpos[j]);
                float scaled = p.joints_at_limit_cost_scale * (ab - 0.98f) / 0.02f;
                tmp[j] = ((ab > 0.98f) ? 1.0f : 0.0f) * scaled * ratio[j];
            }
            const float dof_at_limit_cost = row_sum<D>(tmp);
#pragma unroll
            for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * dvel[j]) * ratio[j];
            const float electricity_cost = row_sum<D>(tmp);
            total = progress_reward + 2.0f + up_reward + heading_reward - p.actions_cost_scale * actions_cost -
                    p.energy_cost_scale * electricity_cost - dof_at_limit_cost;
        }
        if (pos.z < p.termination_height) {
            total = p.death_cost;
            rst = 1;
        }
        if ((float)prog >= p.max_episode_length_m1) rst = 1;

        b.potentials[e] = pot;
        b.prev_potentials[e] = prev_pot;
        b.rew[e] = total;
        b.reset[e] = rst;
        b.progress[e] = prog;
        float *uv = b.up_vec + e * 3;
        uv[0] = up_vec.x; uv[1] = up_vec.y; uv[2] = up_vec.z;
        float *hv = b.heading_vec + e * 3;
        hv[0] = heading_vec.x; hv[1] = heading_vec.y; hv[2] = heading_vec.z;
        if (b.true_objective) b.true_objective[e] = lin.x;
    }
    __syncthreads();

    // ---- phase 3: finished rows out
    copy_out<TILE, FULL, NOBS>(b.obs + base * NOBS, s_tile, nvalid, tid);
}

template <int D, int S, bool HUM, int TILE>
__global__ __launch_bounds__(TILE) void loco_step_kernel(const igx_loco_params p, const igx_loco_buffers b, int64_t n)
{
    using L = LocoLayout<D, S, HUM>;
    __shared__ __align__(16) float s_tile[TILE * L::NOBS];
    __shared__ __align__(16) float s_root[TILE * 13];
    __shared__ float s_lim[2 * D];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    const int64_t left = n - base;
    if (left >= TILE)
        loco_tile<D, S, HUM, TILE, true>(p, b, n, base, TILE, s_tile, s_root, s_lim);
    else
        loco_tile<D, S, HUM, TILE, false>(p, b, n, base, (int)left, s_tile, s_root, s_lim);
}

static int check_loco(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, int D, int S, bool hum,
                      const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 40)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n);
    if (p->num_dof != D || p->num_sensor_floats != S)
        return fail(IGX_E_PARAM, "%s: expects num_dof=%d num_sensor_floats=%d, got %d/%d", who, D, S, p->num_dof,
                    p->num_sensor_floats);
    if (n == 0) return IGX_OK;  // empty batch: pointers may legitimately be NULL
    const struct { const void *ptr; const char *name; bool vec; } req[] = {
        {b->root_states, "root_states", true}, {b->dof_state, "dof_state", true},
        {b->sensors, "sensors", true}, {b->actions, "actions", true},
        {b->initial_root_states, "initial_root_states", false}, {b->initial_dof_pos, "initial_dof_pos", false},
        {b->obs, "obs", true}, {b->potentials, "potentials", false}, {b->prev_potentials, "prev_potentials", false},
        {b->up_vec, "up_vec", false}, {b->heading_vec, "heading_vec", false}, {b->rew, "rew", false},
        {b->reset, "reset", false}, {b->progress, "progress", false}};
    for (const auto &r : req) {
        if (!r.ptr) return fail(IGX_E_NULL, "%s: %s is NULL", who, r.name);
        if (r.vec && !aligned16(r.ptr)) return fail(IGX_E_ALIGN, "%s: %s must be 16-byte aligned", who, r.name);
    }
    if (hum) {
        if (!b->dof_force) return fail(IGX_E_NULL, "%s: dof_force is NULL", who);
        if (!aligned16(b->dof_force)) return fail(IGX_E_ALIGN, "%s: dof_force must be 16-byte aligned", who);
        if (!(p->max_motor_effort > 0.0f)) return fail(IGX_E_PARAM, "%s: max_motor_effort must be > 0", who);
    }
    return IGX_OK;
}

template <int D, int S, bool HUM, int TILE>
static int launch_loco(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, igx_stream_t stream,
                       const char *who)
{
    if (n == 0) return IGX_OK;
    const int64_t tiles = (n + TILE - 1) / TILE;
    if (tiles > 0x7fffffffLL) return fail(IGX_E_SIZE, "%s: too many tiles", who);
    hipLaunchKernelGGL((loco_step_kernel<D, S, HUM, TILE>), dim3((unsigned)tiles), dim3(TILE), 0,
                       static_cast<hipStream_t>(stream), *p, *b, n);
    return launch_status(who);
}

}  // namespace igx

#ifndef IGX_ANT_TILE
#define IGX_ANT_TILE 256
#endif
#ifndef IGX_HUMANOID_TILE
#define IGX_HUMANOID_TILE 128
#endif

extern "C" int igx_ant_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n_envs, igx_stream_t stream)
{
    int rc = igx::check_loco(p, b, n_envs, 8, 24, false, "igx_ant_step");
    if (rc != IGX_OK) return rc;
    return igx::launch_loco<8, 24, false, IGX_ANT_TILE>(p, b, n_envs, stream, "igx_ant_step");
}

extern "C" int igx_humanoid_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n_envs,
                                 igx_stream_t stream)
{
    int rc = igx::check_loco(p, b, n_envs, 21, 12, true, "igx_humanoid_step");
    if (rc != IGX_OK) return rc;
    return igx::launch_loco<21, 12, true, IGX_HUMANOID_TILE>(p, b, n_envs, stream, "igx_humanoid_step");
}
