// Fused post_physics_step of HumanoidImitate (63 dofs, 22 bodies, 205-wide observation).
//
// A wavefront owns a tile of 16 consecutive environments (13 KB of observation rows in
// LDS; a 64-row tile would be 52 KB): the tile's contiguous slices of dof_state, root
// rows and actions are streamed with coalesced 16-byte non-temporal loads, the head of
// every rigid-body row (its position) with one 16-byte load per row, and everything is
// scattered to its observation column in LDS; flagged environments are
// reset by the whole wave (lane j = joint j); four lanes share an environment for the
// reward sums (fp64 where the fp64 motion frame enters, as in the reference); the rows
// leave as one contiguous slab.
//
// The kernel is bound by latency, not by bytes: its time falls in proportion to the waves a
// CU holds, and LDS decides how many that is.  So the tile is the only per-wave LDS: the
// actions (needed for their sum of squares only) pass through the tile's memory BEFORE the
// observations are staged there - their loads are issued first and summed while the other
// streams are still in flight (11 waves per CU instead of 8: 298 -> 267 us at 2^19
// environments).  Workgroups of several waves that share one copy of the unscaled target
// frame would save another 1.5 KB per wave, but their waves start and finish in lockstep,
// which costs more than the extra wave brings (4 waves per workgroup: 298 us, 6: 362 us);
// 8-row tiles (20 waves per CU) are no faster either (266-274 us).
//
// Reference: IsaacGymEnvs/isaacgymenvs/tasks/humanoid_imitate.py:634-643, :579-605,
// :795-831, :672-758; utils/torch_jit_utils.py:670-681.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int HD = IGX_HIM_DOF, HB = IGX_HIM_BODIES, HO = IGX_HIM_OBS;
#ifndef HIM_HT
#define HIM_HT 16
#endif
constexpr int HT = HIM_HT;         // rows per tile
constexpr int LPE = WAVE / HT;     // lanes that share an environment in the reward sums
constexpr int BODY_ROW = HB * 13;  // 286
constexpr int C_VEL = HD, C_ROOT = 2 * HD, C_BODY = 2 * HD + 13;

#ifndef HIM_HW
#define HIM_HW 1
#endif
constexpr int HW = HIM_HW;  // waves (tiles) per workgroup
constexpr int DOF4 = HT * HD * 2 / 4, ROOT4 = HT * 13 / 4, ACT4 = HT * HD / 4;
constexpr int BODY_ROWS = HT * HB, BODY_IT = (BODY_ROWS + WAVE - 1) / WAVE;  // one lane per (env, body) row
constexpr int DOF_IT = (DOF4 + WAVE - 1) / WAVE, ROOT_IT = (ROOT4 + WAVE - 1) / WAVE, ACT_IT = (ACT4 + WAVE - 1) / WAVE;

// One tile.  FULL: all 16 rows exist - straight-line code, every load issued up front in the order the data is
// consumed.  The barrier after the workgroup's constants sits inside, between the issue of the loads and their
// first use; every wave of the workgroup passes it exactly once (a wave without rows included).
template <bool FULL>
__device__ __forceinline__ void him_tile(const igx_humanoid_imitate_params &p, const igx_humanoid_imitate_buffers &b,
                                         const int64_t base, const int nvalid, const int lane, float *s_tile,
                                         double *s_tgt, float *s_lim)
{
    const int t = threadIdx.x;
    // ---- the block's constants: requested first, so they arrive first
    constexpr int NTGT = HD + HB * 3, NC = (NTGT + HW * WAVE - 1) / (HW * WAVE);
    float lim_lo = 0.0f, lim_hi = 0.0f;
    double frame[NC];
    if (t < HD) {
        lim_lo = p.dof_limits_lower[t];
        lim_hi = p.dof_limits_upper[t];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = t + c * HW * WAVE;
        frame[c] = 0.0;
        if (i < HD) {
            frame[c] = b.motion_frame[i];
        } else if (i < NTGT) {
            const int k = i - HD, body = k / 3, x = k - 3 * body;
            frame[c] = b.motion_frame[HD + body * 13 + x];
        }
    }
    const int64_t ef = base + (lane < nvalid ? lane : 0);  // lanes 0..15 carry the per-env scalars
    int64_t prog = 0, rst = 0, rnd = 0;
    if (nvalid > 0) {
        prog = ld_narrow(b.progress + ef);
        rst = ld_narrow(b.reset + ef);
        rnd = ld_narrow((b.randomize ? b.randomize : b.progress) + ef);
    }
    __builtin_amdgcn_sched_barrier(0);
    v4f r_act[ACT_IT], r_dof[DOF_IT], r_root[ROOT_IT], r_body[BODY_IT];
    if (FULL) {
        load_chunks<ACT4>(b.actions + base * HD, r_act, lane);
        __builtin_amdgcn_sched_barrier(0);
        load_chunks<DOF4>(b.dof_state + base * (2 * HD), r_dof, lane);
        load_chunks<ROOT4>(b.root_states + base * 13, r_root, lane);
        // only the position of a body row is observed: one 16-byte load at the head of each 52-byte row
        // (adjacent lanes take adjacent rows, so every line of the slab is still fetched exactly once)
#pragma unroll
        for (int it = 0; it < BODY_IT; ++it) {
            int q = lane + it * WAVE;
            q = q < BODY_ROWS ? q : BODY_ROWS - 1;
            r_body[it] = ld_stream_u(b.rigid_body_states + (base * HB + q) * 13);
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // ---- shared by the block's waves: joint limits, unscaled target pose, target body positions
    if (t < HD) {  // unscale(pose, lower, upper): fp64 numerator, fp32 (upper - lower) widened
        s_lim[t] = lim_lo;
        s_lim[HD + 1 + t] = lim_hi;
        const double span = (double)(lim_hi - lim_lo);
        s_tgt[t] = (2.0 * frame[0] - (double)lim_hi - (double)lim_lo) / span;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = t + c * HW * WAVE;
        if (i >= HD && i < NTGT) s_tgt[i] = frame[c];
    }
    if (HW > 1) __syncthreads(); else wave_sync();
    if (nvalid <= 0) return;

    // ---- the actions pass through the tile's memory first: sum of squares per environment, four lanes each
    const int env4 = lane / LPE, part = lane % LPE;
    float act = 0.0f;
    if (FULL) {
#pragma unroll
        for (int it = 0; it < ACT_IT; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= ACT4 || k < ACT4) *reinterpret_cast<v4f *>(s_tile + 4 * k) = r_act[it];
        }
    } else {
        for (int i = lane; i < nvalid * HD; i += WAVE) s_tile[i] = b.actions[base * HD + i];
    }
    wave_sync();
    if (env4 < nvalid)
        for (int j = part; j < HD; j += LPE) {
            const float a = s_tile[env4 * HD + j];
            act += a * a;
        }
    wave_sync();

    // ---- stage: every input element to its observation column
    if (FULL) {
#pragma unroll
        for (int it = 0; it < DOF_IT; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= DOF4 || k < DOF4) {
                const float pos[2] = {r_dof[it].x, r_dof[it].z}, vel[2] = {r_dof[it].y, r_dof[it].w};
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pr = 2 * k + h, env = pr / HD, j = pr - env * HD;
                    s_tile[env * HO + j] = unscale(pos[h], s_lim[j], s_lim[HD + 1 + j]);
                    s_tile[env * HO + C_VEL + j] = vel[h] * p.dof_vel_scale;
                }
            }
        }
#pragma unroll
        for (int it = 0; it < ROOT_IT; ++it) {
            const int k = lane + it * WAVE;
            if (k < ROOT4) {
                const float x[4] = {r_root[it].x, r_root[it].y, r_root[it].z, r_root[it].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int f = 4 * k + i, env = f / 13, c = f - env * 13;
                    s_tile[env * HO + C_ROOT + c] = x[i];
                }
            }
        }
#pragma unroll
        for (int it = 0; it < BODY_IT; ++it) {
            const int q = lane + it * WAVE;
            if (q < BODY_ROWS) {
                const int env = q / HB, body = q - env * HB;
                float *c = s_tile + env * HO + C_BODY + body * 3;
                c[0] = r_body[it].x; c[1] = r_body[it].y; c[2] = r_body[it].z;
            }
        }
    } else {
        for (int i = lane; i < nvalid * HD; i += WAVE) {
            const int env = i / HD, j = i - env * HD;
            const v2f pv = reinterpret_cast<const v2f *>(b.dof_state)[base * HD + i];
            s_tile[env * HO + j] = unscale(pv.x, s_lim[j], s_lim[HD + 1 + j]);
            s_tile[env * HO + C_VEL + j] = pv.y * p.dof_vel_scale;
        }
        for (int i = lane; i < nvalid * 13; i += WAVE) {
            const int env = i / 13, c = i - env * 13;
            s_tile[env * HO + C_ROOT + c] = b.root_states[base * 13 + i];
        }
        for (int i = lane; i < nvalid * HB * 3; i += WAVE) {
            const int env = i / (HB * 3), k = i - env * (HB * 3), body = k / 3, c = k - 3 * body;
            s_tile[env * HO + C_BODY + k] = b.rigid_body_states[(base + env) * BODY_ROW + body * 13 + c];
        }
    }
    wave_sync();

    // ---- resets (tasks/humanoid_imitate.py:579-605): one flagged environment at a time, lane j = joint j
    uint64_t reset_mask = __ballot(lane < nvalid && rst != 0);
    while (reset_mask) {
        const int src = __builtin_ctzll(reset_mask);
        reset_mask &= reset_mask - 1;
        const int64_t e = base + src;
        if (lane < HD) {
            const int j = lane;
            float up, uv;
            if (b.rand) {
                up = b.rand[e * (2 * HD) + j];
                uv = b.rand[e * (2 * HD) + HD + j];
            } else {
                float q[4];
                philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, j >> 2, q);
                const int a = j & 3;
                up = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
                philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, (HD + j) >> 2, q);
                const int c = (HD + j) & 3;
                uv = c == 0 ? q[0] : c == 1 ? q[1] : c == 2 ? q[2] : q[3];
            }
            const float dp = 0.4f * up + (-0.2f), dv = 0.2f * uv + (-0.1f);
            const float lo = s_lim[j], hi = s_lim[HD + 1 + j];
            const float np_ = tensor_clamp(b.initial_dof_pos[e * HD + j] + dp, lo, hi);
            reinterpret_cast<v2f *>(b.dof_state + e * (2 * HD))[j] = v2f{np_, dv};
            s_tile[src * HO + j] = unscale(np_, lo, hi);
            s_tile[src * HO + C_VEL + j] = dv * p.dof_vel_scale;
        } else {  // lane 63: the per-env scalars of reset_idx
            const float *ir = b.initial_root_states + e * 13;
            const float pot = -norm3(f3{p.targets[0] - ir[0], p.targets[1] - ir[1], 0.0f}) / p.dt;
            b.potentials[e] = pot;
            b.prev_potentials[e] = pot;
        }
    }
    wave_sync();

    // ---- reward: four lanes per environment, strided partial sums, then a 4-lane butterfly
    {
        const int env = env4;
        const float *row = s_tile + env * HO;
        double pose = 0.0, body = 0.0;
        float vel = 0.0f;
        if (env < nvalid) {
            for (int j = part; j < HD; j += LPE) {
                pose += fabs((double)row[j] - s_tgt[j]);
                vel += fabsf(row[C_VEL + j]);
            }
            for (int k = part; k < HB * 3; k += LPE) body += fabs((double)row[C_BODY + k] - s_tgt[HD + k]);
        }
#pragma unroll
        for (int off = 1; off < LPE; off <<= 1) {
            pose += __shfl_xor(pose, off);
            body += __shfl_xor(body, off);
            vel += __shfl_xor(vel, off);
            act += __shfl_xor(act, off);
        }
        // the environment's flags live in lane `env`
        const int64_t prog_e = __shfl(prog, env), rst_e = __shfl(rst, env), rnd_e = __shfl(rnd, env);
        if (part == 0 && env < nvalid) {
            const int64_t e = base + env;
            const float x = row[C_ROOT], y = row[C_ROOT + 1], z = row[C_ROOT + 2];
            const double reward_body = exp(-0.2 * body);
            const double reward_pose = exp(-0.3 * pose);
            const float reward_velocity = expf(-0.05f * vel);
            const double reward_up = (z - p.termination_height > 0.0f) ? (double)p.up_weight : 0.0;
            const float reward_centered = expf(-sqrtf(x * x + y * y));
            double total = 0.0;  // total_reward += term, in the reference's order
            total += reward_body;
            total += reward_pose;
            total += (double)reward_velocity;
            total += reward_up;
            total += (double)(0.1f * reward_centered);
            total += (double)(-(p.actions_cost_scale * act));
            int64_t prog2 = prog_e + 1, rst2 = rst_e;
            if (rst_e != 0) { prog2 = 0; rst2 = 0; }
            if (z < p.termination_height) rst2 = 1;
            if (x * x + y * y > p.termination_radius_sq) rst2 = 1;
            if ((float)prog2 >= p.max_episode_length_m1) rst2 = 1;
            st_narrow(b.rew + e, (float)total);
            st_narrow(b.reset + e, rst2);
            st_narrow(b.progress + e, prog2);
            if (b.randomize) st_narrow(b.randomize + e, rnd_e + 1);
            if (b.time_outs) b.time_outs[e] = ((float)prog2 >= p.max_episode_length_m1 && rst2 != 0) ? 1 : 0;
        }
    }

    // ---- observation rows out as one contiguous slab
    if (FULL && dev_aligned16(b.obs)) {
        constexpr int N4 = HT * HO / 4;
        v4f *o4 = reinterpret_cast<v4f *>(b.obs + base * HO);
#pragma unroll
        for (int it = 0; it < (N4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= N4 || k < N4) st_stream(o4 + k, *reinterpret_cast<const v4f *>(s_tile + 4 * k));
        }
    } else {
        for (int i = lane; i < nvalid * HO; i += WAVE) b.obs[base * HO + i] = s_tile[i];
    }
    if (b.obs_clamped) drain_rows_clamped<HO, HT>(b.obs_clamped + base * HO, s_tile, nvalid, lane, p.clip_obs);
}

__global__ __launch_bounds__(HW * WAVE) void humanoid_imitate_kernel(const igx_humanoid_imitate_params p,
                                                                      const igx_humanoid_imitate_buffers b, int64_t n)
{
    __shared__ __align__(16) float s_tiles[HW][HT * HO];
    __shared__ double s_tgt[HD + HB * 3];  // unscaled target pose, target body positions
    __shared__ float s_lim[2 * HD + 2];
    const int lane = threadIdx.x & (WAVE - 1), w = threadIdx.x / WAVE;
    const int64_t base = ((int64_t)blockIdx.x * HW + w) * HT;
    const int64_t left = n - base;
    const int nvalid = left >= HT ? HT : (left > 0 ? (int)left : 0);
    if (nvalid == HT)
        him_tile<true>(p, b, base, nvalid, lane, s_tiles[w], s_tgt, s_lim);
    else
        him_tile<false>(p, b, base, nvalid, lane, s_tiles[w], s_tgt, s_lim);
}

}  // namespace igx

using namespace igx;

extern "C" int igx_humanoid_imitate_step(const igx_humanoid_imitate_params *p, const igx_humanoid_imitate_buffers *b,
                                         int64_t n, igx_stream_t stream)
{
    const char *who = "igx_humanoid_imitate_step";
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 34)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n);
    if (n == 0) return IGX_OK;
    if (!b->root_states || !b->dof_state || !b->rigid_body_states || !b->actions || !b->initial_root_states ||
        !b->initial_dof_pos || !b->motion_frame || !b->obs || !b->potentials || !b->prev_potentials || !b->rew ||
        !b->reset || !b->progress)
        return fail(IGX_E_NULL, "%s: a required buffer is NULL", who);
    if (b->obs_clamped && (!(p->clip_obs >= 0.0f) || b->obs_clamped == b->obs))
        return fail(IGX_E_PARAM, "%s: obs_clamped needs clip_obs >= 0 and must not alias obs", who);
    if (!aligned16(b->root_states) || !aligned16(b->dof_state) || !aligned16(b->rigid_body_states) || !aligned16(b->actions))
        return fail(IGX_E_ALIGN, "%s: input tensors must be 16-byte aligned", who);
    static const int reserve = tuning_int("IGX_HIM_LDS_RESERVE", 0);  // unused LDS: caps the waves per CU
    const int64_t tiles = (n + HT - 1) / HT;
    hipLaunchKernelGGL(humanoid_imitate_kernel, dim3((unsigned)((tiles + HW - 1) / HW)), dim3(HW * WAVE), (size_t)reserve,
                       static_cast<hipStream_t>(stream), *p, *b, n);
    return launch_status(who);
}
