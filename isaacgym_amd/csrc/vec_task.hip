// The tail of VecTask.step that every task shares: the time-out flags, the
// clamped copy of the observation (and privileged state) buffer that is handed
// to the learner, and the clamp of the incoming actions.
//
// One launch covers all three streams: the grid is cut into a block range per
// segment, every lane moves 16 bytes with non-temporal accesses.  The observation
// / action noise of the domain-randomisation block is folded into the same passes,
// and so is the scaling of the actions into joint forces or PD targets.
//
// Reference: tasks/base/vec_task.py:381-384 (action noise, action clamp), :404
// (timeout_buf), :407-416 (observation noise, time_outs, clamped obs / states),
// :693-729 (the noise lambdas); tasks/ant.py:368-374, tasks/humanoid.py:392-396,
// tasks/amp/humanoid_amp_base.py:410-422, tasks/cartpole.py:164-168 (action
// scaling); paths relative to IsaacGymEnvs/isaacgymenvs/.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

// Four consecutive floats starting at the multiple of four i0; lanes past `count` read as 0.
__device__ __forceinline__ void load4(const float *__restrict__ p, int64_t i0, int64_t count, bool vec, float (&v)[4])
{
    if (vec && i0 + 4 <= count) {
        const v4f t = ld_stream(reinterpret_cast<const v4f *>(p + i0));
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < count ? p[i0 + k] : 0.0f;
    }
}

__device__ __forceinline__ void store4(float *__restrict__ p, int64_t i0, int64_t count, bool vec, const float (&v)[4])
{
    if (vec && i0 + 4 <= count) {
        st_stream(reinterpret_cast<v4f *>(p + i0), v4f{v[0], v[1], v[2], v[3]});
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i0 + k < count) p[i0 + k] = v[k];
    }
}

// This call's draw for elements 4*quad .. 4*quad+3: uniform in [0,1), or standard
// normal by Box-Muller on two pairs of uniforms.
__device__ __forceinline__ void noise_draw4(const igx_noise &nz, int64_t quad, float (&f)[4])
{
    float u[4];
    philox_uniform4(nz.seed, rng_offset_of(nz.offset, nz.offset_dev), quad, 255, u);
    if (nz.distribution == IGX_NOISE_UNIFORM) {
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = u[k];
    } else {
        const float r0 = sqrtf(-2.0f * logf(1.0f - u[0])), a0 = 6.2831855f * u[1];
        const float r1 = sqrtf(-2.0f * logf(1.0f - u[2])), a1 = 6.2831855f * u[3];
        f[0] = r0 * cosf(a0); f[1] = r0 * sinf(a0);
        f[2] = r1 * cosf(a1); f[3] = r1 * sinf(a1);
    }
}

// x <- noise_lambda(x) for the four elements at i0 (tasks/base/vec_task.py:693-701, :720-727)
__device__ __forceinline__ void apply_noise4(const igx_noise &nz, int64_t i0, int64_t count, bool vec, float (&x)[4])
{
    if (!nz.corr) return;
    float c[4], f[4];
    load4(nz.corr, i0, count, vec, c);
    if (nz.fresh) load4(nz.fresh, i0, count, vec, f);
    else noise_draw4(nz, i0 >> 2, f);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float corr = c[k] * nz.corr_scale + nz.corr_shift;
        const float noise = (corr + f[k] * nz.scale) + nz.shift;
        x[k] = nz.operation == IGX_NOISE_SCALING ? x[k] * noise : x[k] + noise;
    }
}

__device__ __forceinline__ void clamp_segment(const float *__restrict__ in, float *__restrict__ out, int64_t count,
                                              float lo, float hi, int64_t block, bool vec)
{
    // 256 lanes x 4 floats per block
    const int64_t i0 = (block * 256 + threadIdx.x) * 4;
    if (i0 >= count) return;
    float v[4];
    load4(in, i0, count, vec, v);
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = clamp1(v[k], lo, hi);
    store4(out, i0, count, vec, v);
}

__global__ __launch_bounds__(256) void vec_task_tail_kernel(const igx_vec_task_tail t, int64_t n, int64_t obs_blocks,
                                                            int64_t state_blocks, bool obs_vec, bool state_vec)
{
    int64_t blk = blockIdx.x;
    if (blk < obs_blocks) {
        const int64_t i0 = (blk * 256 + threadIdx.x) * 4;
        if (i0 >= t.obs_count) return;
        float v[4];
        load4(t.obs_buf, i0, t.obs_count, obs_vec, v);
        if (t.obs_noise.corr) {
            apply_noise4(t.obs_noise, i0, t.obs_count, obs_vec, v);
            if (t.obs_noised) store4(t.obs_noised, i0, t.obs_count, obs_vec, v);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = clamp1(v[k], -t.clip_obs, t.clip_obs);
        store4(t.obs_out, i0, t.obs_count, obs_vec, v);
        return;
    }
    blk -= obs_blocks;
    if (blk < state_blocks) {
        clamp_segment(t.states_buf, t.states_out, t.states_count, -t.clip_obs, t.clip_obs, blk, state_vec);
        return;
    }
    blk -= state_blocks;
    const int64_t e = blk * 256 + threadIdx.x;
    if (e < n)
        t.time_outs[e] = ((double)t.progress[e] >= t.max_episode_length_m1 && t.reset[e] != 0) ? 1 : 0;
}

__global__ __launch_bounds__(256) void clamp_kernel(const float *__restrict__ in, float *__restrict__ out, int64_t count,
                                                    float lo, float hi, bool vec)
{
    clamp_segment(in, out, count, lo, hi, blockIdx.x, vec);
}

__global__ __launch_bounds__(256) void noise_kernel(const igx_noise nz, const float *__restrict__ x,
                                                    float *__restrict__ out, int64_t count, bool vec)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= count) return;
    float v[4];
    load4(x, i0, count, vec, v);
    apply_noise4(nz, i0, count, vec, v);
    store4(out, i0, count, vec, v);
}

// Actions of the force / PD-target tasks: one lane per four consecutive actions; the per-action gains
// and offsets sit in LDS when they fit (per-element reads of them through the vector memory path queue
// behind the streaming loads).
constexpr int ACT_LDS = 256;

__global__ __launch_bounds__(256) void action_stage_kernel(const igx_action_stage s, int64_t count, bool vec, bool small)
{
    __shared__ float s_gain[ACT_LDS], s_off[ACT_LDS];
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool live = i0 < count;
    const int A = s.num_actions;
    const bool pd = s.mode == IGX_ACT_PD_TARGET;
    float a[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (live) {
        load4(s.actions, i0, count, vec, a);
        apply_noise4(s.noise, i0, count, vec, a);
    }
    const bool staged = A <= ACT_LDS;
    if (staged) {
        if ((int)threadIdx.x < A) {
            s_gain[threadIdx.x] = s.gain[threadIdx.x];
            s_off[threadIdx.x] = pd ? s.offset[threadIdx.x] : 0.0f;
        }
        __syncthreads();
    }
    if (!live) return;
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] = clamp1(a[k], -s.clip_actions, s.clip_actions);
    if (s.actions_out) store4(s.actions_out, i0, count, vec, a);
    int j = small ? (int)((uint32_t)i0 % (uint32_t)A) : (int)(i0 % A);
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float g = staged ? s_gain[j] : s.gain[j];
        const float o = !pd ? 0.0f : (staged ? s_off[j] : s.offset[j]);
        t[k] = pd ? o + g * a[k] : (a[k] * g) * s.power_scale;
        j = j + 1 == A ? 0 : j + 1;
    }
    store4(s.targets, i0, count, vec, t);
}

// Cartpole: one action per environment, written to the first dof of its row; one lane per output element.
__global__ __launch_bounds__(256) void action_first_dof_kernel(const igx_action_stage s, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n * s.num_dof) return;
    const int64_t env = i / s.num_dof;
    const int k = (int)(i - env * s.num_dof);
    float v = 0.0f;
    if (k == 0) {
        const int64_t q0 = env & ~(int64_t)3;
        float a[4];
        load4(s.actions, q0, n, false, a);
        apply_noise4(s.noise, q0, n, false, a);
        const float x = clamp1(a[env & 3], -s.clip_actions, s.clip_actions);
        if (s.actions_out) s.actions_out[env] = x;
        v = x * s.power_scale;
    }
    s.targets[i] = v;
}

}  // namespace igx

using namespace igx;

static int check_noise(const char *who, const igx_noise &nz)
{
    if (!nz.corr) return IGX_OK;
    if (nz.operation != IGX_NOISE_ADDITIVE && nz.operation != IGX_NOISE_SCALING)
        return fail(IGX_E_PARAM, "%s: noise operation %d is neither additive nor scaling", who, nz.operation);
    if (!nz.fresh && nz.distribution != IGX_NOISE_GAUSSIAN && nz.distribution != IGX_NOISE_UNIFORM)
        return fail(IGX_E_PARAM, "%s: noise distribution %d is neither gaussian nor uniform", who, nz.distribution);
    return IGX_OK;
}

static bool noise_aligned(const igx_noise &nz) { return !nz.corr || (aligned16(nz.corr) && (!nz.fresh || aligned16(nz.fresh))); }

extern "C" int igx_vec_task_tail_step(const igx_vec_task_tail *t, int64_t n_envs, igx_stream_t stream)
{
    if (!t) return fail(IGX_E_NULL, "igx_vec_task_tail_step: struct is NULL");
    if (n_envs < 0 || t->obs_count < 0 || t->states_count < 0)
        return fail(IGX_E_SIZE, "igx_vec_task_tail_step: negative count");
    if (t->clip_obs != t->clip_obs || t->clip_obs < 0.0f) return fail(IGX_E_PARAM, "igx_vec_task_tail_step: clip_obs must be >= 0");
    if (t->obs_count && (!t->obs_buf || !t->obs_out)) return fail(IGX_E_NULL, "igx_vec_task_tail_step: obs_buf / obs_out is NULL");
    if (t->states_count && (!t->states_buf || !t->states_out))
        return fail(IGX_E_NULL, "igx_vec_task_tail_step: states_buf / states_out is NULL");
    if (int rc = check_noise("igx_vec_task_tail_step", t->obs_noise)) return rc;
    const bool flags = t->time_outs != nullptr;
    if (flags && n_envs && (!t->progress || !t->reset)) return fail(IGX_E_NULL, "igx_vec_task_tail_step: progress / reset is NULL");
    const int64_t ob = (t->obs_count + 1023) / 1024, sb = (t->states_count + 1023) / 1024;
    const int64_t fb = flags ? (n_envs + 255) / 256 : 0;
    const int64_t blocks = ob + sb + fb;
    if (blocks == 0) return IGX_OK;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_vec_task_tail_step: too large for one launch");
    const bool obs_vec = aligned16(t->obs_buf) && aligned16(t->obs_out) && noise_aligned(t->obs_noise) &&
                         (!t->obs_noised || aligned16(t->obs_noised));
    hipLaunchKernelGGL(vec_task_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       n_envs, ob, sb, obs_vec, aligned16(t->states_buf) && aligned16(t->states_out));
    return launch_status("igx_vec_task_tail_step");
}

extern "C" int igx_noise_apply(const igx_noise *nz, const float *x, float *out, int64_t count, igx_stream_t stream)
{
    if (!nz) return fail(IGX_E_NULL, "igx_noise_apply: struct is NULL");
    if (count < 0) return fail(IGX_E_SIZE, "igx_noise_apply: negative count");
    if (count == 0) return IGX_OK;
    if (!x || !out) return fail(IGX_E_NULL, "igx_noise_apply: NULL argument");
    if (!nz->corr) return fail(IGX_E_NULL, "igx_noise_apply: corr is NULL");
    if (int rc = check_noise("igx_noise_apply", *nz)) return rc;
    const int64_t blocks = (count + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_noise_apply: too large for one launch");
    hipLaunchKernelGGL(noise_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *nz, x, out,
                       count, aligned16(x) && aligned16(out) && noise_aligned(*nz));
    return launch_status("igx_noise_apply");
}

extern "C" int igx_action_stage_step(const igx_action_stage *s, int64_t n_envs, igx_stream_t stream)
{
    if (!s) return fail(IGX_E_NULL, "igx_action_stage_step: struct is NULL");
    if (n_envs < 0) return fail(IGX_E_SIZE, "igx_action_stage_step: negative n_envs");
    if (s->mode != IGX_ACT_FORCE && s->mode != IGX_ACT_PD_TARGET && s->mode != IGX_ACT_FIRST_DOF)
        return fail(IGX_E_PARAM, "igx_action_stage_step: unknown mode");
    if (s->num_actions < 1) return fail(IGX_E_PARAM, "igx_action_stage_step: num_actions must be >= 1");
    if (s->clip_actions != s->clip_actions || s->clip_actions < 0.0f)
        return fail(IGX_E_PARAM, "igx_action_stage_step: clip_actions must be >= 0");
    if (int rc = check_noise("igx_action_stage_step", s->noise)) return rc;
    if (n_envs == 0) return IGX_OK;
    if (!s->actions || !s->targets) return fail(IGX_E_NULL, "igx_action_stage_step: actions / targets is NULL");
    const hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->mode == IGX_ACT_FIRST_DOF) {
        if (s->num_actions != 1 || s->num_dof < 1)
            return fail(IGX_E_PARAM, "igx_action_stage_step: first-dof mode takes one action per env and num_dof >= 1");
        const int64_t blocks = (n_envs * s->num_dof + 255) / 256;
        if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_action_stage_step: too large for one launch");
        hipLaunchKernelGGL(action_first_dof_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *s, n_envs);
        return launch_status("igx_action_stage_step");
    }
    if (!s->gain) return fail(IGX_E_NULL, "igx_action_stage_step: gain is NULL");
    if (s->mode == IGX_ACT_PD_TARGET && !s->offset) return fail(IGX_E_NULL, "igx_action_stage_step: offset is NULL");
    const int64_t count = n_envs * s->num_actions;
    const int64_t blocks = (count + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_action_stage_step: too large for one launch");
    const bool vec = aligned16(s->actions) && aligned16(s->targets) && (!s->actions_out || aligned16(s->actions_out)) &&
                     noise_aligned(s->noise);
    hipLaunchKernelGGL(action_stage_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *s, count, vec,
                       count <= 0xffffffffLL);
    return launch_status("igx_action_stage_step");
}

extern "C" int igx_clamp(const float *x, float lo, float hi, float *out, int64_t count, igx_stream_t stream)
{
    if (count < 0) return fail(IGX_E_SIZE, "igx_clamp: negative count");
    if (count == 0) return IGX_OK;
    if (!x || !out) return fail(IGX_E_NULL, "igx_clamp: NULL argument");
    if (!(lo <= hi)) return fail(IGX_E_PARAM, "igx_clamp: need lo <= hi");
    const int64_t blocks = (count + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_clamp: too large for one launch");
    hipLaunchKernelGGL(clamp_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, out, count,
                       lo, hi, aligned16(x) && aligned16(out));
    return launch_status("igx_clamp");
}
