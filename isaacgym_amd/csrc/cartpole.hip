// Fused post_physics_step for Cartpole: one environment per lane, one 16-byte
// load and one 16-byte store of the 4-float state / observation row per lane.
//
// Reference: tasks/cartpole.py:170-178 (post_physics_step), :148-162
// (reset_idx), :135-146 (compute_observations), :185-201
// (compute_cartpole_reward); paths relative to IsaacGymEnvs/isaacgymenvs/.
#include "igx_device.hpp"
#include "igx_host.hpp"

namespace igx {

__global__ __launch_bounds__(256) void cartpole_step_kernel(const igx_cartpole_params p, const igx_cartpole_buffers b,
                                                            int64_t n)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    float4 *dsp = reinterpret_cast<float4 *>(b.dof_state) + e;
    float4 ds = *dsp;  // (cart_pos, cart_vel, pole_angle, pole_vel)
    int64_t prog = b.progress[e] + 1;
    int64_t rst = b.reset[e];
    if (rst != 0) {
        float u[4];
        if (b.rand) {
            const float4 r = reinterpret_cast<const float4 *>(b.rand)[e];
            u[0] = r.x; u[1] = r.y; u[2] = r.z; u[3] = r.w;
        } else {
            philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, 0, u);
        }
        ds.x = 0.9f * (u[0] - 0.5f);
        ds.z = 0.9f * (u[1] - 0.5f);
        ds.y = 0.5f * (u[2] - 0.5f);
        ds.w = 0.5f * (u[3] - 0.5f);
        *dsp = ds;
        rst = 0;
        prog = 0;
    }
    reinterpret_cast<float4 *>(b.obs)[e] = ds;
    const float half_pi = 1.57079637f;  // float32(np.pi / 2)
    float reward;
    if (p.reward_variant == 1)  // tasks/cartpole_sample.py:322
        reward = 1.0f - 0.001f * ds.z * ds.z - 0.001f * fabsf(ds.y) - 0.001f * fabsf(ds.w);
    else
        reward = 1.0f - ds.z * ds.z - 0.01f * fabsf(ds.y) - 0.005f * fabsf(ds.w);
    const bool out = fabsf(ds.x) > p.reset_dist;
    const bool fell = fabsf(ds.z) > half_pi;
    if (out || fell) {
        reward = -2.0f;
        rst = 1;
    }
    if ((float)prog >= p.max_episode_length_m1) rst = 1;
    b.rew[e] = reward;
    b.reset[e] = rst;
    b.progress[e] = prog;
    if (b.obs_clamped) {
        const float c = p.clip_obs;
        reinterpret_cast<float4 *>(b.obs_clamped)[e] =
            float4{clamp1(ds.x, -c, c), clamp1(ds.y, -c, c), clamp1(ds.z, -c, c), clamp1(ds.w, -c, c)};
    }
    if (b.time_outs) b.time_outs[e] = ((float)prog >= p.max_episode_length_m1 && rst != 0) ? 1 : 0;
}

// reset_idx on an explicit id list: one lane per listed environment
__global__ __launch_bounds__(256) void cartpole_reset_kernel(const igx_cartpole_params p, const igx_cartpole_buffers b,
                                                             const int64_t *__restrict__ env_ids, int64_t m)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const int64_t e = env_ids[i];
    float u[4];
    if (b.rand) {
        const float4 r = reinterpret_cast<const float4 *>(b.rand)[e];
        u[0] = r.x; u[1] = r.y; u[2] = r.z; u[3] = r.w;
    } else {
        philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, 0, u);
    }
    float4 ds;
    ds.x = 0.9f * (u[0] - 0.5f);
    ds.z = 0.9f * (u[1] - 0.5f);
    ds.y = 0.5f * (u[2] - 0.5f);
    ds.w = 0.5f * (u[3] - 0.5f);
    reinterpret_cast<float4 *>(b.dof_state)[e] = ds;
    b.reset[e] = 0;
    b.progress[e] = 0;
}

}  // namespace igx

extern "C" int igx_cartpole_reset(const igx_cartpole_params *p, const igx_cartpole_buffers *b, const int64_t *env_ids,
                                  int64_t m, igx_stream_t stream)
{
    if (!p || !b) return igx::fail(IGX_E_NULL, "igx_cartpole_reset: params/buffers struct is NULL");
    if (m < 0 || m > ((int64_t)1 << 38)) return igx::fail(IGX_E_SIZE, "igx_cartpole_reset: m %lld out of range", (long long)m);
    if (m == 0) return IGX_OK;
    IGX_REQUIRE_PTR(env_ids, "env_ids");
    IGX_REQUIRE_PTR(b->dof_state, "dof_state");
    IGX_REQUIRE_PTR(b->reset, "reset");
    IGX_REQUIRE_PTR(b->progress, "progress");
    IGX_REQUIRE_ALIGNED(b->dof_state, "dof_state");
    if (b->rand) IGX_REQUIRE_ALIGNED(b->rand, "rand");
    hipLaunchKernelGGL(igx::cartpole_reset_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *p, *b, env_ids, m);
    return igx::launch_status("igx_cartpole_reset");
}

extern "C" int igx_cartpole_step(const igx_cartpole_params *p, const igx_cartpole_buffers *b, int64_t n_envs,
                                 igx_stream_t stream)
{
    if (!p || !b) return igx::fail(IGX_E_NULL, "igx_cartpole_step: params/buffers struct is NULL");
    if (n_envs < 0 || n_envs > ((int64_t)1 << 38))
        return igx::fail(IGX_E_SIZE, "igx_cartpole_step: n_envs %lld out of range", (long long)n_envs);
    if (p->reward_variant < 0 || p->reward_variant > 1)
        return igx::fail(IGX_E_PARAM, "igx_cartpole_step: unknown reward_variant %d", p->reward_variant);
    if (n_envs == 0) return IGX_OK;  // empty batch: nothing to validate or launch
    IGX_REQUIRE_PTR(b->dof_state, "dof_state");
    IGX_REQUIRE_PTR(b->obs, "obs");
    IGX_REQUIRE_PTR(b->rew, "rew");
    IGX_REQUIRE_PTR(b->reset, "reset");
    IGX_REQUIRE_PTR(b->progress, "progress");
    IGX_REQUIRE_ALIGNED(b->dof_state, "dof_state");
    IGX_REQUIRE_ALIGNED(b->obs, "obs");
    if (b->rand) IGX_REQUIRE_ALIGNED(b->rand, "rand");
    if (b->obs_clamped) {
        IGX_REQUIRE_ALIGNED(b->obs_clamped, "obs_clamped");
        if (!(p->clip_obs >= 0.0f) || b->obs_clamped == b->obs)
            return igx::fail(IGX_E_PARAM, "igx_cartpole_step: obs_clamped needs clip_obs >= 0 and must not alias obs");
    }
    const int64_t blocks = (n_envs + 255) / 256;
    hipLaunchKernelGGL(igx::cartpole_step_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *p, *b, n_envs);
    return igx::launch_status("igx_cartpole_step");
}
