// The tail of VecTask.step that every task shares: the time-out flags, the
// clamped copy of the observation (and privileged state) buffer that is handed
// to the learner, and the clamp of the incoming actions.
//
// One launch covers all three streams: the grid is cut into a block range per
// segment, every lane moves 16 bytes with non-temporal accesses.
//
// Reference: tasks/base/vec_task.py:384 (action clamp), :404 (timeout_buf),
// :410-416 (time_outs, clamped obs / states); paths relative to
// IsaacGymEnvs/isaacgymenvs/.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

// torch.clamp(x, lo, hi): NaN stays NaN, lo wins over hi never matters (lo <= hi)
__device__ __forceinline__ float clamp1(float x, float lo, float hi)
{
    x = x < lo ? lo : x;
    return x > hi ? hi : x;
}

__device__ __forceinline__ void clamp_segment(const float *__restrict__ in, float *__restrict__ out, int64_t count,
                                              float lo, float hi, int64_t block, bool vec)
{
    // 256 lanes x 4 floats per block
    const int64_t i0 = (block * 256 + threadIdx.x) * 4;
    if (i0 >= count) return;
    if (vec && i0 + 4 <= count) {
        v4f v = ld_stream(reinterpret_cast<const v4f *>(in + i0));
        v.x = clamp1(v.x, lo, hi); v.y = clamp1(v.y, lo, hi); v.z = clamp1(v.z, lo, hi); v.w = clamp1(v.w, lo, hi);
        st_stream(reinterpret_cast<v4f *>(out + i0), v);
    } else {
        for (int64_t i = i0; i < count && i < i0 + 4; ++i) out[i] = clamp1(in[i], lo, hi);
    }
}

__global__ __launch_bounds__(256) void vec_task_tail_kernel(const igx_vec_task_tail t, int64_t n, int64_t obs_blocks,
                                                            int64_t state_blocks, bool obs_vec, bool state_vec)
{
    int64_t blk = blockIdx.x;
    if (blk < obs_blocks) {
        clamp_segment(t.obs_buf, t.obs_out, t.obs_count, -t.clip_obs, t.clip_obs, blk, obs_vec);
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

}  // namespace igx

using namespace igx;

extern "C" int igx_vec_task_tail_step(const igx_vec_task_tail *t, int64_t n_envs, igx_stream_t stream)
{
    if (!t) return fail(IGX_E_NULL, "igx_vec_task_tail_step: struct is NULL");
    if (n_envs < 0 || t->obs_count < 0 || t->states_count < 0)
        return fail(IGX_E_SIZE, "igx_vec_task_tail_step: negative count");
    if (t->clip_obs != t->clip_obs || t->clip_obs < 0.0f) return fail(IGX_E_PARAM, "igx_vec_task_tail_step: clip_obs must be >= 0");
    if (t->obs_count && (!t->obs_buf || !t->obs_out)) return fail(IGX_E_NULL, "igx_vec_task_tail_step: obs_buf / obs_out is NULL");
    if (t->states_count && (!t->states_buf || !t->states_out))
        return fail(IGX_E_NULL, "igx_vec_task_tail_step: states_buf / states_out is NULL");
    const bool flags = t->time_outs != nullptr;
    if (flags && n_envs && (!t->progress || !t->reset)) return fail(IGX_E_NULL, "igx_vec_task_tail_step: progress / reset is NULL");
    const int64_t ob = (t->obs_count + 1023) / 1024, sb = (t->states_count + 1023) / 1024;
    const int64_t fb = flags ? (n_envs + 255) / 256 : 0;
    const int64_t blocks = ob + sb + fb;
    if (blocks == 0) return IGX_OK;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "igx_vec_task_tail_step: too large for one launch");
    hipLaunchKernelGGL(vec_task_tail_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       n_envs, ob, sb, aligned16(t->obs_buf) && aligned16(t->obs_out),
                       aligned16(t->states_buf) && aligned16(t->states_out));
    return launch_status("igx_vec_task_tail_step");
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
