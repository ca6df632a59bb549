// Episode statistics: the per-step bookkeeping the learner does on rewards and
// dones after every env step, fused into one streaming kernel plus a
// deterministic reduction, with no nonzero() / .item() host synchronisation.
//
//   current_rewards += rew;  current_lengths += 1;
//   for done envs: n_done += 1, sum_reward += current_rewards, sum_length += current_lengths
//   current_rewards *= (1 - done);  current_lengths *= (1 - done)
//
// Reference call site: IsaacGymEnvs/isaacgymenvs/learning/common_agent.py:284-296
// (game_rewards / game_lengths updates) and utils/rlgames_utils.py:151-174
// (per-done-index Python loop).  Per-block partial sums are written to a
// workspace and added in a fixed order, so the totals are reproducible; the
// totals accumulate on the device until the caller reads them.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

struct EpisodeRecord {
    unsigned long long n_done;
    double sum_reward;
    double sum_length;
    unsigned long long steps;  // kernel invocations accumulated into this record
};
static_assert(sizeof(EpisodeRecord) == IGX_EPISODE_RECORD_BYTES, "record size is part of the ABI");

// one environment's update: rlgames_utils.py:151-211 / common_agent.py:284-296 restated per element
__device__ __forceinline__ void episode_update(int64_t done, float rew, float &cr, float &cl, float &dr, float &dl,
                                               unsigned long long &cnt)
{
    const float not_done = 1.0f - (float)(done != 0 ? 1 : 0);
    cr = cr + rew;
    cl = cl + 1.0f;
    if (not_done == 0.0f) {
        dr += cr;
        dl += cl;
        cnt += 1;
    }
    cr *= not_done;
    cl *= not_done;
}

// Four consecutive environments per lane (16-byte loads and stores when `vec`), one record per block of 1024.
__global__ __launch_bounds__(256) void episode_step_kernel(const igx_episode_buffers b, int64_t n, EpisodeRecord *ws, bool vec)
{
    using v2l = long long __attribute__((ext_vector_type(2)));
    __shared__ unsigned long long s_cnt[4];
    __shared__ double s_rew[4], s_len[4];
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    unsigned long long cnt = 0;
    double r = 0.0, l = 0.0;  // finished episodes of this lane's environments, widened one by one
    if (vec && e0 + 3 < n) {
        const v2l d0 = ld_stream(reinterpret_cast<const v2l *>(b.done + e0));
        const v2l d1 = ld_stream(reinterpret_cast<const v2l *>(b.done + e0) + 1);
        const v4f rw = ld_stream(reinterpret_cast<const v4f *>(b.rew + e0));
        const v4f cr = ld_stream(reinterpret_cast<const v4f *>(b.current_rewards + e0));
        const v4f cl = ld_stream(reinterpret_cast<const v4f *>(b.current_lengths + e0));
        const long long d[4] = {d0.x, d0.y, d1.x, d1.y};
        const float w[4] = {rw.x, rw.y, rw.z, rw.w};
        float c[4] = {cr.x, cr.y, cr.z, cr.w}, len[4] = {cl.x, cl.y, cl.z, cl.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float dr = 0.0f, dl = 0.0f;
            episode_update(d[k], w[k], c[k], len[k], dr, dl, cnt);
            r += (double)dr;
            l += (double)dl;
        }
        st_stream(reinterpret_cast<v4f *>(b.current_rewards + e0), v4f{c[0], c[1], c[2], c[3]});
        st_stream(reinterpret_cast<v4f *>(b.current_lengths + e0), v4f{len[0], len[1], len[2], len[3]});
    } else {
        for (int k = 0; k < 4; ++k) {
            const int64_t e = e0 + k;
            if (e < n) {
                float c = b.current_rewards[e], len = b.current_lengths[e], dr = 0.0f, dl = 0.0f;
                episode_update(b.done[e], b.rew[e], c, len, dr, dl, cnt);
                r += (double)dr;
                l += (double)dl;
                b.current_rewards[e] = c;
                b.current_lengths[e] = len;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_xor(r, off);
        l += __shfl_xor(l, off);
        cnt += __shfl_xor(cnt, off);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_cnt[w] = cnt; s_rew[w] = r; s_len[w] = l; }
    __syncthreads();
    if (threadIdx.x == 0) {
        EpisodeRecord rec;
        rec.n_done = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        rec.sum_reward = ((s_rew[0] + s_rew[1]) + s_rew[2]) + s_rew[3];
        rec.sum_length = ((s_len[0] + s_len[1]) + s_len[2]) + s_len[3];
        rec.steps = 0;
        ws[2 + blockIdx.x] = rec;  // records 0 / 1: running total / this step
    }
}

// fixed-order sum of the block records -> record 1 (this step), added into record 0 (running total)
__global__ __launch_bounds__(1024) void episode_reduce_kernel(EpisodeRecord *ws, int64_t num_blocks)
{
    __shared__ unsigned long long s_cnt[1024];
    __shared__ double s_rew[1024], s_len[1024];
    const int t = threadIdx.x;
    unsigned long long cnt = 0;
    double r = 0.0, l = 0.0;
    for (int64_t i = t; i < num_blocks; i += 1024) {
        cnt += ws[2 + i].n_done;
        r += ws[2 + i].sum_reward;
        l += ws[2 + i].sum_length;
    }
    s_cnt[t] = cnt; s_rew[t] = r; s_len[t] = l;
    __syncthreads();
    for (int half = 512; half > 0; half >>= 1) {
        if (t < half) {
            s_cnt[t] += s_cnt[t + half];
            s_rew[t] += s_rew[t + half];
            s_len[t] += s_len[t + half];
        }
        __syncthreads();
    }
    if (t == 0) {
        ws[1].n_done = s_cnt[0]; ws[1].sum_reward = s_rew[0]; ws[1].sum_length = s_len[0]; ws[1].steps = 1;
        ws[0].n_done += s_cnt[0]; ws[0].sum_reward += s_rew[0]; ws[0].sum_length += s_len[0]; ws[0].steps += 1;
    }
}

}  // namespace igx

using namespace igx;

extern "C" int igx_episode_stats_step(const igx_episode_buffers *b, int64_t n_envs, igx_stream_t stream)
{
    if (!b) return fail(IGX_E_NULL, "igx_episode_stats_step: buffers struct is NULL");
    if (n_envs < 0 || n_envs > ((int64_t)1 << 38)) return fail(IGX_E_SIZE, "igx_episode_stats_step: n_envs out of range");
    if (n_envs == 0) return IGX_OK;
    if (!b->rew || !b->done || !b->current_rewards || !b->current_lengths || !b->workspace)
        return fail(IGX_E_NULL, "igx_episode_stats_step: a buffer is NULL");
    if (!aligned16(b->workspace)) return fail(IGX_E_ALIGN, "igx_episode_stats_step: workspace must be 16-byte aligned");
    const int64_t blocks = (n_envs + 1023) / 1024;  // a quarter of the records the workspace has room for
    hipStream_t s = static_cast<hipStream_t>(stream);
    EpisodeRecord *ws = static_cast<EpisodeRecord *>(b->workspace);
    const bool vec = aligned16(b->rew) && aligned16(b->done) && aligned16(b->current_rewards) && aligned16(b->current_lengths);
    hipLaunchKernelGGL(episode_step_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *b, n_envs, ws, vec);
    int rc = launch_status("igx_episode_stats_step");
    if (rc != IGX_OK) return rc;
    hipLaunchKernelGGL(episode_reduce_kernel, dim3(1), dim3(1024), 0, s, ws, blocks);
    return launch_status("igx_episode_stats_step(reduce)");
}
