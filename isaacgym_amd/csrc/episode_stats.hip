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

// ---- windows of finished-episode values (RLGPUAlgoObserver.process_infos, utils/rlgames_utils.py:159-174) ----
// Per key: cumulative[e] += value[e]; every done environment, in ascending environment order, appends its
// cumulative value to a ring of the last G finished episodes (the reference's deque(maxlen=games_to_track))
// and restarts from zero.  The ascending order needs each done environment's rank among this step's done
// environments: a count per block of 1024 environments, an exclusive scan of the counts by one workgroup,
// then the apply pass.
struct WindowHeader {
    unsigned long long finished_total;  // episodes appended since the caller zeroed the workspace
    unsigned long long base;            // finished_total before this step (ring position of this step's rank 0)
    unsigned long long last_count;      // done environments of this step
    unsigned long long pad;
};
static_assert(sizeof(WindowHeader) == IGX_EPISODE_WINDOW_HEADER_BYTES, "header size is part of the ABI");

__device__ __forceinline__ int done_count4(const int64_t *done, int64_t e0, int64_t n, bool vec, int (&d)[4])
{
    using v2l = long long __attribute__((ext_vector_type(2)));
    if (vec && e0 + 3 < n) {
        const v2l d0 = *reinterpret_cast<const v2l *>(done + e0);
        const v2l d1 = *(reinterpret_cast<const v2l *>(done + e0) + 1);
        d[0] = d0.x != 0; d[1] = d0.y != 0; d[2] = d1.x != 0; d[3] = d1.y != 0;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) d[k] = (e0 + k < n) ? (done[e0 + k] != 0) : 0;
    }
    return d[0] + d[1] + d[2] + d[3];
}

// exclusive prefix of `c` over the 256 threads of the block; `total` = the block's sum
__device__ __forceinline__ int block_exclusive_scan(int c, int &total)
{
    __shared__ int s_wave[4];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int incl = c;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    int before = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) before += i < w ? s_wave[i] : 0;
    total = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    return before + incl - c;
}

__global__ __launch_bounds__(256) void window_count_kernel(const int64_t *__restrict__ done, int64_t n, uint32_t *counts, bool vec)
{
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int d[4];
    int c = e0 < n ? done_count4(done, e0, n, vec, d) : 0;
    int total;
    block_exclusive_scan(c, total);
    if (threadIdx.x == 0) counts[blockIdx.x] = (uint32_t)total;
}

// one workgroup: counts -> exclusive offsets (in place), header updated
__global__ __launch_bounds__(1024) void window_scan_kernel(WindowHeader *h, uint32_t *counts, int64_t num_blocks)
{
    __shared__ uint32_t s[1024];
    __shared__ unsigned long long s_carry;
    const int t = threadIdx.x;
    if (t == 0) s_carry = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < num_blocks; i0 += 1024) {
        const int64_t i = i0 + t;
        const uint32_t c = i < num_blocks ? counts[i] : 0;
        s[t] = c;
        __syncthreads();
        for (int off = 1; off < 1024; off <<= 1) {  // Hillis-Steele inclusive scan
            const uint32_t add = t >= off ? s[t - off] : 0;
            __syncthreads();
            s[t] += add;
            __syncthreads();
        }
        const unsigned long long carry = s_carry;
        if (i < num_blocks) counts[i] = (uint32_t)(carry + s[t] - c);
        __syncthreads();
        if (t == 1023) s_carry = carry + s[1023];
        __syncthreads();
    }
    if (t == 0) {
        h->base = h->finished_total;
        h->last_count = s_carry;
        h->finished_total += s_carry;
    }
}

__global__ __launch_bounds__(256) void window_apply_kernel(const igx_episode_window w, int64_t n, const WindowHeader *h,
                                                           const uint32_t *offsets, bool vec)
{
    const int64_t e0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    int d[4] = {0, 0, 0, 0};
    const int c = e0 < n ? done_count4(w.done, e0, n, vec, d) : 0;
    int block_total;
    const int before = block_exclusive_scan(c, block_total);
    if (e0 >= n) return;
    const unsigned long long base = h->base, total = h->last_count;
    const unsigned long long G = (unsigned long long)w.games_to_track;
    const unsigned long long first_kept = total > G ? total - G : 0;  // deque(maxlen=G): only the last G of a step survive
    const unsigned long long rank0 = (unsigned long long)offsets[blockIdx.x] + (unsigned long long)before;
    for (int k = 0; k < w.num_keys; ++k) {
        const float *val = w.values[k];
        float *cum = w.cumulative[k];
        float *ring = w.window + (int64_t)k * w.games_to_track;
        float a[4], v[4];
        const bool v4 = vec && e0 + 3 < n && dev_aligned16(val) && dev_aligned16(cum);
        if (v4) {
            const v4f x = *reinterpret_cast<const v4f *>(val + e0), y = *reinterpret_cast<const v4f *>(cum + e0);
            v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
            a[0] = y.x; a[1] = y.y; a[2] = y.z; a[3] = y.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                v[i] = e0 + i < n ? val[e0 + i] : 0.0f;
                a[i] = e0 + i < n ? cum[e0 + i] : 0.0f;
            }
        }
        unsigned long long rank = rank0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = a[i] + v[i];
            if (d[i]) {
                if (rank >= first_kept) ring[(base + rank) % G] = a[i];
                a[i] = 0.0f;
                ++rank;
            }
        }
        if (v4) {
            *reinterpret_cast<v4f *>(cum + e0) = v4f{a[0], a[1], a[2], a[3]};
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (e0 + i < n) cum[e0 + i] = a[i];
        }
    }
}

}  // namespace igx

using namespace igx;

extern "C" int igx_episode_window_step(const igx_episode_window *w, int64_t n_envs, igx_stream_t stream)
{
    const char *who = "igx_episode_window_step";
    if (!w) return fail(IGX_E_NULL, "%s: argument struct is NULL", who);
    if (n_envs < 0 || n_envs > ((int64_t)1 << 38)) return fail(IGX_E_SIZE, "%s: n_envs out of range", who);
    if (w->num_keys < 0 || w->num_keys > IGX_EPISODE_WINDOW_MAX_KEYS)
        return fail(IGX_E_PARAM, "%s: num_keys %d outside [0, %d]", who, w->num_keys, IGX_EPISODE_WINDOW_MAX_KEYS);
    if (w->games_to_track < 1) return fail(IGX_E_PARAM, "%s: games_to_track must be >= 1", who);
    if (n_envs == 0) return IGX_OK;
    if (!w->done || !w->workspace || (w->num_keys > 0 && !w->window)) return fail(IGX_E_NULL, "%s: done / window / workspace is NULL", who);
    for (int k = 0; k < w->num_keys; ++k)
        if (!w->values[k] || !w->cumulative[k]) return fail(IGX_E_NULL, "%s: values / cumulative of key %d is NULL", who, k);
    if (!aligned16(w->workspace)) return fail(IGX_E_ALIGN, "%s: workspace must be 16-byte aligned", who);
    const int64_t blocks = (n_envs + 1023) / 1024;
    hipStream_t s = static_cast<hipStream_t>(stream);
    WindowHeader *h = static_cast<WindowHeader *>(w->workspace);
    uint32_t *counts = reinterpret_cast<uint32_t *>(h + 1);
    const bool vec = aligned16(w->done);
    hipLaunchKernelGGL(window_count_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w->done, n_envs, counts, vec);
    int rc = launch_status(who);
    if (rc != IGX_OK) return rc;
    hipLaunchKernelGGL(window_scan_kernel, dim3(1), dim3(1024), 0, s, h, counts, blocks);
    rc = launch_status(who);
    if (rc != IGX_OK) return rc;
    hipLaunchKernelGGL(window_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, *w, n_envs, h, counts, vec);
    return launch_status(who);
}

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
