// Pieces shared by the in-hand reorientation tasks (AllegroHand, ShadowHand): the goal /
// object resets, compute_hand_reward for one environment, the cross-environment sums
// behind consecutive_successes, and the streamed window over evenly spaced actor rows.
//
// Reference: IsaacGymEnvs/isaacgymenvs/tasks/allegro_hand.py:506-522, :663-732 and
// tasks/shadow_hand.py:586-602, :747-813 (the two files hold the same functions).
#pragma once
#include "igx_device.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr float PI_F = 3.14159274f;  // float32(np.pi)

// randomize_rotation
__device__ __forceinline__ f4 randomize_rotation(float r0, float r1)
{
    return quat_mul(quat_from_angle_axis(r0 * PI_F, f3{1.0f, 0.0f, 0.0f}),
                    quat_from_angle_axis(r1 * PI_F, f3{0.0f, 1.0f, 0.0f}));
}

// randomize_rotation_pen (max_angle 0.3)
__device__ __forceinline__ f4 randomize_rotation_pen(float r0)
{
    return quat_mul(quat_from_angle_axis(1.57079637f + r0 * 0.3f, f3{1.0f, 0.0f, 0.0f}),
                    quat_from_angle_axis(r0 * PI_F, f3{0.0f, 0.0f, 1.0f}));
}

// reset_target_pose for environment e, executed by one lane
template <class P, class B>
__device__ __forceinline__ void reset_target_pose(const P &p, const B &b, int64_t e, float r0, float r1)
{
    const f4 rot = randomize_rotation(r0, r1);
    float *gs = b.goal_states + e * 13;
    const float *gi = b.goal_init_state + e * 13;
    float *row = b.root_state_tensor + b.goal_object_indices[e] * 13;
    const float g0 = gi[0], g1 = gi[1], g2 = gi[2];
    gs[0] = g0; gs[1] = g1; gs[2] = g2;
    gs[3] = rot.x; gs[4] = rot.y; gs[5] = rot.z; gs[6] = rot.w;
    row[0] = g0 + p.goal_displacement[0];
    row[1] = g1 + p.goal_displacement[1];
    row[2] = g2 + p.goal_displacement[2];
    row[3] = rot.x; row[4] = rot.y; row[5] = rot.z; row[6] = rot.w;
#pragma unroll
    for (int i = 7; i < 13; ++i) row[i] = 0.0f;
    b.reset_goal[e] = 0;
}

// the object actor's row on an environment reset: init pose + position noise, new rotation, zero velocity
template <class P, class B>
__device__ __forceinline__ void reset_object_row(const P &p, const B &b, int64_t e, float r0, float r1, float r2, float r3,
                                                 float r4)
{
    float *row = b.root_state_tensor + b.object_indices[e] * 13;
    const float *oi = b.object_init_state + e * 13;
    row[0] = oi[0] + p.reset_position_noise * r0;
    row[1] = oi[1] + p.reset_position_noise * r1;
    row[2] = oi[2] + p.reset_position_noise * r2;
    const f4 rot = p.ignore_z_rot ? randomize_rotation_pen(r3) : randomize_rotation(r3, r4);
    row[3] = rot.x; row[4] = rot.y; row[5] = rot.z; row[6] = rot.w;
#pragma unroll
    for (int i = 7; i < 13; ++i) row[i] = 0.0f;
    b.progress[e] = 0;
    b.reset[e] = 0;
    b.successes[e] = 0.0f;
}

// action -> joint target (pre_physics_step)
template <class P>
__device__ __forceinline__ float joint_target(const P &p, float a, float prev, float lo, float hi)
{
    float c;
    if (p.use_relative_control) {
        c = tensor_clamp(prev + p.dof_speed_scale_dt * a, lo, hi);
    } else {
        c = scale(a, lo, hi);
        c = p.act_moving_average * c + p.one_minus_act_moving_average * prev;
        c = tensor_clamp(c, lo, hi);
    }
    return c;
}

struct HandFlags {
    int64_t prog, rst, rg;
    float succ;
};

// compute_hand_reward for one environment.  In: the flags as loaded (prog before its += 1);
// out: updated flags (rst = resets, rg = goal_resets), the reward, and successes * resets.
template <class P>
__device__ __forceinline__ float hand_reward(const P &p, const float (&obj)[13], const float (&goal)[7], f4 qd,
                                             float action_penalty, HandFlags &f, float &fin)
{
    int64_t prog = f.prog + 1;
    const float goal_dist = norm3(f3{obj[0] - goal[0], obj[1] - goal[1], obj[2] - goal[2]});
    const float cn = norm3(f3{qd.x, qd.y, qd.z});
    const float cc = cn > 1.0f ? 1.0f : cn;  // torch.clamp(max=1.0): a NaN stays NaN
    const float rot_dist = 2.0f * asinf(cc);
    const float dist_rew = goal_dist * p.dist_reward_scale;
    const float rot_rew = 1.0f / (fabsf(rot_dist) + p.rot_eps) * p.rot_reward_scale;
    float reward = dist_rew + rot_rew + action_penalty * p.action_penalty_scale;
    // |rot_dist| <= success_tolerance, decided on the asin argument
    const bool success = cc <= p.success_threshold_c;
    const int64_t goal_resets = success ? 1 : f.rg;
    const float succ = f.succ + (float)goal_resets;
    if (goal_resets == 1) reward = reward + p.reach_goal_bonus;
    int64_t resets = f.rst;
    if (goal_dist >= p.fall_dist) {
        reward = reward + p.fall_penalty;
        resets = 1;
    }
    if (p.max_consecutive_successes > 0) {
        if (success) prog = 0;
        if (succ >= (float)p.max_consecutive_successes) resets = 1;
    }
    const bool timed_out = (float)prog >= p.max_episode_length_m1;
    if (timed_out) resets = 1;
    if (p.max_consecutive_successes > 0 && timed_out) reward = reward + 0.5f * p.fall_penalty;
    f.prog = prog;
    f.rst = resets;
    f.rg = goal_resets;
    f.succ = succ;
    fin = succ * (float)resets;
    return reward;
}

// ---- cross-environment sums: 16-byte records, 0 = the reduced sums of the last call, 1.. = one per tile
struct HandRecord {
    unsigned long long num_resets;
    double finished;
};
constexpr int WS_RESULT = 0, WS_TILES = 1;

// wave-level partial sums -> this tile's record
__device__ __forceinline__ void store_tile_record(void *workspace, float fin, int64_t resets, int lane)
{
    float f = fin;
    long long cnt = resets;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        f += __shfl_xor(f, off);
        cnt += __shfl_xor(cnt, off);
    }
    if (lane == 0) {
        HandRecord *ws = reinterpret_cast<HandRecord *>(workspace) + WS_TILES + blockIdx.x;
        ws->num_resets = (unsigned long long)cnt;
        ws->finished = (double)f;
    }
}

// cons = av * finished / num_resets + (1 - av) * cons where num_resets > 0
template <class P, class B>
__device__ __forceinline__ void update_consecutive_successes(const P &p, const B &b, unsigned long long num_resets,
                                                             double finished)
{
    const float fin = (float)finished;
    if (num_resets > 0) {
        const float cons = b.consecutive_successes[0];
        b.consecutive_successes[0] = p.av_factor * fin / (float)(long long)num_resets + p.one_minus_av_factor * cons;
    }
}

// The tiles' records are added in a fixed order (thread t takes records t, t + 1024, ...,
// then a binary tree over the threads), so the sums do not depend on scheduling.
// Result -> record 0 of the workspace; FINALIZE applies it to consecutive_successes.
// (A last-tile-reduces scheme inside the step kernel was measured: the device-scope
// release it needs on every tile costs six times the whole step.)
template <class P, class B, bool FINALIZE>
__global__ __launch_bounds__(1024) void hand_reduce_kernel(const P p, const B b, int64_t num_tiles)
{
    __shared__ unsigned long long s_cnt[1024];
    __shared__ double s_fin[1024];
    HandRecord *ws = reinterpret_cast<HandRecord *>(b.workspace);
    const int t = threadIdx.x;
    unsigned long long cnt = 0;
    double fin = 0.0;
    for (int64_t i = t; i < num_tiles; i += 1024) {
        cnt += ws[WS_TILES + i].num_resets;
        fin += ws[WS_TILES + i].finished;
    }
    s_cnt[t] = cnt;
    s_fin[t] = fin;
    __syncthreads();
    for (int half = 512; half > 0; half >>= 1) {
        if (t < half) {
            s_cnt[t] += s_cnt[t + half];
            s_fin[t] += s_fin[t + half];
        }
        __syncthreads();
    }
    if (t == 0) {
        ws[WS_RESULT].num_resets = s_cnt[0];
        ws[WS_RESULT].finished = s_fin[0];
        if (FINALIZE) {
            update_consecutive_successes(p, b, s_cnt[0], s_fin[0]);
        } else if (b.shard_record) {  // the caller's queue of per-step records
            HandRecord *q = reinterpret_cast<HandRecord *>(b.shard_record);
            q->num_resets = s_cnt[0];
            q->finished = s_fin[0];
        }
    }
}

// consecutive_successes from the records of all shards, added in index order, so every
// rank of a sharded run computes the same bits
template <class P, class B>
__global__ void hand_finalize_kernel(const P p, const B b, const HandRecord *__restrict__ records, int num_records,
                                     int num_steps)
{
    // records[shard * num_steps + step]; steps strictly in order (the moving average is not associative)
    for (int t = 0; t < num_steps; ++t) {
        unsigned long long num_resets = 0;
        double finished = 0.0;
        for (int i = 0; i < num_records; ++i) {
            num_resets += records[(int64_t)i * num_steps + t].num_resets;
            finished += records[(int64_t)i * num_steps + t].finished;
        }
        update_consecutive_successes(p, b, num_resets, finished);
    }
}

// ---- one 13-float actor row per environment of the tile, addressed by an index tensor.
// When the tile's indices are evenly spaced (index = first + s * lane, the layout every
// env-major simulator produces), the window of rows they span is streamed as whole lines
// and the rows are picked out of LDS; any other pattern is gathered row by row (three
// 16-byte loads at a 4-byte-aligned address and one scalar per lane).
// T rows per tile; CAP = floats of LDS the window may use.
template <int T, int CAP>
struct RowWindow {
    static constexpr int MAX_S = (CAP / 13 - 1) / (T - 1) < 4 ? (CAP / 13 - 1) / (T - 1) : 4;
    static constexpr int ITERS = (((MAX_S * (T - 1) + 1) * 13 + 3) / 4 + WAVE - 1) / WAVE;
    static_assert(MAX_S >= 1, "no room for a window");
    v4f r[ITERS];
    float tail;
    int floats;  // 0: gathered
    int stride;

    // issues the loads; `row` receives the gathered row when the tile is not windowed
    __device__ __forceinline__ void load(const float *__restrict__ rows, const int64_t *__restrict__ index, int64_t e,
                                         bool live, bool full, int lane, float (&row)[13])
    {
        const int64_t idx = live ? index[e] : 0;
        const int64_t idx0 = __shfl(idx, 0);
        const int64_t s = __shfl(idx, 1) - idx0;
        const bool windowed = full && s >= 1 && s <= MAX_S && __ballot(lane >= T || idx == idx0 + s * lane) == ~0ull;
        floats = windowed ? ((int)s * (T - 1) + 1) * 13 : 0;
        stride = (int)s;
        tail = 0.0f;
        if (windowed) {
            const float *win = rows + idx0 * 13;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int k = lane + it * WAVE;
                if (4 * k + 4 <= floats) r[it] = ld_stream_u(win + 4 * k);
            }
            // the last one to three floats of the window; reading a whole chunk there could leave the tensor
            const int tail0 = floats & ~3;
            if (tail0 + lane < floats) tail = __builtin_nontemporal_load(win + tail0 + lane);
        } else if (live) {
            const float *orow = rows + idx * 13;
            const v4f o0 = ld_stream_u(orow), o1 = ld_stream_u(orow + 4), o2 = ld_stream_u(orow + 8);
            row[0] = o0.x; row[1] = o0.y; row[2] = o0.z; row[3] = o0.w;
            row[4] = o1.x; row[5] = o1.y; row[6] = o1.z; row[7] = o1.w;
            row[8] = o2.x; row[9] = o2.y; row[10] = o2.z; row[11] = o2.w;
            row[12] = __builtin_nontemporal_load(orow + 12);
        }
    }

    // windowed tiles: through the (free) LDS area s; the branch is uniform over the wave
    __device__ __forceinline__ void pick(float *s, bool live, int lane, float (&row)[13])
    {
        if (floats == 0) return;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int k = lane + it * WAVE;
            if (4 * k + 4 <= floats) *reinterpret_cast<v4f *>(s + 4 * k) = r[it];
        }
        const int tail0 = floats & ~3;
        if (tail0 + lane < floats) s[tail0 + lane] = tail;
        wave_sync();
        if (live) {
            const float *orow = s + stride * lane * 13;
#pragma unroll
            for (int i = 0; i < 13; ++i) row[i] = orow[i];
        }
        wave_sync();
    }
};

}  // namespace igx
