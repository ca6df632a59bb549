// HumanoidAMP step / reset and the AMP motion library.
//
//  igx_motion_state      MotionLib.get_motion_state: 16 lanes per sample (12
//                        dof joints, root, key bodies, 2 x dof velocities); the
//                        frame index / blend arithmetic is fp64 in the
//                        reference's operation order, so indices are exact.
//  igx_motion_amp_obs    the same sampling fused with build_amp_observations:
//                        every lane writes its own observation columns, the
//                        interpolated state never touches memory.
//  igx_amp_observations  build_amp_observations on given state arrays.
//  igx_humanoid_amp_step post_physics_step: a wavefront owns 32 environments,
//                        two lanes each.  Rigid-body rows, contact forces, root
//                        rows and the dof_state slab are streamed coalesced and
//                        transposed through the (still empty) observation tile
//                        in phases; the lane pair splits the 12 joints; the
//                        observation rows leave as whole contiguous lines and
//                        the AMP history is shifted with 16-byte loads at
//                        4-byte-aligned addresses.
//  igx_humanoid_amp_reset reset_idx: scatter of sampled reference states and
//                        recomputation of the touched rows (one lane per env).
//
// Reference: IsaacGymEnvs/isaacgymenvs/tasks/humanoid_amp.py:125-134, :146-171,
// :184-204, :279-330, :337-371; tasks/amp/humanoid_amp_base.py:424-438,
// :510-609; tasks/amp/utils_amp/motion_lib.py:83-153, :233-241, :265-293.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int NB = IGX_AMP_NUM_BODIES;   // 15
constexpr int ND = IGX_AMP_NUM_DOF;      // 28
constexpr int NK = IGX_AMP_NUM_KEY_BODIES;
constexpr int NO = IGX_AMP_OBS;          // 105

// tasks/amp/humanoid_amp_base.py:43-44
__device__ constexpr int DOF_BODY_IDS[12] = {1, 2, 3, 4, 6, 7, 9, 10, 11, 12, 13, 14};
__device__ constexpr int DOF_OFFSETS[13] = {0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28};
// first dof_obs column of joint j (6 columns for a 3-dof joint, 1 for a hinge)
__device__ constexpr int DOF_OBS_OFFSETS[12] = {0, 6, 12, 18, 19, 25, 26, 32, 33, 39, 45, 46};

struct FrameBlend {
    int64_t r0, r1;  // table rows of the two frames
    int64_t i0, i1;  // frame indices within the motion
    float blend;
};

// tasks/amp/utils_amp/motion_lib.py:233-241 _calc_frame_blend (fp64 throughout)
__device__ __forceinline__ FrameBlend calc_frame_blend(const igx_motion_tables &t, int64_t id, double time)
{
    const double len = t.motion_length[id], dt = t.motion_dt[id];
    const int64_t nf = t.motion_num_frames[id];
    double phase = time / len;
    phase = phase < 0.0 ? 0.0 : phase;  // np.clip(phase, 0, 1)
    phase = phase > 1.0 ? 1.0 : phase;
    FrameBlend f;
    f.i0 = (int64_t)(phase * (double)(nf - 1));  // astype(int): truncation
    f.i1 = f.i0 + 1 < nf - 1 ? f.i0 + 1 : nf - 1;
    f.blend = (float)((time - (double)f.i0 * dt) / dt);
    const int64_t off = t.frame_offset[id];
    f.r0 = off + f.i0;
    f.r1 = off + f.i1;
    return f;
}

__device__ __forceinline__ f4 load_q(const float *p) { return f4{p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ f3 load_v(const float *p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ f3 lerp3(f3 a, f3 b, float blend)
{
    const float omb = 1.0f - blend;  // (1.0 - blend) * x0 + blend * x1
    return f3{omb * a.x + blend * b.x, omb * a.y + blend * b.y, omb * a.z + blend * b.z};
}

// dof position(s) of joint j from its interpolated local rotation
// (motion_lib.py:265-293 _local_rotation_to_dof); returns the joint size
__device__ __forceinline__ int joint_dof(f4 q, int j, float (&d)[3])
{
    const int size = DOF_OFFSETS[j + 1] - DOF_OFFSETS[j];
    if (size == 3) {
        const f3 e = quat_to_exp_map(q);
        d[0] = e.x; d[1] = e.y; d[2] = e.z;
    } else {
        float theta;
        f3 axis;
        quat_to_angle_axis(q, theta, axis);
        d[0] = normalize_angle(theta * axis.y);  // joint assumed to turn about y
    }
    return size;
}

// dof_to_obs for one joint (humanoid_amp_base.py:510-540): returns #columns
__device__ __forceinline__ int joint_obs(const float (&d)[3], int size, float (&o)[6])
{
    if (size == 3) {
        const f4 q = exp_map_to_quat(f3{d[0], d[1], d[2]});
        const f3 t = quat_rotate<false>(q, f3{1.0f, 0.0f, 0.0f});
        const f3 n = quat_rotate<false>(q, f3{0.0f, 0.0f, 1.0f});
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
        return 6;
    }
    o[0] = d[0];
    return 1;
}

// root part of build_amp_observations (humanoid_amp.py:337-356): columns 0..12
__device__ __forceinline__ void root_obs(f3 root_pos, f4 root_rot, f3 root_vel, f3 root_ang_vel, bool local_root_obs,
                                         f4 &heading_rot, float (&o)[13])
{
    heading_rot = calc_heading_quat_inv(root_rot);
    const f4 rro = local_root_obs ? quat_mul(heading_rot, root_rot) : root_rot;
    const f3 t = quat_rotate<false>(rro, f3{1.0f, 0.0f, 0.0f});
    const f3 n = quat_rotate<false>(rro, f3{0.0f, 0.0f, 1.0f});
    const f3 lv = quat_rotate<false>(heading_rot, root_vel);
    const f3 la = quat_rotate<false>(heading_rot, root_ang_vel);
    o[0] = root_pos.z;
    o[1] = t.x; o[2] = t.y; o[3] = t.z; o[4] = n.x; o[5] = n.y; o[6] = n.z;
    o[7] = lv.x; o[8] = lv.y; o[9] = lv.z;
    o[10] = la.x; o[11] = la.y; o[12] = la.z;
}

// MotionLib.sample_motions (motion_lib.py:61-65: np.random.choice(m, p=weights) = first index whose cumulative
// weight exceeds a uniform draw) and sample_time (:67-78: uniform phase * motion length) from two of the four
// uniforms of one Philox block
__device__ __forceinline__ void draw_motion(const igx_motion_tables &t, const igx_motion_sampler &s, const float (&u)[4],
                                            int64_t &id, double &time)
{
    const double pick = (double)u[0];
    id = 0;
    while (id < t.num_motions - 1 && !(pick < s.cdf[id])) ++id;
    time = s.random_time ? (double)u[1] * t.motion_length[id] : 0.0;
}

// How the rows of a launch map to drawn samples when the kernel tiles over history steps (steps == 0: it does
// not, row s is sample s): row s is step first + s % steps of sample s / steps, evaluated at the sample's time
// + (-dt * step) - the fp64 operations of tasks/humanoid_amp.py:156-159 and :283-284 - and written to output
// row (envs ? envs[sample] : sample) * row_steps + step.
struct StepTiling {
    int steps, first, row_steps;
    double dt;
    const int64_t *envs;
};

// ---------------------------------------------------------------- motion state
// 16 work items per sample: 0..11 dof joints, 12 root, 13 key bodies, 14/15 dof velocities.
// DRAW: ids / times are not inputs - every drawn sample (row / steps) takes its motion and time from the
// sampler's Philox stream, as igx_motion_sample would have written them (and they are written out when the
// pointers are given): a whole fetch_amp_obs_demo in one launch.
template <bool AMP_OBS, bool DRAW = false>
__global__ __launch_bounds__(256) void motion_kernel(const igx_motion_tables t, const int64_t *__restrict__ ids,
                                                     const double *__restrict__ times, int64_t n,
                                                     const igx_motion_state_out o, int local_root_obs,
                                                     float *__restrict__ obs, const int64_t *__restrict__ out_rows,
                                                     int64_t out_row_stride, int64_t out_col_offset,
                                                     const StepTiling tile = {}, const igx_motion_sampler sampler = {},
                                                     int64_t *__restrict__ ids_out = nullptr,
                                                     double *__restrict__ times_out = nullptr)
{
    // a block owns 16 consecutive samples; thread t does work item t / 16 of sample t % 16, so that a wavefront
    // holds four item kinds x 16 samples: three of the block's four waves run the joint path only and one runs
    // root / key bodies / velocities, instead of every wave running all four paths one after the other
    const int64_t s = (int64_t)blockIdx.x * 16 + (threadIdx.x & 15);
    const int item = (int)(threadIdx.x >> 4);
    if (s >= n) return;
    // steps > 1: row s is history step k of drawn sample s / steps, at time t0 + (-dt * k) - the tiling of
    // fetch_amp_obs_demo (tasks/humanoid_amp.py:156-159) done here instead of on the host (fp64, same operations)
    const int64_t src = tile.steps ? s / tile.steps : s;
    const int step = tile.steps ? tile.first + (int)(s - src * tile.steps) : 0;
    int64_t id;
    double time;
    if constexpr (DRAW) {
        float u[4];
        philox_uniform4(sampler.seed, rng_offset_of(sampler.rng_offset, sampler.rng_offset_dev), src + sampler.env_offset, 0, u);
        draw_motion(t, sampler, u, id, time);
        if (item == 15 && s == src * tile.steps) {  // one lane per drawn sample publishes the draw
            if (ids_out) ids_out[src] = id;
            if (times_out) times_out[src] = time;
        }
    } else {
        id = ids[src];
        if (id < 0 || id >= t.num_motions) return;  // not a motion: this sample's output rows are left as they are
        time = times[src];
    }
    int64_t row = out_rows ? out_rows[s] : s;
    if (tile.steps) {
        time = time + (-tile.dt * (double)step);
        row = (tile.envs ? tile.envs[src] : src) * tile.row_steps + step;
    }
    const FrameBlend f = calc_frame_blend(t, id, time);
    float *orow = AMP_OBS ? obs + row * out_row_stride + out_col_offset : nullptr;

    if (item < 12) {
        // items 0..7 are the eight 3-dof joints, 8..11 the four hinges (a nibble table: no memory access), so a
        // wave's four items share one path
        const int j = (int)((0xA753B9864210ULL >> (4 * item)) & 15), body = DOF_BODY_IDS[j];
        const f4 q = slerp(load_q(t.local_rot + (f.r0 * NB + body) * 4), load_q(t.local_rot + (f.r1 * NB + body) * 4),
                           f.blend);
        float d[3];
        const int size = joint_dof(q, j, d);
        if (AMP_OBS) {
            float c[6];
            joint_obs(d, size, c);
            float *dst = orow + 13 + DOF_OBS_OFFSETS[j];
            dst[0] = c[0];
            if (size == 3) { dst[1] = c[1]; dst[2] = c[2]; dst[3] = c[3]; dst[4] = c[4]; dst[5] = c[5]; }
        } else if (o.dof_pos) {
            float *dst = o.dof_pos + s * ND + DOF_OFFSETS[j];
            dst[0] = d[0];
            if (size == 3) { dst[1] = d[1]; dst[2] = d[2]; }
        }
    } else if (item == 12 || item == 13) {
        const f3 root_pos = lerp3(load_v(t.root_pos + f.r0 * 3), load_v(t.root_pos + f.r1 * 3), f.blend);
        const f4 root_rot = slerp(load_q(t.root_rot + f.r0 * 4), load_q(t.root_rot + f.r1 * 4), f.blend);
        if (item == 12) {
            const f3 rv = load_v(t.root_vel + f.r0 * 3), rav = load_v(t.root_ang_vel + f.r0 * 3);
            if (AMP_OBS) {
                f4 heading_rot;
                float c[13];
                root_obs(root_pos, root_rot, rv, rav, local_root_obs != 0, heading_rot, c);
#pragma unroll
                for (int i = 0; i < 13; ++i) orow[i] = c[i];
            } else {
                if (o.root_pos) { o.root_pos[s * 3] = root_pos.x; o.root_pos[s * 3 + 1] = root_pos.y; o.root_pos[s * 3 + 2] = root_pos.z; }
                if (o.root_rot) { o.root_rot[s * 4] = root_rot.x; o.root_rot[s * 4 + 1] = root_rot.y; o.root_rot[s * 4 + 2] = root_rot.z; o.root_rot[s * 4 + 3] = root_rot.w; }
                if (o.root_vel) { o.root_vel[s * 3] = rv.x; o.root_vel[s * 3 + 1] = rv.y; o.root_vel[s * 3 + 2] = rv.z; }
                if (o.root_ang_vel) { o.root_ang_vel[s * 3] = rav.x; o.root_ang_vel[s * 3 + 1] = rav.y; o.root_ang_vel[s * 3 + 2] = rav.z; }
                if (o.frame_idx0) o.frame_idx0[s] = f.i0;
                if (o.frame_idx1) o.frame_idx1[s] = f.i1;
                if (o.blend) o.blend[s] = f.blend;
            }
        } else {
            const f4 heading_rot = AMP_OBS ? calc_heading_quat_inv(root_rot) : f4{0, 0, 0, 1};
            for (int k = 0; k < NK; ++k) {
                const f3 kp = lerp3(load_v(t.key_pos + (f.r0 * NK + k) * 3), load_v(t.key_pos + (f.r1 * NK + k) * 3), f.blend);
                if (AMP_OBS) {
                    const f3 l = quat_rotate<false>(heading_rot, f3{kp.x - root_pos.x, kp.y - root_pos.y, kp.z - root_pos.z});
                    orow[93 + 3 * k] = l.x; orow[94 + 3 * k] = l.y; orow[95 + 3 * k] = l.z;
                } else if (o.key_pos) {
                    o.key_pos[(s * NK + k) * 3] = kp.x; o.key_pos[(s * NK + k) * 3 + 1] = kp.y; o.key_pos[(s * NK + k) * 3 + 2] = kp.z;
                }
            }
        }
    } else {
        const int h = item - 14;  // velocities are not interpolated: frame 0
        float *dst = AMP_OBS ? orow + 65 : o.dof_vel ? o.dof_vel + s * ND : nullptr;
        if (dst)
            for (int i = 14 * h; i < 14 * h + 14; ++i) dst[i] = t.dof_vel[f.r0 * ND + i];
    }
}

// ------------------------------------------------------- amp obs on given arrays
// full observation of one character; dof_pos / dof_vel read with a stride
// (1 for separate arrays, 2 for the interleaved dof_state)
__device__ __forceinline__ void amp_obs_full(f3 root_pos, f4 root_rot, f3 root_vel, f3 root_ang_vel,
                                             const float *dof_pos, const float *dof_vel, int dstride,
                                             const f3 (&key)[NK], bool local_root_obs, float *obs)
{
    f4 heading_rot;
    float c[13];
    root_obs(root_pos, root_rot, root_vel, root_ang_vel, local_root_obs, heading_rot, c);
#pragma unroll
    for (int i = 0; i < 13; ++i) obs[i] = c[i];
#pragma unroll 1
    for (int j = 0; j < 12; ++j) {
        const int off = DOF_OFFSETS[j], size = DOF_OFFSETS[j + 1] - off;
        float d[3] = {dof_pos[off * dstride], 0.0f, 0.0f}, o6[6];
        if (size == 3) {
            d[1] = dof_pos[(off + 1) * dstride];
            d[2] = dof_pos[(off + 2) * dstride];
        }
        joint_obs(d, size, o6);
        float *dst = obs + 13 + DOF_OBS_OFFSETS[j];
        dst[0] = o6[0];
        if (size == 3) { dst[1] = o6[1]; dst[2] = o6[2]; dst[3] = o6[3]; dst[4] = o6[4]; dst[5] = o6[5]; }
    }
#pragma unroll 4
    for (int j = 0; j < ND; ++j) obs[65 + j] = dof_vel[j * dstride];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const f3 l = quat_rotate<false>(heading_rot, f3{key[k].x - root_pos.x, key[k].y - root_pos.y, key[k].z - root_pos.z});
        obs[93 + 3 * k] = l.x; obs[94 + 3 * k] = l.y; obs[95 + 3 * k] = l.z;
    }
}

__global__ __launch_bounds__(64) void amp_obs_kernel(const float *__restrict__ root_states, const float *__restrict__ dof_pos,
                                                     const float *__restrict__ dof_vel, const float *__restrict__ key_pos,
                                                     int local_root_obs, float *__restrict__ obs, int64_t n)
{
    __shared__ __align__(16) float s_tile[WAVE * NO];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * WAVE;
    const int nvalid = (n - base >= WAVE) ? WAVE : (int)(n - base);
    if (lane < nvalid) {
        const int64_t e = base + lane;
        const float *r = root_states + e * 13;
        f3 key[NK];
        for (int k = 0; k < NK; ++k) key[k] = load_v(key_pos + (e * NK + k) * 3);
        amp_obs_full(load_v(r), load_q(r + 3), load_v(r + 7), load_v(r + 10), dof_pos + e * ND, dof_vel + e * ND, 1, key,
                     local_root_obs != 0, s_tile + lane * NO);
    }
    wave_sync();
    if (nvalid == WAVE && dev_aligned16(obs)) {
        drain_rows<NO>(obs + base * NO, s_tile, nvalid, lane);
    } else {
        for (int i = lane; i < nvalid * NO; i += WAVE) obs[base * NO + i] = s_tile[i];
    }
}

// ------------------------------------------------------------------ env step
__device__ __forceinline__ bool is_contact_body(const igx_amp_params &p, int body)
{
    bool c = false;
    for (int i = 0; i < p.num_contact_bodies; ++i) c |= p.contact_body_ids[i] == body;
    return c;
}

// AMP history of a group of G environments (rows of KK * 105 floats, newest first):
// every slot moves one place back, slot 0 takes the new observation from the LDS tile.
// All of the group's source chunks are in registers before its first store; a row is
// read and written by its own group only.
template <int KK, int G>
__device__ __forceinline__ void shift_history(float *__restrict__ amp, const float *s_tile, int env0, int lane)
{
    constexpr int ROW = KK * NO;
    constexpr int N4 = G * ROW / 4;
    constexpr int ITERS = (N4 + WAVE - 1) / WAVE;
    static_assert((G * ROW) % 4 == 0, "group slab must be whole 16-byte chunks");
    float *slab = amp + (int64_t)env0 * ROW;
    v4f old[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        int k = lane + it * WAVE;
        k = k < N4 ? k : N4 - 1;
        // position g of the slab takes the old value NO floats earlier (same row) unless it is in slot 0;
        // a chunk that lies in slot 0 entirely has nothing to fetch
        const int g = 4 * k, r_first = g % ROW, r_last = (g + 3) % ROW;
        const bool any_old = r_first >= NO || r_last >= NO;
        old[it] = v4f{0.0f, 0.0f, 0.0f, 0.0f};
        if (it == 0 && g < NO) {  // would start before the slab: element-wise
            if (g + 3 >= NO) {
                float o[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = g + i >= NO ? __builtin_nontemporal_load(slab + g + i - NO) : 0.0f;
                old[it] = v4f{o[0], o[1], o[2], o[3]};
            }
        } else if (any_old) {
            old[it] = ld_stream_u(slab + g - NO);
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): every load has returned before the stores below
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int k = lane + it * WAVE;
        if ((it + 1) * WAVE <= N4 || k < N4) {
            const float o[4] = {old[it].x, old[it].y, old[it].z, old[it].w};
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = 4 * k + i;
                const int env = g / ROW, r = g - env * ROW;
                v[i] = r < NO ? s_tile[(env0 + env) * NO + r] : o[i];
            }
            st_stream(reinterpret_cast<v4f *>(slab) + k, v4f{v[0], v[1], v[2], v[3]});
        }
    }
}

// KK below = compile-time number of history steps (0: any, slower path)

// ---- 32-row tiles, two lanes per environment ---------------------------------------------
// Lane l and lane l + 32 share environment l of the tile: the lower lane computes the root
// columns and joints {0,1,2,4 | 3,5}, the upper lane joints {6,8,9,11 | 7,10} and the key-body
// columns, so both lanes meet a 3-dof joint (or a hinge) in the same loop iteration.  Half the
// LDS per wave and half the per-environment latency of the one-lane-per-row kernel above.
constexpr int PAIR_T = 32;
constexpr int POS_STRIDE = ND + 1;  // odd stride: conflict-free row reads
__device__ constexpr int PAIR_DOF_OFF[2][6] = {{0, 3, 6, 10, 9, 13}, {14, 18, 21, 25, 17, 24}};
__device__ constexpr int PAIR_OBS_OFF[2][6] = {{0, 6, 12, 19, 18, 25}, {26, 33, 39, 46, 32, 45}};

template <int KK>
__global__ __launch_bounds__(WAVE) void amp_step_pair_kernel(const igx_amp_params p, const igx_amp_buffers b, int64_t n)
{
    constexpr int T = PAIR_T;
    constexpr int BODY_ROW = NB * 13;
    constexpr int Q = T / 2;  // rigid-body rows per phase
    __shared__ __align__(16) float s_tile[T * NO];
    __shared__ float s_pos[T * POS_STRIDE];
    __shared__ float s_key[T * NK * 3];
    __shared__ unsigned s_flag[T];  // bit 0: a body below the termination height, bit 1: a contact force
    static_assert(Q * BODY_ROW <= T * NO, "half a tile of rigid-body rows fits the tile");
    static_assert(T * (NB * 3 + 13) <= T * NO, "contact + root rows fit the tile");
    const int lane = threadIdx.x;
    const int el = lane & (T - 1), part = lane >> 5;
    const int K = KK ? KK : p.num_amp_obs_steps;
    const int64_t base = (int64_t)blockIdx.x * T;
    const int nvalid = (n - base >= T) ? T : (int)(n - base);
    const bool full = nvalid == T;
    const bool live = el < nvalid;
    const int64_t e = base + (live ? el : 0);

    int64_t prog = ld_narrow(b.progress + e);
    if (lane < T) s_flag[lane] = 0;

    constexpr int B4 = Q * BODY_ROW / 4, CF4 = T * NB * 3 / 4, ROOT4 = T * 13 / 4, DOF4 = T * 2 * ND / 4;
    v4f r_body[2][(B4 + WAVE - 1) / WAVE], r_cf[(CF4 + WAVE - 1) / WAVE], r_root[(ROOT4 + WAVE - 1) / WAVE],
        r_dof[DOF4 / WAVE];
    if (full) {
        load_chunks<B4>(b.rigid_body_state + base * BODY_ROW, r_body[0], lane);
        load_chunks<B4>(b.rigid_body_state + (base + Q) * BODY_ROW, r_body[1], lane);
    }
    // ---- rigid-body rows, 16 environments at a time: lane (row, quarter) takes 4 / 3 body heights
    // and one key-body position
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int cnt = nvalid - h * Q < 0 ? 0 : (nvalid - h * Q > Q ? Q : nvalid - h * Q);
        if (full) {
#pragma unroll
            for (int it = 0; it < (B4 + WAVE - 1) / WAVE; ++it) {
                const int k = lane + it * WAVE;
                if ((it + 1) * WAVE <= B4 || k < B4) *reinterpret_cast<v4f *>(s_tile + 4 * k) = r_body[h][it];
            }
        } else {
            for (int i = lane; i < cnt * BODY_ROW; i += WAVE) s_tile[i] = b.rigid_body_state[(base + h * Q) * BODY_ROW + i];
        }
        if (h == 0 && full) {
            load_chunks<CF4>(b.contact_forces + base * (NB * 3), r_cf, lane);
            load_chunks<ROOT4>(b.root_states + base * 13, r_root, lane);
        }
        if (h == 1 && full) load_chunks<DOF4>(b.dof_state + base * (2 * ND), r_dof, lane);
        wave_sync();
        const int er = lane & (Q - 1), quarter = lane >> 4;
        if (er < cnt) {
            const float *row = s_tile + er * BODY_ROW;
            bool low = false;
            const int b0 = 4 * quarter, b1 = b0 + 4 < NB ? b0 + 4 : NB;
            for (int body = b0; body < b1; ++body)
                low |= (row[body * 13 + 2] < p.termination_height) && !is_contact_body(p, body);
            if (low) atomicOr(&s_flag[h * Q + er], 1u);
            const float *kp = row + p.key_body_ids[quarter] * 13;
            float *dst = s_key + (h * Q + er) * (NK * 3) + quarter * 3;
            dst[0] = kp[0]; dst[1] = kp[1]; dst[2] = kp[2];
        }
        wave_sync();
    }

    // ---- contact forces and root rows through the tile
    if (full) {
#pragma unroll
        for (int it = 0; it < (CF4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= CF4 || k < CF4) *reinterpret_cast<v4f *>(s_tile + 4 * k) = r_cf[it];
        }
#pragma unroll
        for (int it = 0; it < (ROOT4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= ROOT4 || k < ROOT4) *reinterpret_cast<v4f *>(s_tile + T * NB * 3 + 4 * k) = r_root[it];
        }
    } else {
        for (int i = lane; i < nvalid * NB * 3; i += WAVE) s_tile[i] = b.contact_forces[base * (NB * 3) + i];
        for (int i = lane; i < nvalid * 13; i += WAVE) s_tile[T * NB * 3 + i] = b.root_states[base * 13 + i];
    }
    wave_sync();
    float rr[13];
    {
        const float *cf = s_tile + el * (NB * 3);
        bool hit_any = false;
        const int b0 = part ? 8 : 0, b1 = part ? NB : 8;
        for (int body = b0; body < b1; ++body) {
            const bool hit = cf[3 * body] > 0.1f || cf[3 * body + 1] > 0.1f || cf[3 * body + 2] > 0.1f;
            hit_any |= hit && !is_contact_body(p, body);
        }
        if (live && hit_any) atomicOr(&s_flag[el], 2u);
#pragma unroll
        for (int i = 0; i < 13; ++i) rr[i] = s_tile[T * NB * 3 + el * 13 + i];
    }
    wave_sync();

    // ---- dof_state: velocities to their final columns, positions to the side buffer
    if (full) {
#pragma unroll
        for (int it = 0; it < DOF4 / WAVE; ++it) {
            const int k = lane + it * WAVE;
            const int pr = 2 * k;
            const int env = pr / ND, j = pr - env * ND;
            s_pos[env * POS_STRIDE + j] = r_dof[it].x; s_pos[env * POS_STRIDE + j + 1] = r_dof[it].z;
            s_tile[env * NO + 65 + j] = r_dof[it].y; s_tile[env * NO + 65 + j + 1] = r_dof[it].w;
        }
    } else {
        for (int i = lane; i < nvalid * ND; i += WAVE) {
            const int env = i / ND, j = i - env * ND;
            const v2f pv = reinterpret_cast<const v2f *>(b.dof_state)[base * ND + i];
            s_pos[env * POS_STRIDE + j] = pv.x;
            s_tile[env * NO + 65 + j] = pv.y;
        }
    }
    wave_sync();

    // ---- per-env, split over the lane pair
    if (live) {
        float *row = s_tile + el * NO;
        const float *pos = s_pos + el * POS_STRIDE;
        const f3 root_pos{rr[0], rr[1], rr[2]};
        const f4 root_rot{rr[3], rr[4], rr[5], rr[6]};
        f4 heading_rot;
        if (part == 0) {
            float c[13];
            root_obs(root_pos, root_rot, f3{rr[7], rr[8], rr[9]}, f3{rr[10], rr[11], rr[12]}, p.local_root_obs != 0,
                     heading_rot, c);
#pragma unroll
            for (int i = 0; i < 13; ++i) row[i] = c[i];
        } else {
            heading_rot = calc_heading_quat_inv(root_rot);
            const float *key = s_key + el * (NK * 3);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const f3 l = quat_rotate<false>(heading_rot, f3{key[3 * k] - root_pos.x, key[3 * k + 1] - root_pos.y,
                                                                key[3 * k + 2] - root_pos.z});
                row[93 + 3 * k] = l.x; row[94 + 3 * k] = l.y; row[95 + 3 * k] = l.z;
            }
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const int off = part ? PAIR_DOF_OFF[1][i] : PAIR_DOF_OFF[0][i];
            float *dst = row + 13 + (part ? PAIR_OBS_OFF[1][i] : PAIR_OBS_OFF[0][i]);
            if (i < 4) {  // 3-dof joints: exp map -> tangent / normal columns
                float o6[6];
                const float d[3] = {pos[off], pos[off + 1], pos[off + 2]};
                joint_obs(d, 3, o6);
                dst[0] = o6[0]; dst[1] = o6[1]; dst[2] = o6[2]; dst[3] = o6[3]; dst[4] = o6[4]; dst[5] = o6[5];
            } else {
                dst[0] = pos[off];
            }
        }
        if (part == 0) {
            prog += 1;
            const unsigned f = s_flag[el];
            int64_t terminated = 0;
            if (p.enable_early_termination && (f & 2u) && (f & 1u) && prog > 1) terminated = 1;
            const int64_t eo = base + el;
            st_narrow(b.rew + eo, 1.0f);
            st_narrow(b.reset + eo, ((float)prog >= p.max_episode_length_m1) ? (int64_t)1 : terminated);
            st_narrow(b.terminate + eo, terminated);
            st_narrow(b.progress + eo, prog);
            // reset is 1 whenever the episode length is reached, so the time-out flag is that comparison
            if (b.time_outs) b.time_outs[eo] = ((float)prog >= p.max_episode_length_m1) ? 1 : 0;
        }
    }
    wave_sync();

    // ---- observation rows out
    if (full && dev_aligned16(b.obs)) {
        constexpr int N4 = T * NO / 4;
        v4f *o4 = reinterpret_cast<v4f *>(b.obs + base * NO);
#pragma unroll
        for (int it = 0; it < (N4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= N4 || k < N4) st_stream(o4 + k, *reinterpret_cast<const v4f *>(s_tile + 4 * k));
        }
    } else {
        for (int i = lane; i < nvalid * NO; i += WAVE) b.obs[base * NO + i] = s_tile[i];
    }
    if (b.obs_clamped) drain_rows_clamped<NO, T>(b.obs_clamped + base * NO, s_tile, nvalid, lane, p.clip_obs);

    // ---- AMP history
    float *amp = b.amp_obs + base * K * NO;
    if (KK != 0 && full && dev_aligned16(b.amp_obs)) {
        constexpr int G = 16;  // 8 rows per group: 577 us, 32 (one round trip per tile): 561 us, 16: 558-560 us at 2^20 envs
#pragma unroll 1
        for (int g0 = 0; g0 < T; g0 += G) shift_history<KK ? KK : 2, G>(amp, s_tile, g0, lane);
    } else {
        for (int k = K - 2; k >= 0; --k)
            for (int i = lane; i < nvalid * NO; i += WAVE) {
                const int env = i / NO, c = i - env * NO;
                amp[(env * K + k + 1) * NO + c] = amp[(env * K + k) * NO + c];
            }
        for (int i = lane; i < nvalid * NO; i += WAVE) {
            const int env = i / NO, c = i - env * NO;
            amp[(env * K) * NO + c] = s_tile[i];
        }
    }
}

// reset_idx on a list of environments (negative ids are holes of a masked list, igx_motion_sample_flagged).
// A block of four waves owns `per_block` list entries (64 for a long, mostly empty masked list; 8 for a short
// compact one, so that its environments spread over many waves); every wave reads the block's entries and takes
// every fourth environment among them, one after the other, with all its lanes: the new state passes through
// LDS on its way into the simulator tensors, the observation is built from that copy with one work item per
// lane (12 joints, root, key bodies, 28 joint velocities - the arithmetic of amp_obs_full), and the row leaves as
// whole lines into obs and every history slot.  (One lane per environment, as the kernel first did, left 63
// lanes idle behind ~900 scalar accesses: 96 us for 3 000 resets among 2^20 environments.)
constexpr int RESET_WAVES = 4;

__global__ __launch_bounds__(64 * RESET_WAVES) void amp_reset_kernel(const igx_amp_params p, const igx_amp_buffers b,
                                                                     const int64_t *__restrict__ env_ids, int64_t m,
                                                                     const igx_motion_state_out s, int per_block)
{
    __shared__ float s_root_w[RESET_WAVES][13], s_dof_w[RESET_WAVES][2 * ND], s_key_w[RESET_WAVES][NK * 3];
    __shared__ __align__(16) float s_obs_w[RESET_WAVES][NO + 3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *s_root = s_root_w[wave], *s_dof = s_dof_w[wave], *s_key = s_key_w[wave], *s_obs = s_obs_w[wave];
    const int64_t i0 = (int64_t)blockIdx.x * per_block;
    const int K = p.num_amp_obs_steps;
    const int64_t mine = (lane < per_block && i0 + lane < m) ? env_ids[i0 + lane] : -1;
    uint64_t todo = __ballot(mine >= 0);
    for (int rank = 0; todo; ++rank) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        if ((rank & (RESET_WAVES - 1)) != wave) continue;  // wave-uniform
        const int64_t e = __shfl(mine, src);
        const int64_t i = i0 + src;
        float *r = b.root_states + e * 13;
        float *ds = b.dof_state + e * (2 * ND);
        if (lane < 13) {  // root row: position 3, rotation 4, velocity 3, angular velocity 3
            const float *from = lane < 3 ? s.root_pos : lane < 7 ? s.root_rot : lane < 10 ? s.root_vel : s.root_ang_vel;
            const int c = lane < 3 ? lane : lane < 7 ? lane - 3 : lane < 10 ? lane - 7 : lane - 10;
            const int w = (lane >= 3 && lane < 7) ? 4 : 3;
            const float v = from ? from[i * w + c] : r[lane];
            s_root[lane] = v;
            if (from) r[lane] = v;
        }
        if (lane < 2 * ND) {  // interleaved (position, velocity) pairs
            const float *from = (lane & 1) ? s.dof_vel : s.dof_pos;
            const float v = from ? from[i * ND + (lane >> 1)] : ds[lane];
            s_dof[lane] = v;
            if (from) ds[lane] = v;
        }
        if (lane < NK * 3) {
            const int k = lane / 3;
            s_key[lane] = b.rigid_body_state[(e * NB + p.key_body_ids[k]) * 13 + (lane - 3 * k)];
        }
        if (lane == 63) {
            b.progress[e] = 0;
            b.reset[e] = 0;
            b.terminate[e] = 0;
        }
        wave_sync();
        const f3 root_pos{s_root[0], s_root[1], s_root[2]};
        const f4 root_rot{s_root[3], s_root[4], s_root[5], s_root[6]};
        if (lane < 12) {
            const int off = DOF_OFFSETS[lane], size = DOF_OFFSETS[lane + 1] - off;
            float d[3] = {s_dof[2 * off], 0.0f, 0.0f}, o6[6];
            if (size == 3) {
                d[1] = s_dof[2 * (off + 1)];
                d[2] = s_dof[2 * (off + 2)];
            }
            joint_obs(d, size, o6);
            float *dst = s_obs + 13 + DOF_OBS_OFFSETS[lane];
            dst[0] = o6[0];
            if (size == 3) { dst[1] = o6[1]; dst[2] = o6[2]; dst[3] = o6[3]; dst[4] = o6[4]; dst[5] = o6[5]; }
        } else if (lane == 12) {
            f4 heading_rot;
            float c[13];
            root_obs(root_pos, root_rot, f3{s_root[7], s_root[8], s_root[9]}, f3{s_root[10], s_root[11], s_root[12]},
                     p.local_root_obs != 0, heading_rot, c);
#pragma unroll
            for (int k = 0; k < 13; ++k) s_obs[k] = c[k];
        } else if (lane == 13) {
            const f4 heading_rot = calc_heading_quat_inv(root_rot);
#pragma unroll
            for (int k = 0; k < NK; ++k) {
                const f3 l = quat_rotate<false>(heading_rot, f3{s_key[3 * k] - root_pos.x, s_key[3 * k + 1] - root_pos.y,
                                                                s_key[3 * k + 2] - root_pos.z});
                s_obs[93 + 3 * k] = l.x; s_obs[94 + 3 * k] = l.y; s_obs[95 + 3 * k] = l.z;
            }
        } else if (lane < 14 + ND) {
            s_obs[65 + (lane - 14)] = s_dof[2 * (lane - 14) + 1];
        }
        wave_sync();
        for (int c = lane; c < NO; c += WAVE) {
            const float v = s_obs[c];
            b.obs[e * NO + c] = v;
            for (int k = 0; k < K; ++k) b.amp_obs[(e * K + k) * NO + c] = v;
        }
        wave_sync();
    }
}

// MotionLib.sample_motions (motion_lib.py:61-65: np.random.choice(m, p=weights) = first index whose
// cumulative weight exceeds a uniform draw) and sample_time (:67-78: uniform phase * motion length) for
// the environments whose reset flag is set, with the draws of the library's own Philox stream.
__global__ __launch_bounds__(256) void motion_sample_flagged_kernel(const igx_motion_tables t, const igx_motion_sampler s,
                                                                    const int64_t *__restrict__ reset, int64_t n,
                                                                    int64_t *__restrict__ env_ids,
                                                                    int64_t *__restrict__ motion_ids,
                                                                    double *__restrict__ motion_times,
                                                                    int64_t *__restrict__ default_env_ids)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    int64_t id = -1, env = -1, denv = -1;
    double time = 0.0;
    if (!reset || reset[e] != 0) {
        float u[4];
        philox_uniform4(s.seed, rng_offset_of(s.rng_offset, s.rng_offset_dev), e + s.env_offset, 0, u);
        if (u[2] < s.ref_init_prob) {
            draw_motion(t, s, u, id, time);
            env = e;
        } else {
            denv = e;
        }
    }
    if (env_ids) env_ids[e] = env;
    motion_ids[e] = id;
    motion_times[e] = time;
    if (default_env_ids) default_env_ids[e] = denv;
}

static int check_tables(const igx_motion_tables *t, const char *who)
{
    if (!t) return fail(IGX_E_NULL, "%s: tables struct is NULL", who);
    if (t->num_motions <= 0) return fail(IGX_E_PARAM, "%s: no motions loaded", who);
    if (!t->motion_length || !t->motion_dt || !t->motion_num_frames || !t->frame_offset || !t->root_pos ||
        !t->root_rot || !t->local_rot || !t->root_vel || !t->root_ang_vel || !t->key_pos || !t->dof_vel)
        return fail(IGX_E_NULL, "%s: a motion table pointer is NULL", who);
    return IGX_OK;
}

static int check_amp(const igx_amp_params *p, const igx_amp_buffers *b, int64_t n, const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: count %lld out of range", who, (long long)n);
    if (p->num_amp_obs_steps < 2) return fail(IGX_E_PARAM, "%s: num_amp_obs_steps must be >= 2", who);
    if (p->num_contact_bodies < 0 || p->num_contact_bodies > 4) return fail(IGX_E_PARAM, "%s: num_contact_bodies", who);
    for (int k = 0; k < 4; ++k)
        if (p->key_body_ids[k] < 0 || p->key_body_ids[k] >= NB) return fail(IGX_E_PARAM, "%s: key_body_ids[%d]", who, k);
    if (n == 0) return IGX_OK;
    if (!b->root_states || !b->dof_state || !b->rigid_body_state || !b->obs || !b->amp_obs || !b->reset ||
        !b->terminate || !b->progress)
        return fail(IGX_E_NULL, "%s: a required buffer is NULL", who);
    if (!aligned16(b->root_states) || !aligned16(b->dof_state))
        return fail(IGX_E_ALIGN, "%s: root_states / dof_state must be 16-byte aligned", who);
    if (b->obs_clamped) {
        if (!(p->clip_obs >= 0.0f)) return fail(IGX_E_PARAM, "%s: clip_obs must be >= 0 when obs_clamped is given", who);
        if (b->obs_clamped == b->obs) return fail(IGX_E_PARAM, "%s: obs_clamped must not alias obs", who);
    }
    return IGX_OK;
}

}  // namespace igx

using namespace igx;

extern "C" int igx_motion_state(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                                const igx_motion_state_out *out, igx_stream_t stream)
{
    int rc = check_tables(t, "igx_motion_state");
    if (rc != IGX_OK) return rc;
    if (n < 0 || n > ((int64_t)1 << 32)) return fail(IGX_E_SIZE, "igx_motion_state: n out of range");
    if (n == 0) return IGX_OK;
    if (!ids || !times || !out) return fail(IGX_E_NULL, "igx_motion_state: NULL argument");
    const int64_t blocks = (n * 16 + 255) / 256;
    hipLaunchKernelGGL((motion_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       ids, times, n, *out, 0, nullptr, nullptr, (int64_t)0, (int64_t)0);
    return launch_status("igx_motion_state");
}

extern "C" int igx_motion_amp_obs(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                                  int32_t local_root_obs, float *out, const int64_t *out_rows, int64_t out_row_stride,
                                  int64_t out_col_offset, igx_stream_t stream)
{
    int rc = check_tables(t, "igx_motion_amp_obs");
    if (rc != IGX_OK) return rc;
    if (n < 0 || n > ((int64_t)1 << 32)) return fail(IGX_E_SIZE, "igx_motion_amp_obs: n out of range");
    if (n == 0) return IGX_OK;
    if (!ids || !times || !out) return fail(IGX_E_NULL, "igx_motion_amp_obs: NULL argument");
    if (out_row_stride < NO || out_col_offset < 0) return fail(IGX_E_PARAM, "igx_motion_amp_obs: bad output stride / offset");
    igx_motion_state_out none = {};
    const int64_t blocks = (n * 16 + 255) / 256;
    hipLaunchKernelGGL((motion_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       ids, times, n, none, (int)local_root_obs, out, out_rows, out_row_stride, out_col_offset);
    return launch_status("igx_motion_amp_obs");
}

extern "C" int igx_motion_amp_obs_steps(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                                        int32_t num_steps, double step_dt, int32_t local_root_obs, float *out,
                                        igx_stream_t stream)
{
    int rc = check_tables(t, "igx_motion_amp_obs_steps");
    if (rc != IGX_OK) return rc;
    if (num_steps < 1 || num_steps > 64) return fail(IGX_E_SIZE, "igx_motion_amp_obs_steps: num_steps %d", num_steps);
    if (n < 0 || n > ((int64_t)1 << 32) / num_steps) return fail(IGX_E_SIZE, "igx_motion_amp_obs_steps: n out of range");
    if (n == 0) return IGX_OK;
    if (!ids || !times || !out) return fail(IGX_E_NULL, "igx_motion_amp_obs_steps: NULL argument");
    igx_motion_state_out none = {};
    const int64_t rows = n * num_steps;
    const int64_t blocks = (rows * 16 + 255) / 256;
    hipLaunchKernelGGL((motion_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       ids, times, rows, none, (int)local_root_obs, out, nullptr, (int64_t)NO, (int64_t)0,
                       StepTiling{(int)num_steps, 0, (int)num_steps, step_dt, nullptr});
    return launch_status("igx_motion_amp_obs_steps");
}

extern "C" int igx_motion_amp_obs_history(const igx_motion_tables *t, const int64_t *ids, const double *times,
                                          const int64_t *env_ids, int64_t n, int32_t num_steps, int32_t first_step,
                                          double step_dt, int32_t local_root_obs, float *amp_obs, igx_stream_t stream)
{
    const char *who = "igx_motion_amp_obs_history";
    int rc = check_tables(t, who);
    if (rc != IGX_OK) return rc;
    if (num_steps < 1 || num_steps > 64 || first_step < 0 || first_step >= num_steps)
        return fail(IGX_E_SIZE, "%s: num_steps %d, first_step %d", who, num_steps, first_step);
    if (n < 0 || n > ((int64_t)1 << 32) / num_steps) return fail(IGX_E_SIZE, "%s: n out of range", who);
    if (n == 0) return IGX_OK;
    if (!ids || !times || !amp_obs) return fail(IGX_E_NULL, "%s: NULL argument", who);
    igx_motion_state_out none = {};
    const int per = num_steps - first_step;
    const int64_t rows = n * per;
    const int64_t blocks = (rows * 16 + 255) / 256;
    hipLaunchKernelGGL((motion_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       ids, times, rows, none, (int)local_root_obs, amp_obs, nullptr, (int64_t)NO, (int64_t)0,
                       StepTiling{per, (int)first_step, (int)num_steps, step_dt, env_ids});
    return launch_status(who);
}

extern "C" int igx_motion_fetch_amp_obs(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int32_t num_steps,
                                        double step_dt, int32_t local_root_obs, float *out, int64_t *motion_ids,
                                        double *motion_times, igx_stream_t stream)
{
    const char *who = "igx_motion_fetch_amp_obs";
    int rc = check_tables(t, who);
    if (rc != IGX_OK) return rc;
    if (!s) return fail(IGX_E_NULL, "%s: sampler struct is NULL", who);
    if (num_steps < 1 || num_steps > 64) return fail(IGX_E_SIZE, "%s: num_steps %d", who, num_steps);
    if (n < 0 || n > ((int64_t)1 << 32) / num_steps) return fail(IGX_E_SIZE, "%s: n out of range", who);
    if (n == 0) return IGX_OK;
    if (!s->cdf || !out) return fail(IGX_E_NULL, "%s: NULL argument", who);
    igx_motion_state_out none = {};
    const int64_t rows = n * num_steps;
    const int64_t blocks = (rows * 16 + 255) / 256;
    hipLaunchKernelGGL((motion_kernel<true, true>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *t,
                       nullptr, nullptr, rows, none, (int)local_root_obs, out, nullptr, (int64_t)NO, (int64_t)0,
                       StepTiling{(int)num_steps, 0, (int)num_steps, step_dt, nullptr}, *s, motion_ids, motion_times);
    return launch_status(who);
}

extern "C" int igx_motion_sample(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int64_t *motion_ids,
                                 double *motion_times, igx_stream_t stream)
{
    int rc = check_tables(t, "igx_motion_sample");
    if (rc != IGX_OK) return rc;
    if (!s) return fail(IGX_E_NULL, "igx_motion_sample: sampler struct is NULL");
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "igx_motion_sample: n out of range");
    if (n == 0) return IGX_OK;
    if (!s->cdf || !motion_ids || !motion_times) return fail(IGX_E_NULL, "igx_motion_sample: NULL argument");
    igx_motion_sampler all = *s;
    all.ref_init_prob = 2.0f;  // every draw is a motion (u2 < 2 always)
    hipLaunchKernelGGL(motion_sample_flagged_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *t, all, nullptr, n, nullptr, motion_ids, motion_times, nullptr);
    return launch_status("igx_motion_sample");
}

extern "C" int igx_amp_observations(const float *root_states, const float *dof_pos, const float *dof_vel,
                                    const float *key_body_pos, int32_t local_root_obs, float *obs, int64_t n,
                                    igx_stream_t stream)
{
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "igx_amp_observations: n out of range");
    if (n == 0) return IGX_OK;
    if (!root_states || !dof_pos || !dof_vel || !key_body_pos || !obs) return fail(IGX_E_NULL, "igx_amp_observations: NULL argument");
    hipLaunchKernelGGL(amp_obs_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       root_states, dof_pos, dof_vel, key_body_pos, (int)local_root_obs, obs, n);
    return launch_status("igx_amp_observations");
}

extern "C" int igx_humanoid_amp_step(const igx_amp_params *p, const igx_amp_buffers *b, int64_t n, igx_stream_t stream)
{
    int rc = check_amp(p, b, n, "igx_humanoid_amp_step");
    if (rc != IGX_OK || n == 0) return rc;
    if (!b->contact_forces || !b->rew) return fail(IGX_E_NULL, "igx_humanoid_amp_step: contact_forces / rew is NULL");
    hipStream_t s = static_cast<hipStream_t>(stream);
    static const int reserve = tuning_int("IGX_AMP_LDS_RESERVE", 0);  // unused LDS: caps the waves per CU
    const dim3 g2((unsigned)((n + PAIR_T - 1) / PAIR_T));
    switch (p->num_amp_obs_steps) {
    case 2: hipLaunchKernelGGL(amp_step_pair_kernel<2>, g2, dim3(WAVE), (size_t)reserve, s, *p, *b, n); break;
    case 3: hipLaunchKernelGGL(amp_step_pair_kernel<3>, g2, dim3(WAVE), (size_t)reserve, s, *p, *b, n); break;
    case 4: hipLaunchKernelGGL(amp_step_pair_kernel<4>, g2, dim3(WAVE), (size_t)reserve, s, *p, *b, n); break;
    default: hipLaunchKernelGGL(amp_step_pair_kernel<0>, g2, dim3(WAVE), (size_t)reserve, s, *p, *b, n); break;
    }
    return launch_status("igx_humanoid_amp_step");
}

extern "C" int igx_humanoid_amp_reset(const igx_amp_params *p, const igx_amp_buffers *b, const int64_t *env_ids, int64_t m,
                                      const igx_motion_state_out *state, igx_stream_t stream)
{
    int rc = check_amp(p, b, m, "igx_humanoid_amp_reset");
    if (rc != IGX_OK || m == 0) return rc;
    if (!env_ids || !state) return fail(IGX_E_NULL, "igx_humanoid_amp_reset: env_ids / state is NULL");
    const int per_block = m >= 32768 ? 64 : 8;
    hipLaunchKernelGGL(amp_reset_kernel, dim3((unsigned)((m + per_block - 1) / per_block)), dim3(64 * RESET_WAVES), 0,
                       static_cast<hipStream_t>(stream), *p, *b, env_ids, m, *state, per_block);
    return launch_status("igx_humanoid_amp_reset");
}

extern "C" int igx_motion_sample_flagged(const igx_motion_tables *t, const igx_motion_sampler *s, const int64_t *reset,
                                         int64_t n, int64_t *env_ids, int64_t *motion_ids, double *motion_times,
                                         int64_t *default_env_ids, igx_stream_t stream)
{
    int rc = check_tables(t, "igx_motion_sample_flagged");
    if (rc != IGX_OK) return rc;
    if (!s) return fail(IGX_E_NULL, "igx_motion_sample_flagged: sampler struct is NULL");
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "igx_motion_sample_flagged: n out of range");
    if (n == 0) return IGX_OK;
    if (!s->cdf || !reset || !env_ids || !motion_ids || !motion_times)
        return fail(IGX_E_NULL, "igx_motion_sample_flagged: NULL argument");
    if (!(s->ref_init_prob >= 0.0f)) return fail(IGX_E_PARAM, "igx_motion_sample_flagged: ref_init_prob must be >= 0");
    if (s->ref_init_prob < 1.0f && !default_env_ids)
        return fail(IGX_E_NULL, "igx_motion_sample_flagged: default_env_ids is NULL with ref_init_prob < 1");
    hipLaunchKernelGGL(motion_sample_flagged_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *t, *s, reset, n, env_ids, motion_ids, motion_times, default_env_ids);
    return launch_status("igx_motion_sample_flagged");
}
