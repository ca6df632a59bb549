// Fused post_physics_step for the Ant and Humanoid locomotion tasks.
//
// Work decomposition: one wavefront (a 64-thread workgroup) owns one tile of
// 64 consecutive environments; the grid has one workgroup per tile (measured
// on MI355X: many small one-shot workgroups stream this multi-array pattern
// faster than persistent waves with a register prefetch).  Per tile:
//   load     the tile's AoS inputs arrive in registers through coalesced
//            16-byte non-temporal loads (each lane: consecutive 16-byte chunks
//            of the tile's contiguous slice of every input array);
//   root     the 13-float root-state rows are transposed through LDS so that
//            lane i holds environment i's row in registers;
//   stage    pass-through / scaled columns (dof position and velocity, dof
//            force, sensors, actions) go straight from the load registers to
//            their final column of the LDS image of the 64 observation rows;
//   reset    environments whose reset flag is set are re-initialised one at a
//            time by the whole wave (lane j draws and writes joint j);
//   per-env  lane i: quaternion math in registers, the 12 computed columns
//            into its LDS row, reward from its own row, per-env scalars
//            straight to global memory;
//   drain    the finished rows (then the up / heading vectors) leave LDS as
//            whole contiguous lines with 16-byte non-temporal stores.
// Every input byte is read once and every output byte written once; observation
// rows are only ever written as full contiguous lines (partial-row stores run
// 2-3x slower on this memory system).  Lanes of the wave exchange data through
// LDS without a workgroup barrier: a wave's LDS operations execute in order.
//
// Reference: tasks/ant.py:376-386, :339-366, :463-497, :414-460 and
// tasks/humanoid.py:398-407, :360-390, :489-524, :434-486
// (paths relative to IsaacGymEnvs/isaacgymenvs/).
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

// Environments per tile (one wave per tile).  A 64-row Humanoid tile is 27.6 KB of LDS
// (5 waves per CU); with 32-row tiles the upper half of the wave idles during the
// per-environment math, which is hidden behind memory anyway, and twice as many waves
// stream (measured 199 -> 189 us per 2^20 envs).  Ant is insensitive (106 us either way).
template <int D, int S, bool HUM, int T>
struct LocoLayout {
    static constexpr int TILE = T;
    static_assert(T == 32 || T == 64, "tile rows");
    static constexpr int C_POS = 12;
    static constexpr int C_VEL = 12 + D;
    static constexpr int C_FORCE = 12 + 2 * D;  // Humanoid only
    static constexpr int C_SENS = HUM ? 12 + 3 * D : 12 + 2 * D;
    static constexpr int C_ACT = C_SENS + S;
    static constexpr int NOBS = C_ACT + D;
    // 16-byte chunks per full tile of each input array, and load rounds per lane
    static constexpr int ROOT4 = TILE * 13 / 4;
    static constexpr int SENS4 = TILE * S / 4;
    static constexpr int ACT4 = TILE * D / 4;
    static constexpr int DOF4 = TILE * D * 2 / 4;
    static constexpr int IT_ROOT = (ROOT4 + WAVE - 1) / WAVE;
    static constexpr int IT_SENS = (SENS4 + WAVE - 1) / WAVE;
    static constexpr int IT_ACT = (ACT4 + WAVE - 1) / WAVE;
    static constexpr int IT_DOF = (DOF4 + WAVE - 1) / WAVE;
};

// reset_idx for the environments of this tile whose reset flag is set, with the
// wave's lanes spread over (environment, joint) pairs: a lane draws its joint's
// position and velocity uniforms, writes the (pos, vel) pair to dof_state and
// the two derived observation columns to the LDS row.
template <int D, int S, bool HUM, int T>
__device__ __forceinline__ void reset_joints(const igx_loco_params &p, const igx_loco_buffers &b, int64_t base,
                                             int lane, uint64_t reset_mask, float *s_tile, const float *s_lim)
{
    using L = LocoLayout<D, S, HUM, T>;
    // SLOTS flagged environments per round, lane = (slot, joint): when every environment of the
    // tile resets (the first step, synchronised time-outs) the tile needs T / SLOTS rounds, not T
    constexpr int SLOTS = WAVE / D;
    const int slot = lane / D;
    while (reset_mask) {
        uint64_t m = reset_mask;
        for (int s = 0; s < slot; ++s) m &= m - 1;  // drop the `slot` lowest flagged environments
        const bool have = slot < SLOTS && m != 0;
        const int src = have ? __builtin_ctzll(m) : 0;
#pragma unroll
        for (int s = 0; s < SLOTS; ++s) reset_mask &= reset_mask - 1;  // 0 stays 0
        const int64_t e = base + src;
        if (have) {
            const int j = lane - slot * D;
            float up, uv;  // uniforms j (position) and D + j (velocity) of this environment
            if (b.rand) {
                up = b.rand[e * (2 * D) + j];
                uv = b.rand[e * (2 * D) + D + j];
            } else {
                float q[4];
                philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, j >> 2, q);
                const int a = j & 3;
                up = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
                philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, (D + j) >> 2, q);
                const int c = (D + j) & 3;
                uv = c == 0 ? q[0] : c == 1 ? q[1] : c == 2 ? q[2] : q[3];
            }
            const float dp = 0.4f * up + (-0.2f);  // torch_rand_float(-0.2, 0.2)
            const float dv = 0.2f * uv + (-0.1f);  // torch_rand_float(-0.1, 0.1)
            const float lo = s_lim[j], hi = s_lim[D + j];
            const float np_ = tensor_clamp(b.initial_dof_pos[e * D + j] + dp, lo, hi);
            reinterpret_cast<v2f *>(b.dof_state + e * (2 * D))[j] = v2f{np_, dv};
            float *row = s_tile + src * L::NOBS;
            row[L::C_POS + j] = unscale(np_, lo, hi);
            row[L::C_VEL + j] = dv * p.dof_vel_scale;
        }
    }
}

// What lane i computes for environment e, in two parts.  Part A needs only the root row and the
// per-environment scalars, so it runs while the tile's wide streams (dof state, sensors, actions)
// are still arriving; part B needs the staged observation row.
struct EnvA {
    float c[12];          // observation columns 0..11
    f3 up_vec, heading_vec;
    float pot, prev_pot, pos_z, lin_x;
    int64_t prog, rst;
};

template <bool HUM>
__device__ __forceinline__ void per_env_a(const igx_loco_params &p, const igx_loco_buffers &b, int64_t e,
                                          const float (&rr)[13], int64_t prog, int64_t rst, float pot, EnvA &a)
{
    const f3 pos{rr[0], rr[1], rr[2]};
    const f4 rot{rr[3], rr[4], rr[5], rr[6]};
    const f3 lin{rr[7], rr[8], rr[9]};
    const f3 ang{rr[10], rr[11], rr[12]};

    // a flagged environment (reset_idx) restarts from potentials = -|targets - initial root|/dt, which
    // only the reward reads (as prev_potentials): part B fetches the initial root row, so that no
    // memory access sits between the arrival of the root rows and this math
    prog += 1;
    if (rst != 0) prog = 0;

    // compute_*_observations
    const f3 to_target{p.targets[0] - pos.x, p.targets[1] - pos.y, 0.0f};
    a.prev_pot = pot;
    a.pot = -norm3(to_target) / p.dt;
    const f3 target_dirs = normalize3(to_target);
    const f4 isr{p.inv_start_rot[0], p.inv_start_rot[1], p.inv_start_rot[2], p.inv_start_rot[3]};
    const f4 tq = quat_mul(rot, isr);
    a.up_vec = quat_rotate<false>(tq, f3{p.basis_vec1[0], p.basis_vec1[1], p.basis_vec1[2]});
    a.heading_vec = quat_rotate<false>(tq, f3{p.basis_vec0[0], p.basis_vec0[1], p.basis_vec0[2]});
    const float up_proj = a.up_vec.z;
    const float heading_proj = dot3(a.heading_vec, target_dirs);
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
    a.c[0] = pos.z; a.c[1] = vel_loc.x; a.c[2] = vel_loc.y; a.c[3] = vel_loc.z;
    a.c[4] = angvel_loc.x; a.c[5] = angvel_loc.y; a.c[6] = angvel_loc.z; a.c[7] = yaw;
    a.c[8] = roll; a.c[9] = angle_to_target; a.c[10] = up_proj; a.c[11] = heading_proj;
    a.pos_z = pos.z;
    a.lin_x = lin.x;
    a.prog = prog;
    a.rst = rst;
}

template <int D, int S, bool HUM, int T, bool IMIT>
__device__ __forceinline__ void per_env_b(const igx_loco_params &p, const igx_loco_buffers &b, const igx_imitate_args &im, int64_t e,
                                          const EnvA &a, int64_t rnd, float *row, const float *s_lim)
{
    using L = LocoLayout<D, S, HUM, T>;
    constexpr int NOBS = L::NOBS;
    int64_t prog = a.prog, rst = a.rst;
    float prev_pot = a.prev_pot;
    const float pot = a.pot;
    if (rst != 0) {  // reset_idx, per-env scalars (the joints were re-drawn by reset_joints)
        const float *ir = b.initial_root_states + e * 13;
        const f3 tt{p.targets[0] - ir[0], p.targets[1] - ir[1], 0.0f};
        prev_pot = -norm3(tt) / p.dt;
        rst = 0;
    }
    const float up_proj = a.c[10], heading_proj = a.c[11];
    if (b.randomize) st_narrow(b.randomize + e, rnd + 1);
    *reinterpret_cast<v4f *>(row + 0) = v4f{a.c[0], a.c[1], a.c[2], a.c[3]};
    *reinterpret_cast<v4f *>(row + 4) = v4f{a.c[4], a.c[5], a.c[6], a.c[7]};
    *reinterpret_cast<v4f *>(row + 8) = v4f{a.c[8], a.c[9], a.c[10], a.c[11]};

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
    if constexpr (IMIT) {
        // tasks/ant_imitate.py:508-519: fp64 throughout (the motion table is fp64 and promotes)
        static_assert(D == 8 && !HUM, "AntImitate is the 8-dof Ant");
        int64_t idx = im.target_pose_idx[e] + 1;  // :408
        im.target_pose_idx[e] = idx;
        idx = idx < 0 ? 0 : (idx >= im.num_frames ? im.num_frames - 1 : idx);
        const double *m = im.motion + idx * D;
        double diff[D];
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const double lo = (double)s_lim[j], hi = (double)s_lim[D + j];
            const double span = (double)(s_lim[D + j] - s_lim[j]);  // upper - lower is an fp32 subtraction
            const double scaled = (2.0 * m[j] - hi - lo) / span;
            diff[j] = fabs((double)dpos[j] - scaled);
        }
        // torch.sum over 8 doubles: 4 vector lanes over the two blocks of 4, lanes left to right
        const double pose_diff = (((diff[0] + diff[4]) + (diff[1] + diff[5])) + (diff[2] + diff[6])) + (diff[3] + diff[7]);
        im.pose_diff[e] = pose_diff;
        total = (float)exp(-pose_diff);
    } else if constexpr (!HUM) {
#pragma unroll
        for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * dvel[j]);
        const float electricity_cost = row_sum<D>(tmp);
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) cnt += (dpos[j] > 0.99f) ? 1 : 0;
        total = progress_reward + 0.5f + up_reward + heading_reward - p.actions_cost_scale * actions_cost -
                p.energy_cost_scale * electricity_cost - (float)cnt * p.joints_at_limit_cost_scale;
    } else {
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float ab = fabsf(dpos[j]);
            const float scaled = p.joints_at_limit_cost_scale * (ab - 0.98f) / 0.02f;
            tmp[j] = ((ab > 0.98f) ? 1.0f : 0.0f) * scaled * s_lim[2 * D + j];
        }
        const float dof_at_limit_cost = row_sum<D>(tmp);
#pragma unroll
        for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * dvel[j]) * s_lim[2 * D + j];
        const float electricity_cost = row_sum<D>(tmp);
        total = progress_reward + 2.0f + up_reward + heading_reward - p.actions_cost_scale * actions_cost -
                p.energy_cost_scale * electricity_cost - dof_at_limit_cost;
    }
    if (a.pos_z < p.termination_height) {
        total = IMIT ? (float)im.death_cost : p.death_cost;
        rst = 1;
    }
    if ((float)prog >= p.max_episode_length_m1) rst = 1;

    st_narrow(b.potentials + e, pot);
    st_narrow(b.prev_potentials + e, prev_pot);
    st_narrow(b.rew + e, total);
    st_narrow(b.reset + e, rst);
    st_narrow(b.progress + e, prog);
    if (b.true_objective) st_narrow(b.true_objective + e, a.lin_x);
    if (!IMIT && b.time_outs) b.time_outs[e] = ((float)prog >= p.max_episode_length_m1 && rst != 0) ? 1 : 0;
}

// Everything a tile reads, as it sits in registers after the load phase.
template <int D, int S, bool HUM, int T>
struct TileIn {
    using L = LocoLayout<D, S, HUM, T>;
    int64_t prog, rst, rnd;
    float pot;
    v4f r_root[L::IT_ROOT], r_dof[L::IT_DOF], r_sens[L::IT_SENS], r_act[L::IT_ACT], r_force[HUM ? L::IT_ACT : 1];
};

template <int D, int S, bool HUM, int T, bool FULL>
__device__ __forceinline__ void load_tile(const igx_loco_buffers &b, int64_t base, int nvalid, int lane, TileIn<D, S, HUM, T> &t)
{
    using L = LocoLayout<D, S, HUM, T>;
    // branch-free so that every load of the tile is issued before the first wait: lanes
    // without an environment re-read the tile's first one, a missing randomize buffer
    // re-reads progress (both results are never used)
    // order: what the first part of the per-environment math needs (root rows, flags, potentials)
    // comes first - loads return in order, and that math runs while the rest is still arriving
    const int64_t e = base + (lane < nvalid ? lane : 0);
    if constexpr (FULL) load_chunks<L::ROOT4>(b.root_states + base * 13, t.r_root, lane);
    t.prog = ld_narrow(b.progress + e);
    t.rst = ld_narrow(b.reset + e);
    t.pot = ld_narrow(b.potentials + e);
    // the instruction scheduler may not move the wide loads below ahead of these: the hardware
    // returns loads in issue order
    __builtin_amdgcn_sched_barrier(0);
    t.rnd = ld_narrow((b.randomize ? b.randomize : b.progress) + e);
    if constexpr (FULL) {
        load_chunks<L::DOF4>(b.dof_state + base * (2 * D), t.r_dof, lane);
        load_chunks<L::ACT4>(b.actions + base * D, t.r_act, lane);
        __builtin_amdgcn_sched_barrier(0);  // what the reward reads, before the pure pass-through columns
        if constexpr (HUM) load_chunks<L::ACT4>(b.dof_force + base * D, t.r_force, lane);
        load_chunks<L::SENS4>(b.sensors + base * S, t.r_sens, lane);
    }
}

// One tile, start to finish.  FULL (all T environments present) and the one partial tile at the end
// of a batch are separate instantiations, so that the full tile is straight-line code: every load is
// unconditional, and the waits the compiler places are exactly "the loads this instruction needs"
// (a load issued under a condition would, where the paths merge, turn every later wait into "nearly all").
template <int D, int S, bool HUM, int T, bool IMIT, bool FULL>
__device__ __forceinline__ void loco_tile(const igx_loco_params &p, const igx_loco_buffers &b, const igx_imitate_args &im,
                                          int64_t base, int nvalid_in, int lane, float lim, float effort, float *s_tile,
                                          float *s_lim)
{
    using L = LocoLayout<D, S, HUM, T>;
    constexpr int NOBS = L::NOBS;
    constexpr int TILE = L::TILE;
    const int nvalid = FULL ? TILE : nvalid_in;
    const bool live = lane < nvalid;
    const int64_t e = base + lane;

    // ---- load: everything the tile reads, issued back to back
    TileIn<D, S, HUM, T> in;
    load_tile<D, S, HUM, T, FULL>(b, base, nvalid, lane, in);

    if (lane < 2 * D) s_lim[lane] = lim;
    if (HUM && lane < D) s_lim[2 * D + lane] = effort / p.max_motor_effort;
    const int64_t prog = in.prog, rst = in.rst, rnd = in.rnd;
    const float pot = in.pot;

    // ---- root: transpose the 13-float rows through the (still empty) tile
    if constexpr (FULL) {
#pragma unroll
        for (int it = 0; it < L::IT_ROOT; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= L::ROOT4 || k < L::ROOT4) *reinterpret_cast<v4f *>(s_tile + 4 * k) = in.r_root[it];
        }
    } else {
        for (int i = lane; i < nvalid * 13; i += WAVE) s_tile[i] = b.root_states[base * 13 + i];
    }
    wave_sync();
    float rr[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) rr[i] = s_tile[(lane & (TILE - 1)) * 13 + i];  // lanes >= nvalid read stale LDS, never used
    wave_sync();

    // ---- per-env math, part A: everything that needs only the root row, while the wide
    // streams of the tile are still in flight
    EnvA ea;
    if (live) per_env_a<HUM>(p, b, e, rr, prog, rst, pot, ea);

    // ---- stage, first half: the columns the reward reads (joint positions / velocities, actions)
    if constexpr (FULL) {
        stage_dof_state<D, NOBS, L::C_POS, L::C_VEL, TILE>(in.r_dof, s_tile, s_lim, p.dof_vel_scale, lane);
        stage_rows<D, L::C_ACT, NOBS, false, TILE>(in.r_act, s_tile, 1.0f, lane);
    } else {  // plain scalar staging
        for (int i = lane; i < nvalid * D; i += WAVE) {
            const int env = i / D, j = i - env * D;
            const v2f pv = reinterpret_cast<const v2f *>(b.dof_state)[base * D + i];
            s_tile[env * NOBS + L::C_POS + j] = unscale(pv.x, s_lim[j], s_lim[D + j]);
            s_tile[env * NOBS + L::C_VEL + j] = pv.y * p.dof_vel_scale;
            s_tile[env * NOBS + L::C_ACT + j] = b.actions[base * D + i];
            if constexpr (HUM) s_tile[env * NOBS + L::C_FORCE + j] = b.dof_force[base * D + i] * p.contact_force_scale;
        }
        for (int i = lane; i < nvalid * S; i += WAVE) {
            const int env = i / S, c = i - env * S;
            s_tile[env * NOBS + L::C_SENS + c] = b.sensors[base * S + i] * p.contact_force_scale;
        }
    }
    wave_sync();

    // ---- reset: re-draw the joints of flagged environments (rare)
    const uint64_t reset_mask = __ballot(live && rst != 0);
    if (reset_mask) {
        reset_joints<D, S, HUM, T>(p, b, base, lane, reset_mask, s_tile, s_lim);
        wave_sync();
    }

    // ---- per-env math, part B: the computed columns into the row, reward from the row
    if (live) per_env_b<D, S, HUM, T, IMIT>(p, b, im, e, ea, rnd, s_tile + lane * NOBS, s_lim);
    const f3 up_vec = ea.up_vec, heading_vec = ea.heading_vec;  // lanes without an environment: never stored

    // ---- stage, second half: the pure pass-through columns (sensors, joint forces), requested
    // last, so that the reward above ran while they were still arriving
    if constexpr (FULL) {
        stage_rows<S, L::C_SENS, NOBS, true, TILE>(in.r_sens, s_tile, p.contact_force_scale, lane);
        if constexpr (HUM) stage_rows<D, L::C_FORCE, NOBS, true, TILE>(in.r_force, s_tile, p.contact_force_scale, lane);
    }
    wave_sync();

    // ---- drain: finished rows out as whole lines
    drain_rows<NOBS, TILE>(b.obs + base * NOBS, s_tile, nvalid, lane);
    if (b.obs_clamped)  // the copy VecTask.step hands to the learner
        drain_rows_clamped<NOBS, TILE>(b.obs_clamped + base * NOBS, s_tile, nvalid, lane, p.clip_obs);
    wave_sync();
    // up / heading vectors: transpose through the tile's first 384 floats
    constexpr int V3 = 3 * TILE;  // floats per vector array of a tile
    if (lane < TILE) {
        s_tile[lane * 3 + 0] = up_vec.x; s_tile[lane * 3 + 1] = up_vec.y; s_tile[lane * 3 + 2] = up_vec.z;
        s_tile[V3 + lane * 3 + 0] = heading_vec.x;
        s_tile[V3 + lane * 3 + 1] = heading_vec.y;
        s_tile[V3 + lane * 3 + 2] = heading_vec.z;
    }
    wave_sync();
    if constexpr (FULL) {
        if (lane < V3 / 4) {
            st_stream(reinterpret_cast<v4f *>(b.up_vec + base * 3) + lane, *reinterpret_cast<const v4f *>(s_tile + 4 * lane));
            st_stream(reinterpret_cast<v4f *>(b.heading_vec + base * 3) + lane,
                      *reinterpret_cast<const v4f *>(s_tile + V3 + 4 * lane));
        }
    } else {
        for (int i = lane; i < nvalid * 3; i += WAVE) {
            b.up_vec[base * 3 + i] = s_tile[i];
            b.heading_vec[base * 3 + i] = s_tile[V3 + i];
        }
    }
}

template <int D, int S, bool HUM, int T, bool IMIT>
__global__ __launch_bounds__(WAVE) void loco_step_kernel(const igx_loco_params p, const igx_loco_buffers b, int64_t n,
                                                         const igx_imitate_args im)
{
    using L = LocoLayout<D, S, HUM, T>;
    constexpr int NOBS = L::NOBS;
    constexpr int TILE = L::TILE;
    static_assert(NOBS % 4 == 0, "observation rows must be a multiple of 16 bytes");
    static_assert(NOBS >= 13, "the root rows are transposed through the observation tile");
    __shared__ __align__(16) float s_tile[TILE * NOBS];
    // per-dof constants: [0,D) lower limit, [D,2D) upper limit, [2D,3D) motor effort ratio
    __shared__ __align__(16) float s_lim[(3 * D + 4) & ~3];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * TILE;  // one tile per wave: several tiles per wave with
                                                      // next-tile prefetch measured 14-18 us slower

    // per-dof constants picked out of the (scalar) kernel arguments by a select chain: a
    // lane-indexed read of the argument block would be a vector memory load that the
    // tile's streaming loads then have to queue behind
    float lim = 0.0f, effort = 1.0f;
#pragma unroll
    for (int j = 0; j < D; ++j) {
        lim = lane == j ? p.dof_limits_lower[j] : lim;
        lim = lane == D + j ? p.dof_limits_upper[j] : lim;
        if (HUM) effort = lane == j ? p.motor_efforts[j] : effort;
    }
    if (n - base >= TILE)
        loco_tile<D, S, HUM, T, IMIT, true>(p, b, im, base, TILE, lane, lim, effort, s_tile, s_lim);
    else
        loco_tile<D, S, HUM, T, IMIT, false>(p, b, im, base, (int)(n - base), lane, lim, effort, s_tile, s_lim);
}

// reset_idx on an explicit id list: lane (i, j) re-draws joint j of the i-th
// listed environment; the j == 0 lane also writes the per-env scalars.
template <int D>
__global__ __launch_bounds__(256) void loco_reset_kernel(const igx_loco_params p, const igx_loco_buffers b,
                                                         const int64_t *__restrict__ env_ids, int64_t m)
{
    const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = gid / D;
    const int j = (int)(gid - i * D);
    if (i >= m) return;
    const int64_t e = env_ids[i];
    float up, uv;
    if (b.rand) {
        up = b.rand[e * (2 * D) + j];
        uv = b.rand[e * (2 * D) + D + j];
    } else {
        float q[4];
        philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, j >> 2, q);
        const int a = j & 3;
        up = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
        philox_uniform4(p.seed, rng_offset_of(p.rng_offset, p.rng_offset_dev), e + p.env_offset, (D + j) >> 2, q);
        const int c = (D + j) & 3;
        uv = c == 0 ? q[0] : c == 1 ? q[1] : c == 2 ? q[2] : q[3];
    }
    const float dp = 0.4f * up + (-0.2f);
    const float dv = 0.2f * uv + (-0.1f);
    const float np_ = tensor_clamp(b.initial_dof_pos[e * D + j] + dp, p.dof_limits_lower[j], p.dof_limits_upper[j]);
    reinterpret_cast<v2f *>(b.dof_state + e * (2 * D))[j] = v2f{np_, dv};
    if (j == 0) {
        const float *ir = b.initial_root_states + e * 13;
        const float pot = -norm3(f3{p.targets[0] - ir[0], p.targets[1] - ir[1], 0.0f}) / p.dt;
        b.potentials[e] = pot;
        b.prev_potentials[e] = pot;
        b.progress[e] = 0;
        b.reset[e] = 0;
    }
}

template <int D>
static int launch_loco_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m,
                             igx_stream_t stream, const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (m < 0 || m > ((int64_t)1 << 32)) return fail(IGX_E_SIZE, "%s: m %lld out of range", who, (long long)m);
    if (p->num_dof != D) return fail(IGX_E_PARAM, "%s: expects num_dof=%d, got %d", who, D, p->num_dof);
    if (m == 0) return IGX_OK;
    if (!env_ids || !b->dof_state || !b->initial_dof_pos || !b->initial_root_states || !b->potentials ||
        !b->prev_potentials || !b->progress || !b->reset)
        return fail(IGX_E_NULL, "%s: a required pointer is NULL", who);
    if (!aligned16(b->dof_state)) return fail(IGX_E_ALIGN, "%s: dof_state must be 16-byte aligned", who);
    hipLaunchKernelGGL((loco_reset_kernel<D>), dim3((unsigned)((m * D + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *p, *b, env_ids, m);
    return launch_status(who);
}

static int check_loco(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, int D, int S, bool hum,
                      const char *who)
{
    if (!p || !b) return fail(IGX_E_NULL, "%s: params/buffers struct is NULL", who);
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n);
    if (p->num_dof != D || p->num_sensor_floats != S)
        return fail(IGX_E_PARAM, "%s: expects num_dof=%d num_sensor_floats=%d, got %d/%d", who, D, S, p->num_dof,
                    p->num_sensor_floats);
    if (n == 0) return IGX_OK;  // empty batch: pointers may legitimately be NULL
    const struct { const void *ptr; const char *name; bool vec; } req[] = {
        {b->root_states, "root_states", true}, {b->dof_state, "dof_state", true},
        {b->sensors, "sensors", true}, {b->actions, "actions", true},
        {b->initial_root_states, "initial_root_states", false}, {b->initial_dof_pos, "initial_dof_pos", false},
        {b->obs, "obs", true}, {b->potentials, "potentials", false}, {b->prev_potentials, "prev_potentials", false},
        {b->up_vec, "up_vec", true}, {b->heading_vec, "heading_vec", true}, {b->rew, "rew", false},
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
    if (b->obs_clamped) {
        if (!(p->clip_obs >= 0.0f)) return fail(IGX_E_PARAM, "%s: clip_obs must be >= 0 when obs_clamped is given", who);
        if (b->obs_clamped == b->obs) return fail(IGX_E_PARAM, "%s: obs_clamped must not alias obs", who);
    }
    return IGX_OK;
}

template <int D, int S, bool HUM, int T, bool IMIT = false>
static int launch_loco_tile(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, int lds_reserve,
                            igx_stream_t stream, const char *who, const igx_imitate_args *im = nullptr)
{
    const int64_t tiles = (n + T - 1) / T;
    if (tiles > 0x7fffffffLL) return fail(IGX_E_SIZE, "%s: n_envs too large for one launch", who);
    const int64_t blocks = tiles;
    const igx_imitate_args none = {};
    hipLaunchKernelGGL((loco_step_kernel<D, S, HUM, T, IMIT>), dim3((unsigned)blocks), dim3(WAVE), (size_t)lds_reserve,
                       static_cast<hipStream_t>(stream), *p, *b, n, im ? *im : none);
    return launch_status(who);
}

// Launch shape.  Two measured knobs per task (2^20 envs, MI355X): the rows per tile and a
// reservation of unused LDS that caps the resident waves per CU - this multi-array streaming
// pattern runs fastest with FEWER waves in flight than the LDS tile alone would allow
// (Ant, 64-row tiles: 10 waves/CU 96.2 us, 7 and 6 waves 94.7 us, 5 waves 96.0 us, 4 waves 99.6 us;
// Humanoid is fastest with 32-row tiles and no reservation: 168.9 us against 172.8 us with 64-row tiles).
// IGX_{ANT,HUMANOID}_TILE / _LDS_RESERVE override the defaults for tuning runs.
template <int D, int S, bool HUM>
static int launch_loco(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, igx_stream_t stream,
                       const char *who)
{
    if (n == 0) return IGX_OK;
    static const int tile = tuning_int(HUM ? "IGX_HUMANOID_TILE" : "IGX_ANT_TILE", HUM ? 32 : 64);
    static const int reserve = tuning_int(HUM ? "IGX_HUMANOID_LDS_RESERVE" : "IGX_ANT_LDS_RESERVE", HUM ? 0 : 8000);
    if (reserve < 0 || reserve > 120 * 1024) return fail(IGX_E_PARAM, "%s: LDS reservation %d out of range", who, reserve);
    if (tile == 32) return launch_loco_tile<D, S, HUM, 32>(p, b, n, reserve, stream, who);
    return launch_loco_tile<D, S, HUM, 64>(p, b, n, reserve, stream, who);
}

// tasks/ant_imitate.py:523-555: the decisions that need the global threshold
__global__ __launch_bounds__(256) void imitate_finish_kernel(const igx_loco_buffers b, const igx_imitate_args im, int64_t n,
                                                             float max_episode_length_m1)
{
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const double q = reinterpret_cast<const double *>(im.workspace)[5];  // QuantileState::result
    // Python's max(1.8, quantile): the quantile only if it compares greater - a NaN quantile leaves 1.8
    const double thr = q > im.distance_floor ? q : im.distance_floor;
    const double d = ld_narrow(im.pose_diff + e);
    int64_t idx = im.target_pose_idx[e];
    const bool far = d > thr;
    const bool finished = (far ? 0 : idx) >= im.num_frames - 3;
    if (far) {
        b.reset[e] = 1;
        b.rew[e] = (float)im.death_cost;
        idx = 0;
    }
    if (finished) {
        b.rew[e] = (float)(-3.0 * im.death_cost);
        idx = 0;
    }
    if (far || finished) im.target_pose_idx[e] = idx;
    if (b.time_outs) {  // VecTask.step's time-out flag, on the reset flag as this kernel leaves it
        const int64_t rst = far ? 1 : b.reset[e];
        b.time_outs[e] = ((float)b.progress[e] >= max_episode_length_m1 && rst != 0) ? 1 : 0;
    }
}

static int check_imitate(const igx_imitate_args *im, int64_t n, const char *who)
{
    if (!im) return fail(IGX_E_NULL, "%s: imitate args struct is NULL", who);
    if (im->num_frames < 4) return fail(IGX_E_PARAM, "%s: the motion needs at least 4 frames", who);
    if (!(im->quantile >= 0.0 && im->quantile <= 1.0)) return fail(IGX_E_PARAM, "%s: quantile outside [0, 1]", who);
    if (im->n_total < n) return fail(IGX_E_SIZE, "%s: n_total %lld < n_envs %lld", who, (long long)im->n_total, (long long)n);
    if (n == 0) return IGX_OK;
    if (!im->motion || !im->target_pose_idx || !im->pose_diff || !im->workspace)
        return fail(IGX_E_NULL, "%s: motion / target_pose_idx / pose_diff / workspace is NULL", who);
    if (!aligned16(im->workspace)) return fail(IGX_E_ALIGN, "%s: workspace must be 16-byte aligned", who);
    return IGX_OK;
}

}  // namespace igx

extern "C" int igx_ant_imitate_observe(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                                       int64_t n_envs, igx_stream_t stream)
{
    int rc = igx::check_loco(p, b, n_envs, 8, 24, false, "igx_ant_imitate_observe");
    if (rc == IGX_OK) rc = igx::check_imitate(im, n_envs, "igx_ant_imitate_observe");
    if (rc != IGX_OK || n_envs == 0) return rc;
    static const int reserve = igx::tuning_int("IGX_ANT_LDS_RESERVE", 8000);
    return igx::launch_loco_tile<8, 24, false, 64, true>(p, b, n_envs, reserve, stream, "igx_ant_imitate_observe", im);
}

extern "C" int igx_ant_imitate_finish(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                                      int64_t n_envs, igx_stream_t stream)
{
    if (!p || !b) return igx::fail(IGX_E_NULL, "igx_ant_imitate_finish: params/buffers struct is NULL");
    int rc = igx::check_imitate(im, n_envs, "igx_ant_imitate_finish");
    if (rc != IGX_OK || n_envs == 0) return rc;
    if (!b->reset || !b->rew) return igx::fail(IGX_E_NULL, "igx_ant_imitate_finish: reset / rew is NULL");
    if (b->time_outs && !b->progress) return igx::fail(IGX_E_NULL, "igx_ant_imitate_finish: time_outs needs progress");
    hipLaunchKernelGGL(igx::imitate_finish_kernel, dim3((unsigned)((n_envs + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *b, *im, n_envs, p->max_episode_length_m1);
    return igx::launch_status("igx_ant_imitate_finish");
}

extern "C" int igx_ant_imitate_step(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im,
                                    int64_t n_envs, igx_stream_t stream)
{
    int rc = igx_ant_imitate_observe(p, b, im, n_envs, stream);
    if (rc != IGX_OK || n_envs == 0) return rc;
    double *result = reinterpret_cast<double *>(im->workspace) + 5;  // the quantile lands in the state itself
    rc = igx_quantile_f64_windowed(im->pose_diff, n_envs, im->quantile, result, im->workspace, stream);
    if (rc != IGX_OK) return rc;
    return igx_ant_imitate_finish(p, b, im, n_envs, stream);
}

extern "C" int igx_ant_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n_envs, igx_stream_t stream)
{
    int rc = igx::check_loco(p, b, n_envs, 8, 24, false, "igx_ant_step");
    if (rc != IGX_OK) return rc;
    return igx::launch_loco<8, 24, false>(p, b, n_envs, stream, "igx_ant_step");
}

extern "C" int igx_humanoid_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n_envs,
                                 igx_stream_t stream)
{
    int rc = igx::check_loco(p, b, n_envs, 21, 12, true, "igx_humanoid_step");
    if (rc != IGX_OK) return rc;
    return igx::launch_loco<21, 12, true>(p, b, n_envs, stream, "igx_humanoid_step");
}

extern "C" int igx_ant_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m,
                             igx_stream_t stream)
{
    return igx::launch_loco_reset<8>(p, b, env_ids, m, stream, "igx_ant_reset");
}

extern "C" int igx_humanoid_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m,
                                  igx_stream_t stream)
{
    return igx::launch_loco_reset<21>(p, b, env_ids, m, stream, "igx_humanoid_reset");
}
