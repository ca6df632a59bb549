/*
 * igx_oracle.c - CPU restatement of the reference's per-step tensor path.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the checker the GPU kernels are
 * compared against; it is never linked into, imported by or called from the
 * product library (libigx.so) or the isaacgym_amd package.  Only tests/,
 * __graft_entry__.smoke() and the cpu_baseline leg of bench.py load it.
 *
 * It restates, one environment at a time in scalar fp32, what the reference's
 * torch.jit functions compute on CPU.  Every ATen op rounds its result to
 * fp32, so each C expression below is one rounding per operation (compile with
 * -ffp-contract=off); fmaf() appears exactly where the ATen CPU kernel fuses
 * (cross product, vector norm).  Parity pin: the .npz files of tests/golden/, produced by
 * oracle/gen_golden.py from the reference's own functions
 * (torch 2.10.0 CPU) - see tests/test_oracle_golden.py.
 *
 * Paths cited are relative to IsaacGymEnvs/isaacgymenvs/ in the reference.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include "../include/igx.h"

#define TWO_PI_F 6.2831855f /* float32(2*np.pi) */

#ifdef _OPENMP
#include <omp.h>
#endif
/* Environments are independent, so the per-env loops below are split over
 * host threads (used for the CPU-baseline timing); results do not depend on
 * the thread count. */
int orc_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n;
    return 1;
#endif
}

/* ---------------------------------------------------------------- RNG --- */
/* Philox4x32-10 (Salmon et al. 2011), the counter-based generator the
 * library uses when no pre-drawn uniforms are supplied.  Not part of the
 * reference (which draws torch.rand of a data-dependent shape); restated here
 * so the in-kernel draw can be checked bit for bit. */
static void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
}

/* j-th uniform in [0,1) of environment `env` at step `offset`. */
float orc_philox_uniform(uint64_t seed, uint64_t offset, int64_t env, int j)
{
    uint32_t c[4];
    c[0] = (uint32_t)(uint64_t)env;
    c[1] = (uint32_t)((uint64_t)env >> 32);
    c[2] = (uint32_t)offset;
    c[3] = (((uint32_t)(offset >> 32) & 0xFFFFFFu) << 8) | (uint32_t)(j >> 2);
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    return (float)(c[j & 3] >> 8) * (1.0f / 16777216.0f);
}

static float draw(const float *rand, int64_t e, int per_env, int j, uint64_t seed, uint64_t off, int64_t env_offset)
{
    /* given draws are indexed by the local environment, the generator by its index in the whole job */
    return rand ? rand[e * per_env + j] : orc_philox_uniform(seed, off, e + env_offset, j);
}

/* ------------------------------------------------------ math helpers --- */
/* utils/torch_jit_utils.py:41-62 quat_mul (8-multiply form), xyzw order */
static void quat_mul(const float a[4], const float b[4], float o[4])
{
    float x1 = a[0], y1 = a[1], z1 = a[2], w1 = a[3];
    float x2 = b[0], y2 = b[1], z2 = b[2], w2 = b[3];
    float ww = (z1 + x1) * (x2 + y2);
    float yy = (w1 - y1) * (w2 + z2);
    float zz = (w1 + y1) * (w2 - z2);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (z1 - x1) * (x2 - y2));
    float w = qq - ww + (z1 - y1) * (y2 - z2);
    float x = qq - xx + (x1 + w1) * (x2 + w2);
    float y = qq - yy + (w1 - x1) * (y2 + z2);
    float z = qq - zz + (z1 + y1) * (w2 - x2);
    o[0] = x; o[1] = y; o[2] = z; o[3] = w;
}

/* utils/torch_jit_utils.py:106-110 */
static void quat_conjugate(const float a[4], float o[4])
{
    o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3];
}

/* torch.cross(a,b,dim=-1) on CPU: each component is one fused multiply-add */
static void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = fmaf(a[1], b[2], -(a[2] * b[1]));
    o[1] = fmaf(a[2], b[0], -(a[0] * b[2]));
    o[2] = fmaf(a[0], b[1], -(a[1] * b[0]));
}

/* torch.bmm([N,1,3],[N,3,1]) on CPU: plain left-to-right dot, no fusion,
 * accumulated onto +0 (so an all-negative-zero product sum comes out +0) */
static float dot3(const float a[3], const float b[3])
{
    return ((0.0f + a[0] * b[0]) + a[1] * b[1]) + a[2] * b[2];
}

/* torch.norm(x, p=2, dim=-1) of a 3-vector on CPU */
static float norm3(const float v[3])
{
    float acc = v[0] * v[0];
    acc = fmaf(v[1], v[1], acc);
    acc = fmaf(v[2], v[2], acc);
    return sqrtf(acc);
}

/* norm of a 4-vector (quat_unit inside quat_from_angle_axis); the ATen
 * vectorised path is not a simple chain, so this feeds tolerance-class
 * outputs only */
static float norm4(const float v[4])
{
    float acc = v[0] * v[0];
    acc = fmaf(v[1], v[1], acc);
    acc = fmaf(v[2], v[2], acc);
    acc = fmaf(v[3], v[3], acc);
    return sqrtf(acc);
}

/* utils/torch_jit_utils.py:65-67 normalize: x / clamp(norm, min=1e-9) */
static void normalize3(const float v[3], float o[3])
{
    float n = norm3(v);
    if (n < 1e-9f) n = 1e-9f;
    o[0] = v[0] / n; o[1] = v[1] / n; o[2] = v[2] / n;
}

/* utils/torch_jit_utils.py:80-90 quat_rotate (sign=+1), :93-103
 * quat_rotate_inverse (sign=-1), :411-421 my_quat_rotate (sign=+1) */
static void quat_rotate_s(const float q[4], const float v[3], float o[3], int inverse)
{
    float qw = q[3];
    float s = 2.0f * (qw * qw) - 1.0f;
    float cr[3];
    cross3(q, v, cr);
    float d = dot3(q, v);
    for (int i = 0; i < 3; ++i) {
        float a = v[i] * s;
        float b = cr[i] * qw * 2.0f;
        float c = q[i] * d * 2.0f;
        o[i] = inverse ? (a - b + c) : (a + b + c);
    }
}

/* torch.min / torch.max of two tensors: a NaN on either side gives NaN (fminf / fmaxf would drop it) */
static float torch_min(float a, float b) { return (a != a || b != b) ? a + b : (b < a ? b : a); }
static float torch_max(float a, float b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }

/* torch.remainder(x, float32(2*pi)) */
static float remainder_2pi(float x)
{
    float m = fmodf(x, TWO_PI_F);
    if (m != 0.0f && m < 0.0f) m += TWO_PI_F;
    return m;
}

/* utils/torch_jit_utils.py:126-128 */
static float normalize_angle(float x)
{
    return atan2f(sinf(x), cosf(x));
}

/* utils/torch_jit_utils.py:175-195 get_euler_xyz; only roll and yaw are
 * consumed by the tasks on this path (pitch is computed and dropped) */
static void euler_roll_yaw(const float q[4], float *roll, float *yaw)
{
    float qx = q[0], qy = q[1], qz = q[2], qw = q[3];
    float sinr_cosp = 2.0f * (qw * qx + qy * qz);
    float cosr_cosp = qw * qw - qx * qx - qy * qy + qz * qz;
    float siny_cosp = 2.0f * (qw * qz + qx * qy);
    float cosy_cosp = qw * qw + qx * qx - qy * qy - qz * qz;
    *roll = remainder_2pi(atan2f(sinr_cosp, cosr_cosp));
    *yaw = remainder_2pi(atan2f(siny_cosp, cosy_cosp));
}

/* utils/torch_jit_utils.py:239-241 unscale */
static float unscale1(float x, float lo, float hi)
{
    return (2.0f * x - hi - lo) / (hi - lo);
}

/* torch.sum(x, dim=-1) over a contiguous row of D floats on CPU: 8 vector
 * lanes accumulate the full blocks of 8, the tail is summed first, then the
 * lanes are added left to right. */
static float row_sum(const float *x, int D)
{
    int nb = D / 8;
    float lane[8];
    if (nb == 0) {
        float r = x[0];
        for (int i = 1; i < D; ++i) r += x[i];
        return r;
    }
    for (int l = 0; l < 8; ++l) {
        float a = x[l];
        for (int b = 1; b < nb; ++b) a += x[8 * b + l];
        lane[l] = a;
    }
    int t = 8 * nb;
    float r;
    if (t < D) {
        r = x[t];
        for (int i = t + 1; i < D; ++i) r += x[i];
        r += lane[0];
    } else {
        r = lane[0];
    }
    for (int l = 1; l < 8; ++l) r += lane[l];
    return r;
}

/* ---------------------------------------------------------- Cartpole --- */
/* tasks/cartpole.py:170-178, :148-162, :135-146, :185-201 */
int orc_cartpole_step(const igx_cartpole_params *p, const igx_cartpole_buffers *b, int64_t n)
{
    const float half_pi = 1.57079637f; /* float32(np.pi / 2) */
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        float *ds = b->dof_state + e * 4;
        int64_t prog = b->progress[e] + 1;
        int64_t rst = b->reset[e];
        if (rst != 0) {
            float u0 = draw(b->rand, e, 4, 0, p->seed, p->rng_offset, p->env_offset);
            float u1 = draw(b->rand, e, 4, 1, p->seed, p->rng_offset, p->env_offset);
            float u2 = draw(b->rand, e, 4, 2, p->seed, p->rng_offset, p->env_offset);
            float u3 = draw(b->rand, e, 4, 3, p->seed, p->rng_offset, p->env_offset);
            ds[0] = 0.9f * (u0 - 0.5f);
            ds[2] = 0.9f * (u1 - 0.5f);
            ds[1] = 0.5f * (u2 - 0.5f);
            ds[3] = 0.5f * (u3 - 0.5f);
            rst = 0;
            prog = 0;
        }
        float cart_pos = ds[0], cart_vel = ds[1], pole_angle = ds[2], pole_vel = ds[3];
        float *o = b->obs + e * 4;
        o[0] = cart_pos; o[1] = cart_vel; o[2] = pole_angle; o[3] = pole_vel;
        float reward;
        if (p->reward_variant == 1) /* tasks/cartpole_sample.py:322 */
            reward = 1.0f - 0.001f * pole_angle * pole_angle - 0.001f * fabsf(cart_vel) - 0.001f * fabsf(pole_vel);
        else
            reward = 1.0f - pole_angle * pole_angle - 0.01f * fabsf(cart_vel) - 0.005f * fabsf(pole_vel);
        if (fabsf(cart_pos) > p->reset_dist) reward = -2.0f;
        if (fabsf(pole_angle) > half_pi) reward = -2.0f;
        if (fabsf(cart_pos) > p->reset_dist) rst = 1;
        if (fabsf(pole_angle) > half_pi) rst = 1;
        if ((float)prog >= p->max_episode_length_m1) rst = 1;
        b->rew[e] = reward;
        b->reset[e] = rst;
        b->progress[e] = prog;
        /* the tail of VecTask.step (tasks/base/vec_task.py:404, :412) */
        if (b->obs_clamped)
            for (int c = 0; c < 4; ++c) { /* torch.clamp: NaN stays NaN */
                float v = b->obs[e * 4 + c];
                v = v < -p->clip_obs ? -p->clip_obs : v;
                b->obs_clamped[e * 4 + c] = v > p->clip_obs ? p->clip_obs : v;
            }
        if (b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && rst != 0) ? 1 : 0;
    }
    return 0;
}

/* tasks/cartpole.py:148-162 on an explicit id list */
int orc_cartpole_reset(const igx_cartpole_params *p, const igx_cartpole_buffers *b, const int64_t *env_ids, int64_t m)
{
    for (int64_t i = 0; i < m; ++i) {
        const int64_t e = env_ids[i];
        float *ds = b->dof_state + e * 4;
        float u0 = draw(b->rand, e, 4, 0, p->seed, p->rng_offset, p->env_offset);
        float u1 = draw(b->rand, e, 4, 1, p->seed, p->rng_offset, p->env_offset);
        float u2 = draw(b->rand, e, 4, 2, p->seed, p->rng_offset, p->env_offset);
        float u3 = draw(b->rand, e, 4, 3, p->seed, p->rng_offset, p->env_offset);
        ds[0] = 0.9f * (u0 - 0.5f);
        ds[2] = 0.9f * (u1 - 0.5f);
        ds[1] = 0.5f * (u2 - 0.5f);
        ds[3] = 0.5f * (u3 - 0.5f);
        b->reset[e] = 0;
        b->progress[e] = 0;
    }
    return 0;
}

/* tasks/ant.py:339-366 / tasks/humanoid.py:360-390 on an explicit id list */
int orc_loco_reset(const igx_loco_params *p, const igx_loco_buffers *b, const int64_t *env_ids, int64_t m)
{
    const int D = p->num_dof;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t e = env_ids[i];
        float *ds = b->dof_state + e * D * 2;
        for (int j = 0; j < D; ++j) {
            float up = draw(b->rand, e, 2 * D, j, p->seed, p->rng_offset, p->env_offset);
            float uv = draw(b->rand, e, 2 * D, D + j, p->seed, p->rng_offset, p->env_offset);
            float pos = 0.4f * up + (-0.2f);
            float vel = 0.2f * uv + (-0.1f);
            float t = b->initial_dof_pos[e * D + j] + pos;
            t = torch_min(t, p->dof_limits_upper[j]);
            t = torch_max(t, p->dof_limits_lower[j]);
            ds[2 * j] = t;
            ds[2 * j + 1] = vel;
        }
        const float *ir = b->initial_root_states + e * 13;
        float tt[3] = { p->targets[0] - ir[0], p->targets[1] - ir[1], 0.0f };
        float pot = -norm3(tt) / p->dt;
        b->potentials[e] = pot;
        b->prev_potentials[e] = pot;
        b->progress[e] = 0;
        b->reset[e] = 0;
    }
    return 0;
}

/* ------------------------------------------------- Ant and Humanoid --- */
/* tasks/ant.py:376-386, :339-366, :463-497, :414-460, :332-337 and
 * tasks/humanoid.py:398-407, :360-390, :489-524, :434-486.
 * humanoid != 0 selects the Humanoid observation layout and reward. */
static int loco_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n, int humanoid,
                     const igx_imitate_args *im)
{
    const int D = p->num_dof;
    const int S = p->num_sensor_floats;
    const int NOBS = humanoid ? 12 + 3 * D + S + D : 12 + 2 * D + S + D;
    float ratio[IGX_MAX_DOF];
    for (int j = 0; j < D; ++j) ratio[j] = humanoid ? p->motor_efforts[j] / p->max_motor_effort : 0.0f;

#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const float *root = b->root_states + e * 13;
        float *ds = b->dof_state + e * D * 2;
        const float *act = b->actions + e * D;
        float *obs = b->obs + e * NOBS;

        int64_t prog = b->progress[e] + 1;
        int64_t rst = b->reset[e];
        float pot = b->potentials[e];
        if (b->randomize) b->randomize[e] += 1;

        if (rst != 0) { /* reset_idx */
            for (int j = 0; j < D; ++j) {
                float up = draw(b->rand, e, 2 * D, j, p->seed, p->rng_offset, p->env_offset);
                float uv = draw(b->rand, e, 2 * D, D + j, p->seed, p->rng_offset, p->env_offset);
                /* torch_rand_float: (upper - lower) * rand + lower */
                float pos = 0.4f * up + (-0.2f);
                float vel = 0.2f * uv + (-0.1f);
                float t = b->initial_dof_pos[e * D + j] + pos;
                /* tensor_clamp: max(min(t, hi), lo) */
                t = torch_min(t, p->dof_limits_upper[j]);
                t = torch_max(t, p->dof_limits_lower[j]);
                ds[2 * j] = t;
                ds[2 * j + 1] = vel;
            }
            const float *ir = b->initial_root_states + e * 13;
            float tt[3] = { p->targets[0] - ir[0], p->targets[1] - ir[1], 0.0f };
            pot = -norm3(tt) / p->dt;
            prog = 0;
            rst = 0;
        }

        /* compute_*_observations */
        float to_target[3] = { p->targets[0] - root[0], p->targets[1] - root[1], 0.0f };
        float prev_pot = pot;
        pot = -norm3(to_target) / p->dt;

        float target_dirs[3];
        normalize3(to_target, target_dirs);
        float tq[4];
        quat_mul(root + 3, p->inv_start_rot, tq);
        float up_vec[3], heading_vec[3];
        quat_rotate_s(tq, p->basis_vec1, up_vec, 0);
        quat_rotate_s(tq, p->basis_vec0, heading_vec, 0);
        float up_proj = up_vec[2];
        float heading_proj = dot3(heading_vec, target_dirs);

        float vel_loc[3], angvel_loc[3];
        quat_rotate_s(tq, root + 7, vel_loc, 1);
        quat_rotate_s(tq, root + 10, angvel_loc, 1);
        float roll, yaw;
        euler_roll_yaw(tq, &roll, &yaw);
        float walk_target_angle = atan2f(p->targets[2] - root[2], p->targets[0] - root[0]);
        float angle_to_target = walk_target_angle - yaw;
        if (humanoid) {
            roll = normalize_angle(roll);
            yaw = normalize_angle(yaw);
            angle_to_target = normalize_angle(angle_to_target);
        }

        obs[0] = root[2];
        for (int i = 0; i < 3; ++i) obs[1 + i] = vel_loc[i];
        for (int i = 0; i < 3; ++i)
            obs[4 + i] = humanoid ? angvel_loc[i] * p->angular_velocity_scale : angvel_loc[i];
        obs[7] = yaw; obs[8] = roll; obs[9] = angle_to_target;
        obs[10] = up_proj; obs[11] = heading_proj;
        int c = 12;
        for (int j = 0; j < D; ++j)
            obs[c + j] = unscale1(ds[2 * j], p->dof_limits_lower[j], p->dof_limits_upper[j]);
        c += D;
        for (int j = 0; j < D; ++j) obs[c + j] = ds[2 * j + 1] * p->dof_vel_scale;
        c += D;
        if (humanoid) {
            for (int j = 0; j < D; ++j) obs[c + j] = b->dof_force[e * D + j] * p->contact_force_scale;
            c += D;
        }
        for (int j = 0; j < S; ++j) obs[c + j] = b->sensors[e * S + j] * p->contact_force_scale;
        c += S;
        for (int j = 0; j < D; ++j) obs[c + j] = act[j];

        b->potentials[e] = pot;
        b->prev_potentials[e] = prev_pot;
        for (int i = 0; i < 3; ++i) {
            b->up_vec[e * 3 + i] = up_vec[i];
            b->heading_vec[e * 3 + i] = heading_vec[i];
        }

        /* compute_*_reward */
        float heading_reward = (obs[11] > 0.8f) ? p->heading_weight : p->heading_weight * obs[11] / 0.8f;
        float up_reward = (obs[10] > 0.93f) ? 0.0f + p->up_weight : 0.0f;
        float tmp[IGX_MAX_DOF];
        for (int j = 0; j < D; ++j) tmp[j] = act[j] * act[j];
        float actions_cost = row_sum(tmp, D);
        float total;
        float progress_reward = pot - prev_pot;
        if (im) { /* tasks/ant_imitate.py:405-419, :508-519 (fp64: the motion table promotes) */
            int64_t idx = im->target_pose_idx[e] + 1;
            im->target_pose_idx[e] = idx;
            if (idx < 0) idx = 0;
            if (idx >= im->num_frames) idx = im->num_frames - 1;
            const double *m = im->motion + idx * D;
            double diff[IGX_MAX_DOF];
            for (int j = 0; j < D; ++j) {
                double lo = (double)p->dof_limits_lower[j], hi = (double)p->dof_limits_upper[j];
                double span = (double)(float)(p->dof_limits_upper[j] - p->dof_limits_lower[j]);
                double scaled = (2.0 * m[j] - hi - lo) / span;
                diff[j] = fabs((double)obs[12 + j] - scaled);
            }
            /* torch.sum over 8 doubles: 4 lanes over 2 blocks, lanes left to right */
            double pose_diff = (((diff[0] + diff[4]) + (diff[1] + diff[5])) + (diff[2] + diff[6])) + (diff[3] + diff[7]);
            im->pose_diff[e] = pose_diff;
            total = (float)exp(-pose_diff);
        } else if (!humanoid) {
            for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * obs[12 + D + j]);
            float electricity_cost = row_sum(tmp, D);
            int64_t cnt = 0;
            for (int j = 0; j < D; ++j) cnt += (obs[12 + j] > 0.99f);
            total = progress_reward + 0.5f + up_reward + heading_reward
                  - p->actions_cost_scale * actions_cost - p->energy_cost_scale * electricity_cost
                  - (float)cnt * p->joints_at_limit_cost_scale;
        } else {
            for (int j = 0; j < D; ++j) {
                float ab = fabsf(obs[12 + j]);
                float scaled = p->joints_at_limit_cost_scale * (ab - 0.98f) / 0.02f;
                tmp[j] = ((ab > 0.98f) ? 1.0f : 0.0f) * scaled * ratio[j];
            }
            float dof_at_limit_cost = row_sum(tmp, D);
            for (int j = 0; j < D; ++j) tmp[j] = fabsf(act[j] * obs[12 + D + j]) * ratio[j];
            float electricity_cost = row_sum(tmp, D);
            total = progress_reward + 2.0f + up_reward + heading_reward
                  - p->actions_cost_scale * actions_cost - p->energy_cost_scale * electricity_cost
                  - dof_at_limit_cost;
        }
        if (obs[0] < p->termination_height) total = im ? (float)im->death_cost : p->death_cost;
        if (obs[0] < p->termination_height) rst = 1;
        if ((float)prog >= p->max_episode_length_m1) rst = 1;

        b->rew[e] = total;
        b->reset[e] = rst;
        b->progress[e] = prog;
        if (b->true_objective) b->true_objective[e] = root[7];
        /* the tail of VecTask.step (tasks/base/vec_task.py:404, :412); the imitation variant decides its
         * time-out flag in orc_ant_imitate_finish, after the threshold has set the last reset flags */
        if (!im && b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && rst != 0) ? 1 : 0;
        if (b->obs_clamped) {
            for (int c = 0; c < NOBS; ++c) { /* torch.clamp: NaN stays NaN */
                float v = obs[c];
                v = v < -p->clip_obs ? -p->clip_obs : v;
                b->obs_clamped[e * NOBS + c] = v > p->clip_obs ? p->clip_obs : v;
            }
        }
    }
    return 0;
}

int orc_ant_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n)
{
    return loco_step(p, b, n, 0, NULL);
}

int orc_humanoid_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n)
{
    return loco_step(p, b, n, 1, NULL);
}

/* ------------------------------------------- torch_jit_utils, batched --- */
/* utils/torch_jit_utils.py:113-115 quat_unit = normalize of a 4-vector */
static void quat_unit(const float q[4], float o[4])
{
    float n = norm4(q);
    if (n < 1e-9f) n = 1e-9f;
    for (int i = 0; i < 4; ++i) o[i] = q[i] / n;
}

/* :118-123 */
static void quat_from_angle_axis(float angle, const float axis[3], float o[4])
{
    float theta = angle / 2.0f;
    float a[3];
    normalize3(axis, a);
    float s = sinf(theta);
    float q[4] = { a[0] * s, a[1] * s, a[2] * s, cosf(theta) };
    quat_unit(q, o);
}

/* :570-593 exp_map_to_angle_axis + exp_map_to_quat */
static void exp_map_to_quat(const float e[3], float o[4])
{
    float n = norm3(e);
    float axis[3] = { e[0] / n, e[1] / n, e[2] / n };
    float angle = normalize_angle(n);
    if (!(angle > 1e-5f)) {
        angle = 0.0f;
        axis[0] = 0.0f; axis[1] = 0.0f; axis[2] = 1.0f;
    }
    quat_from_angle_axis(angle, axis, o);
}

/* :423-444 */
static void quat_to_angle_axis(const float q[4], float *angle, float axis[3])
{
    float sin_theta = sqrtf(1.0f - q[3] * q[3]);
    float a = normalize_angle(2.0f * acosf(q[3]));
    if (sin_theta > 1e-5f) {
        *angle = a;
        axis[0] = q[0] / sin_theta; axis[1] = q[1] / sin_theta; axis[2] = q[2] / sin_theta;
    } else {
        *angle = 0.0f;
        axis[0] = 0.0f; axis[1] = 0.0f; axis[2] = 1.0f;
    }
}

/* :454-461 */
static void quat_to_exp_map(const float q[4], float o[3])
{
    float angle, axis[3];
    quat_to_angle_axis(q, &angle, axis);
    for (int i = 0; i < 3; ++i) o[i] = angle * axis[i];
}

/* :595-628 */
static void slerp(const float q0[4], const float q1in[4], float t, float o[4])
{
    float q1[4] = { q1in[0], q1in[1], q1in[2], q1in[3] };
    float c = q0[3] * q1[3] + q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2];
    if (c < 0.0f) for (int i = 0; i < 4; ++i) q1[i] = -q1[i];
    c = fabsf(c);
    float half_theta = acosf(c);
    float s = sqrtf(1.0f - c * c);
    float ra = sinf((1.0f - t) * half_theta) / s;
    float rb = sinf(t * half_theta) / s;
    for (int i = 0; i < 4; ++i) o[i] = ra * q0[i] + rb * q1[i];
    if (fabsf(s) < 0.001f) for (int i = 0; i < 4; ++i) o[i] = 0.5f * q0[i] + 0.5f * q1[i];
    if (fabsf(c) >= 1.0f) for (int i = 0; i < 4; ++i) o[i] = q0[i];
}

/* :630-641 / :656-667 */
static void calc_heading_quat_inv(const float q[4], float o[4])
{
    const float ref[3] = { 1.0f, 0.0f, 0.0f }, zax[3] = { 0.0f, 0.0f, 1.0f };
    float d[3];
    quat_rotate_s(q, ref, d, 0);
    float heading = atan2f(d[1], d[0]);
    quat_from_angle_axis(-heading, zax, o);
}

/* :548-561 */
static void quat_to_tan_norm(const float q[4], float o[6])
{
    const float rt[3] = { 1.0f, 0.0f, 0.0f }, rn[3] = { 0.0f, 0.0f, 1.0f };
    quat_rotate_s(q, rt, o, 0);
    quat_rotate_s(q, rn, o + 3, 0);
}

int orc_quat_mul(const float *a, const float *b, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; quat_mul(a + 4 * i, b + 4 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}
int orc_quat_conjugate(const float *a, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; quat_conjugate(a + 4 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}
int orc_quat_rotate(const float *q, const float *v, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[3]; quat_rotate_s(q + 4 * i, v + 3 * i, t, 0); memcpy(o + 3 * i, t, 12); }
    return 0;
}
int orc_quat_rotate_inverse(const float *q, const float *v, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[3]; quat_rotate_s(q + 4 * i, v + 3 * i, t, 1); memcpy(o + 3 * i, t, 12); }
    return 0;
}
int orc_slerp(const float *q0, const float *q1, const float *t, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float r[4]; slerp(q0 + 4 * i, q1 + 4 * i, t[i], r); memcpy(o + 4 * i, r, 16); }
    return 0;
}
int orc_scale(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    for (int64_t i = 0; i < n * d; ++i) { int c = (int)(i % d); o[i] = 0.5f * (x[i] + 1.0f) * (hi[c] - lo[c]) + lo[c]; }
    return 0;
}
int orc_unscale(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    for (int64_t i = 0; i < n * d; ++i) { int c = (int)(i % d); o[i] = unscale1(x[i], lo[c], hi[c]); }
    return 0;
}
int orc_tensor_clamp(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    for (int64_t i = 0; i < n * d; ++i) { int c = (int)(i % d); o[i] = torch_max(torch_min(x[i], hi[c]), lo[c]); }
    return 0;
}
int orc_quat_from_angle_axis(const float *angle, const float *axis, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; quat_from_angle_axis(angle[i], axis + 3 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}
int orc_normalize_angle(const float *x, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) o[i] = normalize_angle(x[i]);
    return 0;
}
int orc_quat_to_exp_map(const float *q, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[3]; quat_to_exp_map(q + 4 * i, t); memcpy(o + 3 * i, t, 12); }
    return 0;
}
int orc_exp_map_to_quat(const float *e, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; exp_map_to_quat(e + 3 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}
int orc_quat_to_tan_norm(const float *q, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[6]; quat_to_tan_norm(q + 4 * i, t); memcpy(o + 6 * i, t, 24); }
    return 0;
}
int orc_calc_heading_quat_inv(const float *q, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; calc_heading_quat_inv(q + 4 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}

/* ---- the rest of utils/torch_jit_utils.py ---- */
/* :70-77 quat_apply */
static void quat_apply(const float a[4], const float b[3], float o[3])
{
    float c[3], t[3], u[3];
    cross3(a, b, c);
    for (int i = 0; i < 3; ++i) t[i] = c[i] * 2.0f;
    cross3(a, t, u);
    for (int i = 0; i < 3; ++i) o[i] = b[i] + a[3] * t[i] + u[i];
}

/* :168-172 copysign(a, b) = |a| * sign(b); torch.sign is 0 for +-0 and NaN */
static float copysign_sign(float a, float b)
{
    return fabsf(a) * (float)((0.0f < b) - (b < 0.0f));
}

/* :184-187, :195 pitch of get_euler_xyz */
static float euler_pitch(const float q[4])
{
    float sinp = 2.0f * (q[3] * q[1] - q[2] * q[0]);
    float pitch = (fabsf(sinp) >= 1.0f) ? copysign_sign(1.5707964f, sinp) : asinf(sinp);
    return remainder_2pi(pitch);
}

/* :198-212 */
static void quat_from_euler_xyz(float roll, float pitch, float yaw, float o[4])
{
    float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
    float cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
    float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
    o[3] = cy * cr * cp + sy * sr * sp;
    o[0] = cy * sr * cp - sy * cr * sp;
    o[1] = cy * cr * sp + sy * sr * cp;
    o[2] = sy * cr * cp - cy * sr * sp;
}

/* :570-587 */
static void exp_map_to_angle_axis(const float e[3], float *angle, float axis[3])
{
    float n = norm3(e);
    float a = normalize_angle(n);
    if (a > 1e-5f) {
        *angle = a;
        axis[0] = e[0] / n; axis[1] = e[1] / n; axis[2] = e[2] / n;
    } else {
        *angle = 0.0f;
        axis[0] = 0.0f; axis[1] = 0.0f; axis[2] = 1.0f;
    }
}

int orc_normalize(const float *x, float *o, int64_t n, int32_t d)
{
    if (d != 3 && d != 4) return -1;
    for (int64_t i = 0; i < n; ++i) {
        if (d == 3) normalize3(x + 3 * i, o + 3 * i);
        else { float t[4]; quat_unit(x + 4 * i, t); memcpy(o + 4 * i, t, 16); }
    }
    return 0;
}
int orc_quat_unit(const float *q, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[4]; quat_unit(q + 4 * i, t); memcpy(o + 4 * i, t, 16); }
    return 0;
}
int orc_quat_apply(const float *a, const float *b, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) { float t[3]; quat_apply(a + 4 * i, b + 3 * i, t); memcpy(o + 3 * i, t, 12); }
    return 0;
}
/* :131-134 */
int orc_tf_inverse(const float *q, const float *t, float *qo, float *to, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float qi[4], r[3];
        quat_conjugate(q + 4 * i, qi);
        quat_apply(qi, t + 3 * i, r);
        memcpy(qo + 4 * i, qi, 16);
        for (int k = 0; k < 3; ++k) to[3 * i + k] = -r[k];
    }
    return 0;
}
/* :137-139 */
int orc_tf_apply(const float *q, const float *t, const float *v, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float r[3];
        quat_apply(q + 4 * i, v + 3 * i, r);
        for (int k = 0; k < 3; ++k) o[3 * i + k] = r[k] + t[3 * i + k];
    }
    return 0;
}
/* :147-149 */
int orc_tf_combine(const float *q1, const float *t1, const float *q2, const float *t2, float *qo, float *to, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float qq[4], r[3];
        quat_mul(q1 + 4 * i, q2 + 4 * i, qq);
        quat_apply(q1 + 4 * i, t2 + 3 * i, r);
        memcpy(qo + 4 * i, qq, 16);
        for (int k = 0; k < 3; ++k) to[3 * i + k] = r[k] + t1[3 * i + k];
    }
    return 0;
}
int orc_copysign(float a, const float *b, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) o[i] = copysign_sign(a, b[i]);
    return 0;
}
/* :175-195 */
int orc_get_euler_xyz(const float *q, float *roll, float *pitch, float *yaw, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        euler_roll_yaw(q + 4 * i, roll + i, yaw + i);
        pitch[i] = euler_pitch(q + 4 * i);
    }
    return 0;
}
int orc_quat_from_euler_xyz(const float *roll, const float *pitch, const float *yaw, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) quat_from_euler_xyz(roll[i], pitch[i], yaw[i], o + 4 * i);
    return 0;
}
/* :563-568 */
int orc_euler_xyz_to_exp_map(const float *roll, const float *pitch, const float *yaw, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float q[4];
        quat_from_euler_xyz(roll[i], pitch[i], yaw[i], q);
        quat_to_exp_map(q, o + 3 * i);
    }
    return 0;
}
/* :215-218; (upper - lower) is a Python float subtraction, the rest fp32 tensor arithmetic */
int orc_torch_rand_float(double lower, double upper, const float *u, uint64_t seed, uint64_t off, float *o, int64_t count)
{
    const float span = (float)(upper - lower), lo = (float)lower;
    for (int64_t i = 0; i < count; ++i) {
        float r = u ? u[i] : orc_philox_uniform(seed, off, i >> 2, (int)(i & 3));
        o[i] = span * r + lo;
    }
    return 0;
}
/* :248-263 */
int orc_compute_heading_and_up(const float *rot, const float *isr, const float *to_target, const float *vec0,
                               const float *vec1, int32_t up_idx, float *tq, float *up_proj, float *heading_proj,
                               float *up_vec, float *heading_vec, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float dirs[3], q[4], up[3], hd[3];
        normalize3(to_target + 3 * i, dirs);
        quat_mul(rot + 4 * i, isr + 4 * i, q);
        quat_rotate_s(q, vec1 + 3 * i, up, 0);
        quat_rotate_s(q, vec0 + 3 * i, hd, 0);
        memcpy(tq + 4 * i, q, 16);
        up_proj[i] = up[up_idx];
        heading_proj[i] = dot3(hd, dirs);
        memcpy(up_vec + 3 * i, up, 12);
        memcpy(heading_vec + 3 * i, hd, 12);
    }
    return 0;
}
/* :266-277 */
int orc_compute_rot(const float *tq, const float *vel, const float *ang, const float *targets, const float *pos,
                    float *vel_loc, float *angvel_loc, float *roll, float *pitch, float *yaw, float *angle_to_target,
                    int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float a[3], b[3];
        quat_rotate_s(tq + 4 * i, vel + 3 * i, a, 1);
        quat_rotate_s(tq + 4 * i, ang + 3 * i, b, 1);
        memcpy(vel_loc + 3 * i, a, 12);
        memcpy(angvel_loc + 3 * i, b, 12);
        euler_roll_yaw(tq + 4 * i, roll + i, yaw + i);
        pitch[i] = euler_pitch(tq + 4 * i);
        angle_to_target[i] = atan2f(targets[3 * i + 2] - pos[3 * i + 2], targets[3 * i] - pos[3 * i]) - yaw[i];
    }
    return 0;
}
/* :280-285 */
int orc_quat_axis(const float *q, int32_t axis, float *o, int64_t n)
{
    float basis[3] = { 0.0f, 0.0f, 0.0f };
    basis[axis] = 1.0f;
    for (int64_t i = 0; i < n; ++i) { float t[3]; quat_rotate_s(q + 4 * i, basis, t, 0); memcpy(o + 3 * i, t, 12); }
    return 0;
}
/* :670-681 extract_positions_from_rigid_body_states: columns 13 i + {0, 1, 2}, i < num_bodies, of every row */
int orc_extract_positions(const float *x, int64_t row_stride, int32_t num_bodies, float *o, int64_t n)
{
    for (int64_t r = 0; r < n; ++r)
        for (int i = 0; i < num_bodies; ++i)
            for (int j = 0; j < 3; ++j) o[r * 3 * num_bodies + 3 * i + j] = x[r * row_stride + 13 * i + j];
    return 0;
}
/* :292-310 */
int orc_scale_transform(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    for (int64_t i = 0; i < n * d; ++i) { int c = (int)(i % d); float off = (lo[c] + hi[c]) * 0.5f; o[i] = 2.0f * (x[i] - off) / (hi[c] - lo[c]); }
    return 0;
}
/* :313-331 */
int orc_unscale_transform(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    for (int64_t i = 0; i < n * d; ++i) { int c = (int)(i % d); float off = (lo[c] + hi[c]) * 0.5f; o[i] = x[i] * (hi[c] - lo[c]) * 0.5f + off; }
    return 0;
}
/* :333-348 */
int orc_saturate(const float *x, const float *lo, const float *hi, float *o, int64_t n, int32_t d)
{
    return orc_tensor_clamp(x, lo, hi, o, n, d);
}
/* :354-373 */
int orc_quat_diff_rad(const float *a, const float *b, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        float bc[4], m[4];
        quat_conjugate(b + 4 * i, bc);
        quat_mul(a + 4 * i, bc, m);
        float nrm = norm3(m);
        o[i] = 2.0f * asinf(nrm > 1.0f ? 1.0f : nrm);
    }
    return 0;
}
/* :376-395 */
int orc_local_to_world_space(const float *local, const float *pose, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *ps = pose + 7 * i;
        float ql[4] = { local[3 * i], local[3 * i + 1], local[3 * i + 2], 0.0f };
        float qc[4], t[4], r[4];
        quat_conjugate(ps + 3, qc);
        quat_mul(ql, qc, t);
        quat_mul(ps + 3, t, r);
        for (int k = 0; k < 3; ++k) o[3 * i + k] = r[k] + ps[k];
    }
    return 0;
}
int orc_quat_to_angle_axis(const float *q, float *angle, float *axis, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) quat_to_angle_axis(q + 4 * i, angle + i, axis + 3 * i);
    return 0;
}
/* :446-452 */
int orc_angle_axis_to_exp_map(const float *angle, const float *axis, float *o, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) o[3 * i + k] = angle[i] * axis[3 * i + k];
    return 0;
}
int orc_exp_map_to_angle_axis(const float *e, float *angle, float *axis, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) exp_map_to_angle_axis(e + 3 * i, angle + i, axis + 3 * i);
    return 0;
}
/* :630-641 */
int orc_calc_heading(const float *q, float *o, int64_t n)
{
    const float ref[3] = { 1.0f, 0.0f, 0.0f };
    for (int64_t i = 0; i < n; ++i) { float d[3]; quat_rotate_s(q + 4 * i, ref, d, 0); o[i] = atan2f(d[1], d[0]); }
    return 0;
}
/* :643-654 */
int orc_calc_heading_quat(const float *q, float *o, int64_t n)
{
    const float ref[3] = { 1.0f, 0.0f, 0.0f }, zax[3] = { 0.0f, 0.0f, 1.0f };
    for (int64_t i = 0; i < n; ++i) {
        float d[3], t[4];
        quat_rotate_s(q + 4 * i, ref, d, 0);
        quat_from_angle_axis(atan2f(d[1], d[0]), zax, t);
        memcpy(o + 4 * i, t, 16);
    }
    return 0;
}

/* torch.clamp(row, -clip, clip) of one observation row: NaN stays NaN (tasks/base/vec_task.py:412) */
static void clamp_row(const float *src, float *dst, int count, float clip)
{
    for (int c = 0; c < count; ++c) {
        float v = src[c];
        v = v < -clip ? -clip : v;
        dst[c] = v > clip ? clip : v;
    }
}

/* ------------------------------------------------------- AllegroHand --- */
#define PI_F 3.14159274f /* float32(np.pi) */

/* tasks/allegro_hand.py:722-725 randomize_rotation */
static void randomize_rotation(float r0, float r1, float o[4])
{
    const float xu[3] = { 1.0f, 0.0f, 0.0f }, yu[3] = { 0.0f, 1.0f, 0.0f };
    float a[4], b[4];
    quat_from_angle_axis(r0 * PI_F, xu, a);
    quat_from_angle_axis(r1 * PI_F, yu, b);
    quat_mul(a, b, o);
}

/* tasks/allegro_hand.py:728-732 randomize_rotation_pen (max_angle = 0.3) */
static void randomize_rotation_pen(float r0, float o[4])
{
    const float xu[3] = { 1.0f, 0.0f, 0.0f }, zu[3] = { 0.0f, 0.0f, 1.0f };
    float a[4], b[4];
    quat_from_angle_axis(1.57079637f + r0 * 0.3f, xu, a);
    quat_from_angle_axis(r0 * PI_F, zu, b);
    quat_mul(a, b, o);
}

/* tasks/allegro_hand.py:506-522 reset_target_pose for one env */
static void allegro_reset_target_pose(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t e,
                                      float r0, float r1)
{
    float rot[4];
    randomize_rotation(r0, r1, rot);
    float *gs = b->goal_states + e * 13;
    const float *gi = b->goal_init_state + e * 13;
    for (int i = 0; i < 3; ++i) gs[i] = gi[i];
    for (int i = 0; i < 4; ++i) gs[3 + i] = rot[i];
    float *row = b->root_state_tensor + b->goal_object_indices[e] * 13;
    for (int i = 0; i < 3; ++i) row[i] = gs[i] + p->goal_displacement[i];
    for (int i = 0; i < 4; ++i) row[3 + i] = rot[i];
    for (int i = 7; i < 13; ++i) row[i] = 0.0f;
    b->reset_goal[e] = 0;
}

/* tasks/allegro_hand.py:586-615 pre_physics_step (+ :524-584 reset_idx) */
int orc_allegro_pre_physics(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n)
{
    const int D = IGX_ALLEGRO_DOF, R = IGX_ALLEGRO_RAND_PER_ENV;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const int rst = b->reset[e] != 0, rg = b->reset_goal[e] != 0;
        float r[IGX_ALLEGRO_RAND_PER_ENV];
        if (rst || rg)
            for (int k = 0; k < R; ++k) /* torch_rand_float(-1, 1): (upper - lower) * u + lower */
                r[k] = 2.0f * draw(b->rand, e, R, k, p->seed, p->rng_offset, p->env_offset) + (-1.0f);
        if (rg) allegro_reset_target_pose(p, b, e, r[0], r[1]);
        if (rst) {
            const float *rf = r + 4; /* the 37 floats of reset_idx */
            allegro_reset_target_pose(p, b, e, r[41], r[42]);
            float *row = b->root_state_tensor + b->object_indices[e] * 13;
            const float *oi = b->object_init_state + e * 13;
            for (int i = 0; i < 13; ++i) row[i] = oi[i];
            row[0] = oi[0] + p->reset_position_noise * rf[0];
            row[1] = oi[1] + p->reset_position_noise * rf[1];
            row[2] = oi[2] + p->reset_position_noise * rf[2];
            float rot[4];
            if (p->ignore_z_rot) randomize_rotation_pen(rf[3], rot);
            else randomize_rotation(rf[3], rf[4], rot);
            for (int i = 0; i < 4; ++i) row[3 + i] = rot[i];
            for (int i = 7; i < 13; ++i) row[i] = 0.0f;
            for (int j = 0; j < D; ++j) {
                float delta_max = p->dof_upper[j] - 0.0f, delta_min = p->dof_lower[j] - 0.0f;
                float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rf[5 + j] + 1.0f);
                float pos = 0.0f + p->reset_dof_pos_noise * rand_delta;
                b->dof_state[(e * D + j) * 2] = pos;
                b->dof_state[(e * D + j) * 2 + 1] = 0.0f + p->reset_dof_vel_noise * rf[5 + D + j];
                b->prev_targets[e * D + j] = pos;
                b->cur_targets[e * D + j] = pos;
            }
            b->progress[e] = 0;
            b->reset[e] = 0;
            b->successes[e] = 0.0f;
        }
        for (int j = 0; j < D; ++j) {
            float a = b->actions[e * D + j], lo = p->dof_lower[j], hi = p->dof_upper[j], c;
            if (b->actions_out) {  /* tasks/base/vec_task.py:384: torch.clamp(actions, -clip, clip), kept as self.actions */
                a = a < -p->clip_actions ? -p->clip_actions : a;
                a = a > p->clip_actions ? p->clip_actions : a;
                b->actions_out[e * D + j] = a;
            }
            if (p->use_relative_control) {
                c = b->prev_targets[e * D + j] + p->dof_speed_scale_dt * a;
                c = torch_max(torch_min(c, hi), lo);
            } else {
                c = 0.5f * (a + 1.0f) * (hi - lo) + lo;
                c = p->act_moving_average * c + p->one_minus_act_moving_average * b->prev_targets[e * D + j];
                c = torch_max(torch_min(c, hi), lo);
            }
            b->cur_targets[e * D + j] = c;
            b->prev_targets[e * D + j] = c;
        }
    }
    return 0;
}

static void allegro_obs_row(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t e, int obs_type,
                            const float *obj, const float *goal, const float qd[4], float *o)
{
    const int D = IGX_ALLEGRO_DOF;
    const float *ds = b->dof_state + e * D * 2;
    int c = 0;
    for (int j = 0; j < D; ++j) o[c + j] = unscale1(ds[2 * j], p->dof_lower[j], p->dof_upper[j]);
    c += D;
    if (obs_type != IGX_ALLEGRO_OBS_FULL_NO_VEL) {
        for (int j = 0; j < D; ++j) o[c + j] = p->vel_obs_scale * ds[2 * j + 1];
        c += D;
    }
    if (obs_type == IGX_ALLEGRO_OBS_FULL_STATE) {
        for (int j = 0; j < D; ++j) o[c + j] = p->force_torque_obs_scale * b->dof_force[e * D + j];
        c += D;
    }
    for (int i = 0; i < 7; ++i) o[c + i] = obj[i];
    c += 7;
    if (obs_type != IGX_ALLEGRO_OBS_FULL_NO_VEL) {
        for (int i = 0; i < 3; ++i) o[c + i] = obj[7 + i];
        for (int i = 0; i < 3; ++i) o[c + 3 + i] = p->vel_obs_scale * obj[10 + i];
        c += 6;
    }
    for (int i = 0; i < 7; ++i) o[c + i] = goal[i];
    c += 7;
    for (int i = 0; i < 4; ++i) o[c + i] = qd[i];
    c += 4;
    for (int j = 0; j < D; ++j) o[c + j] = b->actions[e * D + j];
}

/* tasks/allegro_hand.py:627-632 post_physics_step, :406-504 observations,
 * :663-719 compute_hand_reward */
typedef struct { uint64_t num_resets; double finished; } allegro_record;

static int allegro_post(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n, allegro_record *rec)
{
    const int D = IGX_ALLEGRO_DOF;
    const int nobs = p->obs_type == IGX_ALLEGRO_OBS_FULL_STATE ? 88 : p->obs_type == IGX_ALLEGRO_OBS_FULL ? 72 : 50;
    int64_t num_resets = 0;
    double fin = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : num_resets, fin)
    for (int64_t e = 0; e < n; ++e) {
        int64_t prog = b->progress[e] + 1;
        if (b->randomize) b->randomize[e] += 1;
        const float *obj = b->root_state_tensor + b->object_indices[e] * 13;
        const float *goal = b->goal_states + e * 13;
        float gconj[4], qd[4];
        quat_conjugate(goal + 3, gconj);
        quat_mul(obj + 3, gconj, qd);
        allegro_obs_row(p, b, e, p->obs_type, obj, goal, qd, b->obs + e * nobs);
        if (b->states) allegro_obs_row(p, b, e, IGX_ALLEGRO_OBS_FULL_STATE, obj, goal, qd, b->states + e * 88);
        if (b->obs_clamped) clamp_row(b->obs + e * nobs, b->obs_clamped + e * nobs, nobs, p->clip_obs);
        if (b->states && b->states_clamped) clamp_row(b->states + e * 88, b->states_clamped + e * 88, 88, p->clip_obs);

        /* compute_hand_reward */
        float dpos[3] = { obj[0] - goal[0], obj[1] - goal[1], obj[2] - goal[2] };
        float goal_dist = norm3(dpos);
        float cn = norm3(qd);
        float cc = cn > 1.0f ? 1.0f : cn; /* torch.clamp(max=1.0): a NaN stays NaN */
        float rot_dist = 2.0f * asinf(cc);
        float dist_rew = goal_dist * p->dist_reward_scale;
        float rot_rew = 1.0f / (fabsf(rot_dist) + p->rot_eps) * p->rot_reward_scale;
        float sq[IGX_ALLEGRO_DOF];
        for (int j = 0; j < D; ++j) sq[j] = b->actions[e * D + j] * b->actions[e * D + j];
        float action_penalty = row_sum(sq, D);
        float reward = dist_rew + rot_rew + action_penalty * p->action_penalty_scale;
        int success = cc <= p->success_threshold_c; /* == |rot_dist| <= success_tolerance */
        int64_t goal_resets = success ? 1 : b->reset_goal[e];
        float successes = b->successes[e] + (float)goal_resets;
        if (goal_resets == 1) reward = reward + p->reach_goal_bonus;
        int64_t resets = b->reset[e];
        if (goal_dist >= p->fall_dist) {
            reward = reward + p->fall_penalty;
            resets = 1;
        }
        if (p->max_consecutive_successes > 0) {
            if (success) prog = 0;
            if (successes >= (float)p->max_consecutive_successes) resets = 1;
        }
        int timed_out = (float)prog >= p->max_episode_length_m1;
        if (timed_out) resets = 1;
        if (p->max_consecutive_successes > 0 && timed_out) reward = reward + 0.5f * p->fall_penalty;

        b->rew[e] = reward;
        b->reset[e] = resets;
        b->reset_goal[e] = goal_resets;
        b->progress[e] = prog;
        b->successes[e] = successes;
        if (b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && resets != 0) ? 1 : 0;
        num_resets += resets;
        fin += (double)(successes * (float)resets);
    }
    rec->num_resets = (uint64_t)num_resets;
    rec->finished = fin;
    return 0;
}

static void allegro_finalize(const igx_allegro_params *p, const igx_allegro_buffers *b, const allegro_record *recs,
                             int count)
{
    uint64_t num_resets = 0;
    double fin = 0.0;
    for (int i = 0; i < count; ++i) {
        num_resets += recs[i].num_resets;
        fin += recs[i].finished;
    }
    if (num_resets > 0) {
        float f = (float)fin;
        b->consecutive_successes[0] =
            p->av_factor * f / (float)(int64_t)num_resets + p->one_minus_av_factor * b->consecutive_successes[0];
    }
}

int orc_allegro_post_physics(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n)
{
    allegro_record rec;
    allegro_post(p, b, n, &rec);
    allegro_finalize(p, b, &rec, 1);
    return 0;
}

/* one shard of an environment-sharded run: the record goes to b->workspace */
int orc_allegro_post_physics_shard(const igx_allegro_params *p, const igx_allegro_buffers *b, int64_t n)
{
    int rc = allegro_post(p, b, n, (allegro_record *)b->workspace);
    if (rc == 0 && b->shard_record) *(allegro_record *)b->shard_record = *(const allegro_record *)b->workspace;
    return rc;
}

/* a queue of steps: records[shard * num_steps + step], applied strictly in step order */
int orc_allegro_finalize_shards_queued(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                       int32_t num_shards, int32_t num_steps)
{
    const allegro_record *r = (const allegro_record *)records;
    for (int t = 0; t < num_steps; ++t) {
        allegro_record step[num_shards];
        for (int i = 0; i < num_shards; ++i) step[i] = r[(int64_t)i * num_steps + t];
        allegro_finalize(p, b, step, num_shards);
    }
    return 0;
}

int orc_allegro_finalize_shards(const igx_allegro_params *p, const igx_allegro_buffers *b, const void *records,
                                int32_t num_shards)
{
    allegro_finalize(p, b, (const allegro_record *)records, num_shards);
    return 0;
}

/* -------------------------------------------------------- ShadowHand --- */
/* tasks/shadow_hand.py:586-602 reset_target_pose for one env */
static void shadow_reset_target_pose(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t e, float r0, float r1)
{
    float rot[4];
    randomize_rotation(r0, r1, rot); /* :803-806, the same function as AllegroHand's */
    float *gs = b->goal_states + e * 13;
    const float *gi = b->goal_init_state + e * 13;
    for (int i = 0; i < 3; ++i) gs[i] = gi[i];
    for (int i = 0; i < 4; ++i) gs[3 + i] = rot[i];
    float *row = b->root_state_tensor + b->goal_object_indices[e] * 13;
    for (int i = 0; i < 3; ++i) row[i] = gs[i] + p->goal_displacement[i];
    for (int i = 0; i < 4; ++i) row[3 + i] = rot[i];
    for (int i = 7; i < 13; ++i) row[i] = 0.0f;
    b->reset_goal[e] = 0;
}

/* tasks/shadow_hand.py:670-698 pre_physics_step (+ :604-668 reset_idx).  Draws of an environment:
 * [0,4) the goal-only reset_target_pose, [4,57) the 53 floats of reset_idx, [57,61) its inner
 * reset_target_pose (torch draws them in the order 53, 4). */
int orc_shadow_pre_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n)
{
    const int D = IGX_SHADOW_DOF, A = IGX_SHADOW_ACTIONS, R = IGX_SHADOW_RAND_PER_ENV;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        const int rst = b->reset[e] != 0, rg = b->reset_goal[e] != 0;
        float r[IGX_SHADOW_RAND_PER_ENV];
        if (rst || rg)
            for (int k = 0; k < R; ++k) r[k] = 2.0f * draw(b->rand, e, R, k, p->seed, p->rng_offset, p->env_offset) + (-1.0f);
        if (rg) shadow_reset_target_pose(p, b, e, r[0], r[1]);
        if (rst) {
            const float *rf = r + 4;
            shadow_reset_target_pose(p, b, e, r[57], r[58]);
            float *row = b->root_state_tensor + b->object_indices[e] * 13;
            const float *oi = b->object_init_state + e * 13;
            for (int i = 0; i < 13; ++i) row[i] = oi[i];
            row[0] = oi[0] + p->reset_position_noise * rf[0];
            row[1] = oi[1] + p->reset_position_noise * rf[1];
            row[2] = oi[2] + p->reset_position_noise * rf[2]; /* up_axis_idx = 2 */
            float rot[4];
            if (p->ignore_z_rot) randomize_rotation_pen(rf[3], rot);
            else randomize_rotation(rf[3], rf[4], rot);
            for (int i = 0; i < 4; ++i) row[3 + i] = rot[i];
            for (int i = 7; i < 13; ++i) row[i] = 0.0f;
            for (int j = 0; j < D; ++j) { /* default pos / vel are zero (:163, shadow_hand_dof_default_vel) */
                float delta_max = p->dof_upper[j] - 0.0f, delta_min = p->dof_lower[j] - 0.0f;
                float rand_delta = delta_min + (delta_max - delta_min) * 0.5f * (rf[5 + j] + 1.0f);
                float pos = 0.0f + p->reset_dof_pos_noise * rand_delta;
                b->dof_state[(e * D + j) * 2] = pos;
                b->dof_state[(e * D + j) * 2 + 1] = 0.0f + p->reset_dof_vel_noise * rf[5 + D + j];
                b->prev_targets[e * D + j] = pos;
                b->cur_targets[e * D + j] = pos;
            }
            b->progress[e] = 0;
            b->reset[e] = 0;
            b->successes[e] = 0.0f;
        }
        for (int a = 0; a < A; ++a) { /* :684-697, actuated joints only */
            const int j = p->actuated_dof_indices[a];
            float x = b->actions[e * A + a], lo = p->dof_lower[j], hi = p->dof_upper[j], c;
            if (b->actions_out) {  /* tasks/base/vec_task.py:384 */
                x = x < -p->clip_actions ? -p->clip_actions : x;
                x = x > p->clip_actions ? p->clip_actions : x;
                b->actions_out[e * A + a] = x;
            }
            if (p->use_relative_control) {
                c = b->prev_targets[e * D + j] + p->dof_speed_scale_dt * x;
                c = torch_max(torch_min(c, hi), lo);
            } else {
                c = 0.5f * (x + 1.0f) * (hi - lo) + lo;
                c = p->act_moving_average * c + p->one_minus_act_moving_average * b->prev_targets[e * D + j];
                c = torch_max(torch_min(c, hi), lo);
            }
            b->cur_targets[e * D + j] = c;
            b->prev_targets[e * D + j] = c;
        }
    }
    return 0;
}

static int shadow_num_obs(int obs_type)
{
    return obs_type == IGX_SHADOW_OBS_OPENAI ? 42 : obs_type == IGX_SHADOW_OBS_FULL_NO_VEL ? 77
         : obs_type == IGX_SHADOW_OBS_FULL ? 157 : 211;
}

/* tasks/shadow_hand.py:473-584: one observation row in the given layout */
static void shadow_obs_row(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t e, int obs_type,
                           const float *obj, const float *goal, const float qd[4], float *o)
{
    const int D = IGX_SHADOW_DOF, A = IGX_SHADOW_ACTIONS, F = IGX_SHADOW_FINGERTIPS;
    const float *ds = b->dof_state + e * D * 2;
    const float *bodies = b->rigid_body_states + e * (int64_t)p->num_bodies * 13;
    int c = 0;
    if (obs_type == IGX_SHADOW_OBS_OPENAI) { /* :481-485 */
        for (int f = 0; f < F; ++f)
            for (int i = 0; i < 3; ++i) o[c++] = bodies[p->fingertip_handles[f] * 13 + i];
        for (int i = 0; i < 3; ++i) o[c++] = obj[i];
        for (int i = 0; i < 4; ++i) o[c++] = qd[i];
        for (int a = 0; a < A; ++a) o[c++] = b->actions[e * A + a];
        return;
    }
    const int vel = obs_type != IGX_SHADOW_OBS_FULL_NO_VEL, full_state = obs_type == IGX_SHADOW_OBS_FULL_STATE;
    for (int j = 0; j < D; ++j) o[c++] = unscale1(ds[2 * j], p->dof_lower[j], p->dof_upper[j]);
    if (vel)
        for (int j = 0; j < D; ++j) o[c++] = p->vel_obs_scale * ds[2 * j + 1];
    if (full_state)
        for (int j = 0; j < D; ++j) o[c++] = p->force_torque_obs_scale * b->dof_force[e * D + j];
    for (int i = 0; i < 7; ++i) o[c++] = obj[i];
    if (vel) {
        for (int i = 0; i < 3; ++i) o[c++] = obj[7 + i];
        for (int i = 0; i < 3; ++i) o[c++] = p->vel_obs_scale * obj[10 + i];
    }
    for (int i = 0; i < 7; ++i) o[c++] = goal[i];
    for (int i = 0; i < 4; ++i) o[c++] = qd[i];
    for (int f = 0; f < F; ++f) /* fingertip state (13) or position (3), unscaled */
        for (int i = 0; i < (vel ? 13 : 3); ++i) o[c++] = bodies[p->fingertip_handles[f] * 13 + i];
    if (full_state)
        for (int k = 0; k < 6 * F; ++k) o[c++] = p->force_torque_obs_scale * b->vec_sensor_tensor[e * 6 * F + k];
    for (int a = 0; a < A; ++a) o[c++] = b->actions[e * A + a];
}

/* tasks/shadow_hand.py:710-715 post_physics_step, :437-471 compute_observations, :747-800 compute_hand_reward */
static int shadow_post(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n, allegro_record *rec)
{
    const int A = IGX_SHADOW_ACTIONS;
    const int nobs = shadow_num_obs(p->obs_type);
    int64_t num_resets = 0;
    double fin = 0.0;
#pragma omp parallel for schedule(static) reduction(+ : num_resets, fin)
    for (int64_t e = 0; e < n; ++e) {
        int64_t prog = b->progress[e] + 1;
        if (b->randomize) b->randomize[e] += 1;
        const float *obj = b->root_state_tensor + b->object_indices[e] * 13;
        const float *goal = b->goal_states + e * 13;
        float gconj[4], qd[4];
        quat_conjugate(goal + 3, gconj);
        quat_mul(obj + 3, gconj, qd);
        shadow_obs_row(p, b, e, p->obs_type, obj, goal, qd, b->obs + e * nobs);
        if (b->states) shadow_obs_row(p, b, e, IGX_SHADOW_OBS_FULL_STATE, obj, goal, qd, b->states + e * 211);
        if (b->obs_clamped) clamp_row(b->obs + e * nobs, b->obs_clamped + e * nobs, nobs, p->clip_obs);
        if (b->states && b->states_clamped) clamp_row(b->states + e * 211, b->states_clamped + e * 211, 211, p->clip_obs);

        float dpos[3] = { obj[0] - goal[0], obj[1] - goal[1], obj[2] - goal[2] };
        float goal_dist = norm3(dpos);
        float cn = norm3(qd);
        float cc = cn > 1.0f ? 1.0f : cn; /* torch.clamp(max=1.0): a NaN stays NaN */
        float rot_dist = 2.0f * asinf(cc);
        float dist_rew = goal_dist * p->dist_reward_scale;
        float rot_rew = 1.0f / (fabsf(rot_dist) + p->rot_eps) * p->rot_reward_scale;
        float sq[IGX_SHADOW_ACTIONS];
        for (int a = 0; a < A; ++a) sq[a] = b->actions[e * A + a] * b->actions[e * A + a];
        float action_penalty = row_sum(sq, A);
        float reward = dist_rew + rot_rew + action_penalty * p->action_penalty_scale;
        int success = cc <= p->success_threshold_c;
        int64_t goal_resets = success ? 1 : b->reset_goal[e];
        float successes = b->successes[e] + (float)goal_resets;
        if (goal_resets == 1) reward = reward + p->reach_goal_bonus;
        int64_t resets = b->reset[e];
        if (goal_dist >= p->fall_dist) {
            reward = reward + p->fall_penalty;
            resets = 1;
        }
        if (p->max_consecutive_successes > 0) {
            if (success) prog = 0;
            if (successes >= (float)p->max_consecutive_successes) resets = 1;
        }
        int timed_out = (float)prog >= p->max_episode_length_m1;
        if (timed_out) resets = 1;
        if (p->max_consecutive_successes > 0 && timed_out) reward = reward + 0.5f * p->fall_penalty;

        b->rew[e] = reward;
        b->reset[e] = resets;
        b->reset_goal[e] = goal_resets;
        b->progress[e] = prog;
        b->successes[e] = successes;
        if (b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && resets != 0) ? 1 : 0;
        num_resets += resets;
        fin += (double)(successes * (float)resets);
    }
    rec->num_resets = (uint64_t)num_resets;
    rec->finished = fin;
    return 0;
}

static void shadow_finalize(const igx_shadow_params *p, const igx_shadow_buffers *b, const allegro_record *recs, int count)
{
    uint64_t num_resets = 0;
    double fin = 0.0;
    for (int i = 0; i < count; ++i) {
        num_resets += recs[i].num_resets;
        fin += recs[i].finished;
    }
    if (num_resets > 0) {
        float f = (float)fin;
        b->consecutive_successes[0] =
            p->av_factor * f / (float)(int64_t)num_resets + p->one_minus_av_factor * b->consecutive_successes[0];
    }
}

int orc_shadow_post_physics(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n)
{
    allegro_record rec;
    shadow_post(p, b, n, &rec);
    shadow_finalize(p, b, &rec, 1);
    return 0;
}

int orc_shadow_post_physics_shard(const igx_shadow_params *p, const igx_shadow_buffers *b, int64_t n)
{
    int rc = shadow_post(p, b, n, (allegro_record *)b->workspace);
    if (rc == 0 && b->shard_record) *(allegro_record *)b->shard_record = *(const allegro_record *)b->workspace;
    return rc;
}

int orc_shadow_finalize_shards_queued(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records,
                                      int32_t num_shards, int32_t num_steps)
{
    const allegro_record *r = (const allegro_record *)records;
    for (int t = 0; t < num_steps; ++t) {
        allegro_record step[num_shards];
        for (int i = 0; i < num_shards; ++i) step[i] = r[(int64_t)i * num_steps + t];
        shadow_finalize(p, b, step, num_shards);
    }
    return 0;
}

int orc_shadow_finalize_shards(const igx_shadow_params *p, const igx_shadow_buffers *b, const void *records, int32_t num_shards)
{
    shadow_finalize(p, b, (const allegro_record *)records, num_shards);
    return 0;
}

/* ------------------------------------------ AMP motion library / obs --- */
/* tasks/amp/humanoid_amp_base.py:43-44 */
static const int DOF_BODY_IDS[12] = { 1, 2, 3, 4, 6, 7, 9, 10, 11, 12, 13, 14 };
static const int DOF_OFFSETS[13] = { 0, 3, 6, 9, 10, 13, 14, 17, 18, 21, 24, 25, 28 };

/* tasks/amp/utils_amp/motion_lib.py:233-241 _calc_frame_blend, all fp64 */
static void calc_frame_blend(double time, double len, int64_t num_frames, double dt, int64_t *i0, int64_t *i1,
                             double *blend)
{
    double phase = time / len;
    if (phase < 0.0) phase = 0.0;
    if (phase > 1.0) phase = 1.0;
    int64_t f0 = (int64_t)(phase * (double)(num_frames - 1)); /* astype(int): truncation */
    int64_t f1 = f0 + 1 < num_frames - 1 ? f0 + 1 : num_frames - 1;
    *i0 = f0; *i1 = f1;
    *blend = (time - (double)f0 * dt) / dt;
}

/* one sample of MotionLib.get_motion_state (:83-153) + _local_rotation_to_dof (:265-293) */
static void motion_state_one(const igx_motion_tables *t, int64_t id, double time, float root_pos[3],
                             float root_rot[4], float dof_pos[28], float root_vel[3], float root_ang_vel[3],
                             float dof_vel[28], float key_pos[12], int64_t *fi0, int64_t *fi1, float *fblend)
{
    int64_t i0, i1;
    double bl;
    calc_frame_blend(time, t->motion_length[id], t->motion_num_frames[id], t->motion_dt[id], &i0, &i1, &bl);
    const float blend = (float)bl; /* to_torch(..., dtype=float) */
    const int64_t r0 = t->frame_offset[id] + i0, r1 = t->frame_offset[id] + i1;
    const float omb = 1.0f - blend;
    for (int i = 0; i < 3; ++i) root_pos[i] = omb * t->root_pos[r0 * 3 + i] + blend * t->root_pos[r1 * 3 + i];
    slerp(t->root_rot + r0 * 4, t->root_rot + r1 * 4, blend, root_rot);
    for (int i = 0; i < 12; ++i) key_pos[i] = omb * t->key_pos[r0 * 12 + i] + blend * t->key_pos[r1 * 12 + i];
    for (int i = 0; i < 3; ++i) {
        root_vel[i] = t->root_vel[r0 * 3 + i];
        root_ang_vel[i] = t->root_ang_vel[r0 * 3 + i];
    }
    for (int i = 0; i < 28; ++i) dof_vel[i] = t->dof_vel[r0 * 28 + i];
    for (int j = 0; j < 12; ++j) {
        const int body = DOF_BODY_IDS[j], off = DOF_OFFSETS[j], size = DOF_OFFSETS[j + 1] - off;
        float q[4];
        slerp(t->local_rot + (r0 * 15 + body) * 4, t->local_rot + (r1 * 15 + body) * 4, blend, q);
        if (size == 3) {
            quat_to_exp_map(q, dof_pos + off);
        } else {
            float theta, axis[3];
            quat_to_angle_axis(q, &theta, axis);
            theta = theta * axis[1]; /* joint assumed to turn about y */
            dof_pos[off] = normalize_angle(theta);
        }
    }
    if (fi0) *fi0 = i0;
    if (fi1) *fi1 = i1;
    if (fblend) *fblend = blend;
}

int orc_motion_state(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                     const igx_motion_state_out *o)
{
    for (int64_t s = 0; s < n; ++s) {
        float rp[3], rr[4], dp[28], rv[3], rav[3], dv[28], kp[12], bl;
        int64_t i0, i1;
        if (ids[s] < 0 || ids[s] >= t->num_motions) continue; /* a hole of a masked list */
        motion_state_one(t, ids[s], times[s], rp, rr, dp, rv, rav, dv, kp, &i0, &i1, &bl);
        if (o->root_pos) memcpy(o->root_pos + s * 3, rp, 12);
        if (o->root_rot) memcpy(o->root_rot + s * 4, rr, 16);
        if (o->dof_pos) memcpy(o->dof_pos + s * 28, dp, 112);
        if (o->root_vel) memcpy(o->root_vel + s * 3, rv, 12);
        if (o->root_ang_vel) memcpy(o->root_ang_vel + s * 3, rav, 12);
        if (o->dof_vel) memcpy(o->dof_vel + s * 28, dv, 112);
        if (o->key_pos) memcpy(o->key_pos + s * 12, kp, 48);
        if (o->frame_idx0) o->frame_idx0[s] = i0;
        if (o->frame_idx1) o->frame_idx1[s] = i1;
        if (o->blend) o->blend[s] = bl;
    }
    return 0;
}

/* tasks/humanoid_amp.py:337-371 build_amp_observations for one character */
static void amp_obs_one(const float root_pos[3], const float root_rot[4], const float root_vel[3],
                        const float root_ang_vel[3], const float *dof_pos, int dof_pos_stride, const float *dof_vel,
                        int dof_vel_stride, const float key_pos[12], int local_root_obs, float obs[105])
{
    float heading_rot[4], rro[4];
    calc_heading_quat_inv(root_rot, heading_rot);
    if (local_root_obs) quat_mul(heading_rot, root_rot, rro);
    else memcpy(rro, root_rot, 16);
    obs[0] = root_pos[2];
    quat_to_tan_norm(rro, obs + 1);
    quat_rotate_s(heading_rot, root_vel, obs + 7, 0);
    quat_rotate_s(heading_rot, root_ang_vel, obs + 10, 0);
    int c = 13;
    for (int j = 0; j < 12; ++j) { /* humanoid_amp_base.py:510-540 dof_to_obs */
        const int off = DOF_OFFSETS[j], size = DOF_OFFSETS[j + 1] - off;
        if (size == 3) {
            float e[3] = { dof_pos[off * dof_pos_stride], dof_pos[(off + 1) * dof_pos_stride],
                           dof_pos[(off + 2) * dof_pos_stride] };
            float q[4];
            exp_map_to_quat(e, q);
            quat_to_tan_norm(q, obs + c);
            c += 6;
        } else {
            obs[c] = dof_pos[off * dof_pos_stride];
            c += 1;
        }
    }
    for (int j = 0; j < 28; ++j) obs[c + j] = dof_vel[j * dof_vel_stride];
    c += 28;
    for (int k = 0; k < 4; ++k) {
        float d[3] = { key_pos[3 * k] - root_pos[0], key_pos[3 * k + 1] - root_pos[1], key_pos[3 * k + 2] - root_pos[2] };
        quat_rotate_s(heading_rot, d, obs + c + 3 * k, 0);
    }
}

int orc_amp_observations(const float *root_states, const float *dof_pos, const float *dof_vel,
                         const float *key_body_pos, int32_t local_root_obs, float *obs, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) {
        const float *r = root_states + i * 13;
        amp_obs_one(r, r + 3, r + 7, r + 10, dof_pos + i * 28, 1, dof_vel + i * 28, 1, key_body_pos + i * 12,
                    local_root_obs, obs + i * 105);
    }
    return 0;
}

int orc_motion_amp_obs(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                       int32_t local_root_obs, float *out, const int64_t *out_rows, int64_t out_row_stride,
                       int64_t out_col_offset)
{
    for (int64_t s = 0; s < n; ++s) {
        float rp[3], rr[4], dp[28], rv[3], rav[3], dv[28], kp[12];
        if (ids[s] < 0 || ids[s] >= t->num_motions) continue; /* a hole of a masked list */
        motion_state_one(t, ids[s], times[s], rp, rr, dp, rv, rav, dv, kp, NULL, NULL, NULL);
        float *o = out + (out_rows ? out_rows[s] : s) * out_row_stride + out_col_offset;
        amp_obs_one(rp, rr, rv, rav, dp, 1, dv, 1, kp, local_root_obs, o);
    }
    return 0;
}

/* HumanoidAMP.fetch_amp_obs_demo after its draws (tasks/humanoid_amp.py:156-171): ids tiled over the history
 * steps, times0 + (-dt * k) in fp64, get_motion_state, build_amp_observations; out [n, num_steps * 105] */
int orc_motion_amp_obs_steps(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                             int32_t num_steps, double step_dt, int32_t local_root_obs, float *out)
{
    for (int64_t i = 0; i < n; ++i)
        for (int k = 0; k < num_steps; ++k) {
            float rp[3], rr[4], dp[28], rv[3], rav[3], dv[28], kp[12];
            if (ids[i] < 0 || ids[i] >= t->num_motions) continue;
            const double time = times[i] + (-step_dt * (double)k);
            motion_state_one(t, ids[i], time, rp, rr, dp, rv, rav, dv, kp, NULL, NULL, NULL);
            amp_obs_one(rp, rr, rv, rav, dp, 1, dv, 1, kp, local_root_obs, out + (i * num_steps + k) * 105);
        }
    return 0;
}

/* observation of environment e from the simulator tensors */
static void amp_env_obs(const igx_amp_params *p, const igx_amp_buffers *b, int64_t e, float obs[105])
{
    const float *r = b->root_states + e * 13;
    const float *ds = b->dof_state + e * 56;
    float key[12];
    for (int k = 0; k < 4; ++k)
        for (int i = 0; i < 3; ++i) key[3 * k + i] = b->rigid_body_state[(e * 15 + p->key_body_ids[k]) * 13 + i];
    amp_obs_one(r, r + 3, r + 7, r + 10, ds, 2, ds + 1, 2, key, p->local_root_obs, obs);
}

/* tasks/humanoid_amp.py:125-134 over tasks/amp/humanoid_amp_base.py:424-438, :584-609 */
int orc_humanoid_amp_step(const igx_amp_params *p, const igx_amp_buffers *b, int64_t n)
{
    const int K = p->num_amp_obs_steps;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        int64_t prog = b->progress[e] + 1;
        float obs[105];
        amp_env_obs(p, b, e, obs);
        memcpy(b->obs + e * 105, obs, sizeof obs);
        b->rew[e] = 1.0f;
        int64_t terminated = 0;
        if (p->enable_early_termination) {
            int fall_contact = 0, fall_height = 0;
            for (int body = 0; body < 15; ++body) {
                int is_contact_body = 0;
                for (int c = 0; c < p->num_contact_bodies; ++c) is_contact_body |= p->contact_body_ids[c] == body;
                if (is_contact_body) continue;
                const float *cf = b->contact_forces + (e * 15 + body) * 3;
                if (cf[0] > 0.1f || cf[1] > 0.1f || cf[2] > 0.1f) fall_contact = 1;
                if (b->rigid_body_state[(e * 15 + body) * 13 + 2] < p->termination_height) fall_height = 1;
            }
            if (fall_contact && fall_height && prog > 1) terminated = 1;
        }
        b->reset[e] = ((float)prog >= p->max_episode_length_m1) ? 1 : terminated;
        b->terminate[e] = terminated;
        b->progress[e] = prog;
        if (b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && b->reset[e] != 0) ? 1 : 0;
        if (b->obs_clamped) clamp_row(b->obs + e * 105, b->obs_clamped + e * 105, 105, p->clip_obs);
        float *a = b->amp_obs + e * K * 105;
        for (int k = K - 2; k >= 0; --k) memcpy(a + (k + 1) * 105, a + k * 105, 105 * sizeof(float));
        memcpy(a, obs, sizeof obs);
    }
    return 0;
}

/* tasks/humanoid_amp.py:184-204, :296-310, :274-277 (see include/igx.h) */
/* Sampling of reference states for the flagged environments with the library's Philox stream:
 * tasks/amp/utils_amp/motion_lib.py:61-65 (np.random.choice(m, p=weights): first index whose cumulative
 * weight exceeds the draw) and :67-78 (uniform phase * motion length). */
int orc_motion_sample_flagged(const igx_motion_tables *t, const igx_motion_sampler *s, const int64_t *reset, int64_t n,
                              int64_t *env_ids, int64_t *motion_ids, double *motion_times, int64_t *default_env_ids)
{
    for (int64_t e = 0; e < n; ++e) {
        int64_t id = -1, env = -1, denv = -1;
        double time = 0.0;
        if (reset[e] != 0 && !(orc_philox_uniform(s->seed, s->rng_offset, e + s->env_offset, 2) < s->ref_init_prob)) {
            denv = e; /* tasks/humanoid_amp.py:248-261: this one takes the default pose */
        } else if (reset[e] != 0) {
            double pick = (double)orc_philox_uniform(s->seed, s->rng_offset, e + s->env_offset, 0);
            id = 0;
            while (id < t->num_motions - 1 && !(pick < s->cdf[id])) ++id;
            time = s->random_time ? (double)orc_philox_uniform(s->seed, s->rng_offset, e + s->env_offset, 1) * t->motion_length[id] : 0.0;
            env = e;
        }
        env_ids[e] = env;
        motion_ids[e] = id;
        motion_times[e] = time;
        if (default_env_ids) default_env_ids[e] = denv;
    }
    return 0;
}

/* MotionLib.sample_motions / sample_time (tasks/amp/utils_amp/motion_lib.py:61-78) for n samples from the
 * library's Philox stream: u0 picks the motion by cumulative weight, u1 is the phase */
int orc_motion_sample(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int64_t *motion_ids,
                      double *motion_times)
{
    for (int64_t e = 0; e < n; ++e) {
        const double pick = (double)orc_philox_uniform(s->seed, s->rng_offset, e + s->env_offset, 0);
        int64_t id = 0;
        while (id < t->num_motions - 1 && !(pick < s->cdf[id])) ++id;
        motion_ids[e] = id;
        motion_times[e] = s->random_time ? (double)orc_philox_uniform(s->seed, s->rng_offset, e + s->env_offset, 1) * t->motion_length[id] : 0.0;
    }
    return 0;
}

int orc_humanoid_amp_reset(const igx_amp_params *p, const igx_amp_buffers *b, const int64_t *env_ids, int64_t m,
                           const igx_motion_state_out *s)
{
    const int K = p->num_amp_obs_steps;
    for (int64_t i = 0; i < m; ++i) {
        const int64_t e = env_ids[i];
        if (e < 0) continue; /* a hole of a masked list */
        float *r = b->root_states + e * 13;
        if (s->root_pos) memcpy(r, s->root_pos + i * 3, 12);
        if (s->root_rot) memcpy(r + 3, s->root_rot + i * 4, 16);
        if (s->root_vel) memcpy(r + 7, s->root_vel + i * 3, 12);
        if (s->root_ang_vel) memcpy(r + 10, s->root_ang_vel + i * 3, 12);
        for (int j = 0; j < 28; ++j) {
            if (s->dof_pos) b->dof_state[(e * 28 + j) * 2] = s->dof_pos[i * 28 + j];
            if (s->dof_vel) b->dof_state[(e * 28 + j) * 2 + 1] = s->dof_vel[i * 28 + j];
        }
        b->progress[e] = 0;
        b->reset[e] = 0;
        b->terminate[e] = 0;
        float obs[105];
        amp_env_obs(p, b, e, obs);
        memcpy(b->obs + e * 105, obs, sizeof obs);
        for (int k = 0; k < K; ++k) memcpy(b->amp_obs + (e * K + k) * 105, obs, sizeof obs);
    }
    return 0;
}

/* ------------------------------------------------------- episode statistics --- */
/* learning/common_agent.py:284-296: current_rewards += rewards; current_lengths += 1;
 * totals over done envs; current_* *= (1 - done).  out = {n_done, sum_reward, sum_length}. */
int orc_episode_stats_step(const igx_episode_buffers *b, int64_t n, double out[3])
{
    int64_t cnt = 0;
    double sr = 0.0, sl = 0.0;
    for (int64_t e = 0; e < n; ++e) {
        const float not_done = 1.0f - (b->done[e] != 0 ? 1.0f : 0.0f);
        const float cr = b->current_rewards[e] + b->rew[e];
        const float cl = b->current_lengths[e] + 1.0f;
        if (not_done == 0.0f) {
            cnt += 1;
            sr += (double)cr;
            sl += (double)cl;
        }
        b->current_rewards[e] = cr * not_done;
        b->current_lengths[e] = cl * not_done;
    }
    out[0] = (double)cnt; out[1] = sr; out[2] = sl;
    return 0;
}

/* ------------------------------------------------------------------ AntImitate --- */
static int cmp_double(const void *a, const void *b)
{
    double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

/* torch.quantile(x, q), 'linear': sort, ranks = q * (n - 1), lerp(below, above, frac) */
double orc_quantile_f64(const double *x, int64_t n, double q)
{
    for (int64_t i = 0; i < n; ++i)
        if (x[i] != x[i]) return NAN;  /* torch.quantile propagates NaN */
    double *s = (double *)malloc(sizeof(double) * (size_t)n);
    memcpy(s, x, sizeof(double) * (size_t)n);
    qsort(s, (size_t)n, sizeof(double), cmp_double);
    double ranks = q * (double)(n - 1);
    int64_t below = (int64_t)ranks, above = (int64_t)ceil(ranks);
    double w = ranks - (double)below, a = s[below], b = s[above], diff = b - a;
    free(s);
    /* at::native lerp as the CPU kernel evaluates it: one fused multiply-add onto the nearer end */
    return (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
}

/* The same quantile as a radix select, in the shape the device runs it (and shards it): the two
 * neighbouring order statistics are found most significant digit first, 6 passes of 9 + 5 x 11 bits over
 * an order-preserving integer image of the doubles; every pass first picks, from the complete histogram
 * of the pass before, the bin that holds each rank.  Shards all-reduce (sum) histogram `pass` between
 * orc_quantile_histogram(pass) and the next call.  Workspace layout as in include/igx.h: eight 64-byte
 * states, then per pass two histograms of 2048 uint64 at IGX_QUANTILE_HIST_OFFSET(pass). */
typedef struct { uint64_t prefix[2], rank[2]; double weight, result; uint64_t n_total, pad; } quantile_state;

static int q_bits(int pass) { return pass == 0 ? 9 : 11; }
static int q_shift(int pass) { return pass == 0 ? 55 : 55 - 11 * pass; }
static uint64_t q_key(double x)
{
    uint64_t u;
    memcpy(&u, &x, 8);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
static double q_value(uint64_t k)
{
    uint64_t u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    double x;
    memcpy(&x, &u, 8);
    return x;
}
static uint64_t *q_hist(void *ws, int pass) { return (uint64_t *)((char *)ws + IGX_QUANTILE_HIST_OFFSET(pass)); }

static void q_advance(const quantile_state *in, const uint64_t *hist, int pass, quantile_state *out)
{
    *out = *in;
    const int bits = q_bits(pass), nb = 1 << bits;
    const int same = in->prefix[0] == in->prefix[1];
    for (int r = 0; r < 2; ++r) {
        const uint64_t *h = hist + ((r == 1 && !same) ? 2048 : 0);
        uint64_t run = 0, rank = in->rank[r];
        for (int bin = 0; bin < nb; ++bin) {
            if (rank >= run && rank < run + h[bin]) {
                out->prefix[r] = (in->prefix[r] << bits) | (uint64_t)bin;
                out->rank[r] = rank - run;
            }
            run += h[bin];
        }
    }
}

int orc_quantile_begin(void *ws, double q, int64_t n_total)
{
    quantile_state *st = (quantile_state *)ws;
    memset(q_hist(ws, 0), 0, (size_t)(IGX_QUANTILE_HIST_OFFSET(IGX_QUANTILE_PASSES) - IGX_QUANTILE_HIST_OFFSET(0)));
    double ranks = q * (double)(n_total - 1);
    int64_t below = (int64_t)ranks, above = (int64_t)ceil(ranks);
    memset(&st[0], 0, sizeof st[0]);
    st[0].rank[0] = (uint64_t)below;
    st[0].rank[1] = (uint64_t)above;
    st[0].weight = ranks - (double)below;
    st[0].n_total = (uint64_t)n_total;
    return 0;
}

int orc_quantile_histogram(const double *x, int64_t n, void *ws, int32_t pass)
{
    quantile_state *st = (quantile_state *)ws;
    if (pass > 0) q_advance(&st[pass - 1], q_hist(ws, pass - 1), pass - 1, &st[pass]);
    const quantile_state *cur = &st[pass];
    uint64_t *hist = q_hist(ws, pass);
    const int shift = q_shift(pass), bits = q_bits(pass);
    const int same = cur->prefix[0] == cur->prefix[1];
    for (int64_t i = 0; i < n; ++i) {
        if (pass == 0 && x[i] != x[i]) hist[2047] += 1;  /* spare bin of pass 0: the NaN count */
        uint64_t k = q_key(x[i]);
        uint64_t hi = pass == 0 ? 0 : (k >> (shift + bits));
        unsigned d = (unsigned)((k >> shift) & ((1u << bits) - 1u));
        if (hi == cur->prefix[0]) hist[d] += 1;
        if (!same && hi == cur->prefix[1]) hist[2048 + d] += 1;
    }
    return 0;
}

int orc_quantile_finish(void *ws, double *out)
{
    quantile_state *st = (quantile_state *)ws;
    const int last = IGX_QUANTILE_PASSES - 1;
    q_advance(&st[last], q_hist(ws, last), last, &st[last + 1]);
    double a = q_value(st[last + 1].prefix[0]), b = q_value(st[last + 1].prefix[1]), w = st[last + 1].weight;
    double diff = b - a, r = (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
    if (q_hist(ws, 0)[2047] != 0) r = NAN;
    st[last + 1].result = r;
    st[0].result = r;
    if (out) *out = r;
    return 0;
}

/* The quantile in two collectives (igx_quantile_window_* of include/igx.h), restated serially: the shards are
 * counted over the bins of a shared window of the key space, the all-reduced histogram names the bin that holds
 * rank[0]; a bin of at most `capacity` keys is exchanged and sorted, a fuller one becomes the next window. */
typedef struct {
    uint64_t lo, rank[2];
    double weight;
    uint64_t n_total, below;
    double result;
    uint32_t shift, magic, bin, count, rounds, status, nan, inside, capacity, pad[9];
} window_state;
#define QW_MAGIC 0x51574e44u
#define QW_SHIFT_MAX 50u
#define QW_NONE 0xffffffffu
#define QW_BAD_GATHER 5u
#define QW_INTERIOR ((uint64_t)(IGX_QUANTILE_WINDOW_BINS - 2))

static uint32_t *qw_hist(void *ws) { return (uint32_t *)((char *)ws + IGX_QUANTILE_WINDOW_HIST_OFFSET); }
static uint32_t qw_bin(uint64_t k, uint64_t lo, uint32_t shift)
{
    if (k < lo) return 0;
    uint64_t d = (k - lo) >> shift;
    return d >= QW_INTERIOR ? (uint32_t)(IGX_QUANTILE_WINDOW_BINS - 1) : (uint32_t)d + 1;
}
static uint32_t qw_shift_for(uint64_t span)
{
    uint32_t s = 0;
    while (s < QW_SHIFT_MAX && (QW_INTERIOR << s) < span) ++s;
    return s;
}

int orc_quantile_window_begin(void *ws, double q, int64_t n_total, int32_t reset)
{
    window_state *st = (window_state *)ws, keep = *st;
    memset(qw_hist(ws), 0, 4 * IGX_QUANTILE_WINDOW_HIST_WORDS);
    memset(st, 0, sizeof *st);
    int fresh = reset || keep.magic != QW_MAGIC || keep.shift > QW_SHIFT_MAX;
    st->lo = fresh ? 0 : keep.lo;
    st->shift = fresh ? QW_SHIFT_MAX : keep.shift;
    double ranks = q * (double)(n_total - 1);
    int64_t below = (int64_t)ranks;
    st->rank[0] = (uint64_t)below;
    st->rank[1] = (uint64_t)(int64_t)ceil(ranks);
    st->weight = ranks - (double)below;
    st->n_total = (uint64_t)n_total;
    st->magic = QW_MAGIC;
    st->bin = QW_NONE;
    st->status = IGX_QW_REFINED;
    return 0;
}

int orc_quantile_window_histogram(const double *x, int64_t n, void *ws)
{
    const window_state *st = (const window_state *)ws;
    uint32_t *hist = qw_hist(ws);
    for (int64_t i = 0; i < n; ++i) {
        if (x[i] != x[i]) hist[IGX_QUANTILE_WINDOW_BINS] += 1;
        hist[qw_bin(q_key(x[i]), st->lo, st->shift)] += 1;
    }
    return 0;
}

int orc_quantile_window_locate(void *ws, int32_t capacity, int64_t *decision)
{
    window_state *st = (window_state *)ws;
    uint32_t *hist = qw_hist(ws);
    uint64_t run = 0;
    uint32_t bin = QW_NONE;
    for (uint32_t b = 0; b < IGX_QUANTILE_WINDOW_BINS; ++b) {
        if (st->rank[0] >= run && st->rank[0] < run + hist[b]) { bin = b; break; }
        run += hist[b];
    }
    uint32_t count = bin == QW_NONE ? 0 : hist[bin];
    st->rounds += 1;
    st->bin = bin;
    st->count = count;
    st->below = bin == QW_NONE ? 0 : run;
    st->inside = (bin != QW_NONE && st->rank[1] - st->below < count) ? 1 : 0;
    st->nan |= hist[IGX_QUANTILE_WINDOW_BINS] != 0;
    st->capacity = (uint32_t)capacity;
    int interior = bin >= 1 && bin <= IGX_QUANTILE_WINDOW_BINS - 2;
    if (bin == QW_NONE) st->status = IGX_QW_GIVE_UP;
    else if (count <= (uint32_t)capacity) st->status = IGX_QW_COLLECT;
    else if (interior && st->shift == 0) st->status = IGX_QW_VALUE;
    else if (st->rounds >= IGX_QUANTILE_WINDOW_MAX_ROUNDS) st->status = IGX_QW_GIVE_UP;
    else {
        st->status = IGX_QW_REFINED;
        if (interior) {                        /* the bin becomes the window */
            st->lo += (uint64_t)(bin - 1) << st->shift;
            st->shift = st->shift > 13 ? st->shift - 13 : 0;
        } else if (bin == 0) {                 /* everything below the window */
            st->shift = qw_shift_for(st->lo);
            st->lo = 0;
        } else {                               /* everything above it */
            st->lo += QW_INTERIOR << st->shift;
            st->shift = qw_shift_for(0 - st->lo);
        }
        memset(hist, 0, 4 * IGX_QUANTILE_WINDOW_HIST_WORDS);
    }
    if (st->status == IGX_QW_GIVE_UP) st->magic = 0;
    decision[0] = st->status;
    decision[1] = st->status == IGX_QW_COLLECT ? count : 0;
    decision[2] = bin;
    decision[3] = st->rounds;
    return 0;
}

/* slots[0] = keys of this shard in the located bin, slots[1] = its smallest key above the bin, then the keys
 * (in the order of x; the device writes them in any order - the exchange sorts them) */
int orc_quantile_window_collect(const double *x, int64_t n, const void *ws, uint64_t *slots, int64_t num_slots)
{
    const window_state *st = (const window_state *)ws;
    slots[0] = 0;
    slots[1] = ~0ull;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t k = q_key(x[i]);
        uint32_t b = qw_bin(k, st->lo, st->shift);
        if (b == st->bin) {
            if (st->status == IGX_QW_COLLECT) {
                if ((int64_t)slots[0] + 2 < num_slots) slots[2 + slots[0]] = k;
                slots[0] += 1;
            }
        } else if (b > st->bin && k < slots[1]) slots[1] = k;
    }
    return 0;
}

static int qw_cmp(const void *a, const void *b)
{
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return x < y ? -1 : x > y;
}

int orc_quantile_window_finish(void *ws, const uint64_t *gathered, int32_t num_ranks, int64_t num_slots, double *out)
{
    window_state *st = (window_state *)ws;
    uint64_t keys[IGX_QUANTILE_WINDOW_MAX_CANDIDATES], total = 0, above = ~0ull;
    int ok = st->status == IGX_QW_COLLECT || st->status == IGX_QW_VALUE;
    for (int r = 0; r < num_ranks; ++r) {
        const uint64_t *g = gathered + (int64_t)r * num_slots;
        if ((int64_t)g[0] + 2 > num_slots) ok = 0;
        for (uint64_t j = 0; ok && j < g[0]; ++j)
            if (total + j < IGX_QUANTILE_WINDOW_MAX_CANDIDATES) keys[total + j] = g[2 + j];
        total += g[0];
        if (g[1] < above) above = g[1];
    }
    if (total != (st->status == IGX_QW_COLLECT ? st->count : 0u) || total > IGX_QUANTILE_WINDOW_MAX_CANDIDATES) ok = 0;
    uint64_t k0 = 0, k1 = 0;
    if (ok && st->status == IGX_QW_VALUE) {
        k0 = st->lo + (uint64_t)(st->bin - 1);
        k1 = (st->rank[1] == st->rank[0] || st->inside) ? k0 : above;
    } else if (ok) {
        qsort(keys, (size_t)total, 8, qw_cmp);
        uint64_t i0 = st->rank[0] - st->below;
        k0 = keys[i0];
        k1 = st->rank[1] == st->rank[0] ? k0 : (st->inside ? keys[i0 + 1] : above);
    }
    double a = q_value(k0), b = q_value(k1), w = st->weight, diff = b - a;
    double r = (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
    if (st->nan || !ok) r = NAN;
    st->result = r;
    st->rounds = 0;
    if (!ok) {
        st->status = QW_BAD_GATHER;
        st->magic = 0;
    } else {
        if (st->status == IGX_QW_COLLECT) {
            if (st->count > st->capacity / 2 && st->shift > 0) st->shift -= 1;
            else if (st->count < st->capacity / 64 && st->shift < QW_SHIFT_MAX) st->shift += 1;
        }
        uint64_t half = (uint64_t)(IGX_QUANTILE_WINDOW_BINS / 2 - 1) << st->shift;
        st->lo = k0 > half ? k0 - half : 0;
    }
    if (out) *out = r;
    return 0;
}

int orc_ant_imitate_observe(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im, int64_t n)
{
    return loco_step(p, b, n, 0, im);
}

/* tasks/ant_imitate.py:523-555 given the global threshold */
int orc_ant_imitate_finish(const igx_loco_params *p, const igx_loco_buffers *b, const igx_imitate_args *im, int64_t n,
                           double quantile_value)
{
    /* Python's max(1.8, quantile): the second argument only if it compares greater (a NaN quantile leaves 1.8) */
    double thr = quantile_value > im->distance_floor ? quantile_value : im->distance_floor;
    for (int64_t e = 0; e < n; ++e) {
        int64_t idx = im->target_pose_idx[e];
        if (im->pose_diff[e] > thr) {
            b->reset[e] = 1;
            b->rew[e] = (float)im->death_cost;
            idx = 0;
        }
        if (idx >= im->num_frames - 3) {
            b->rew[e] = (float)(-3.0 * im->death_cost);
            idx = 0;
        }
        im->target_pose_idx[e] = idx;
        if (b->time_outs) /* vec_task.py:404 on the flags as they are now */
            b->time_outs[e] = ((float)b->progress[e] >= p->max_episode_length_m1 && b->reset[e] != 0) ? 1 : 0;
    }
    return 0;
}

/* ------------------------------------------------------------- HumanoidImitate --- */
/* tasks/humanoid_imitate.py:634-643 (post_physics_step), :579-605 (reset_idx), :795-831
 * (observations), :672-758 (reward; fp64 where the fp64 motion frame enters).  The 63- and
 * 66-term sums are sequential here; the reward is compared at 1e-5, the flags exactly. */
int orc_humanoid_imitate_step(const igx_humanoid_imitate_params *p, const igx_humanoid_imitate_buffers *b, int64_t n)
{
    const int D = IGX_HIM_DOF, NB = IGX_HIM_BODIES, NO = IGX_HIM_OBS;
    double tgt_pose[IGX_HIM_DOF], tgt_pos[IGX_HIM_BODIES * 3];
    for (int j = 0; j < D; ++j) {
        double lo = (double)p->dof_limits_lower[j], hi = (double)p->dof_limits_upper[j];
        double span = (double)(float)(p->dof_limits_upper[j] - p->dof_limits_lower[j]);
        tgt_pose[j] = (2.0 * b->motion_frame[j] - hi - lo) / span;
    }
    for (int k = 0; k < NB * 3; ++k) tgt_pos[k] = b->motion_frame[D + (k / 3) * 13 + k % 3];
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        float *ds = b->dof_state + e * D * 2;
        float *obs = b->obs + e * NO;
        int64_t prog = b->progress[e] + 1;
        int64_t rst = b->reset[e];
        if (b->randomize) b->randomize[e] += 1;
        if (rst != 0) {
            for (int j = 0; j < D; ++j) {
                float up = draw(b->rand, e, 2 * D, j, p->seed, p->rng_offset, p->env_offset);
                float uv = draw(b->rand, e, 2 * D, D + j, p->seed, p->rng_offset, p->env_offset);
                float t = b->initial_dof_pos[e * D + j] + (0.4f * up + (-0.2f));
                t = torch_min(t, p->dof_limits_upper[j]);
                t = torch_max(t, p->dof_limits_lower[j]);
                ds[2 * j] = t;
                ds[2 * j + 1] = 0.2f * uv + (-0.1f);
            }
            const float *ir = b->initial_root_states + e * 13;
            float tt[3] = { p->targets[0] - ir[0], p->targets[1] - ir[1], 0.0f };
            float pot = -norm3(tt) / p->dt;
            b->potentials[e] = pot;
            b->prev_potentials[e] = pot;
            prog = 0;
            rst = 0;
        }
        for (int j = 0; j < D; ++j) {
            obs[j] = unscale1(ds[2 * j], p->dof_limits_lower[j], p->dof_limits_upper[j]);
            obs[D + j] = ds[2 * j + 1] * p->dof_vel_scale;
        }
        for (int c = 0; c < 13; ++c) obs[2 * D + c] = b->root_states[e * 13 + c];
        for (int k = 0; k < NB * 3; ++k) obs[2 * D + 13 + k] = b->rigid_body_states[(e * NB + k / 3) * 13 + k % 3];

        double pose = 0.0, body = 0.0;
        float vel = 0.0f, act = 0.0f;
        for (int j = 0; j < D; ++j) {
            pose += fabs((double)obs[j] - tgt_pose[j]);
            vel += fabsf(obs[D + j]);
            act += b->actions[e * D + j] * b->actions[e * D + j];
        }
        for (int k = 0; k < NB * 3; ++k) body += fabs((double)obs[2 * D + 13 + k] - tgt_pos[k]);
        float x = obs[126], y = obs[127], z = obs[128];
        double total = 0.0;
        total += exp(-0.2 * body);
        total += exp(-0.3 * pose);
        total += (double)expf(-0.05f * vel);
        total += (z - p->termination_height > 0.0f) ? (double)p->up_weight : 0.0;
        total += (double)(0.1f * expf(-sqrtf(x * x + y * y)));
        total += (double)(-(p->actions_cost_scale * act));
        if (z < p->termination_height) rst = 1;
        if (x * x + y * y > p->termination_radius_sq) rst = 1;
        if ((float)prog >= p->max_episode_length_m1) rst = 1;
        b->rew[e] = (float)total;
        b->reset[e] = rst;
        b->progress[e] = prog;
        if (b->time_outs) b->time_outs[e] = ((float)prog >= p->max_episode_length_m1 && rst != 0) ? 1 : 0;
        if (b->obs_clamped) clamp_row(b->obs + e * NO, b->obs_clamped + e * NO, NO, p->clip_obs);
    }
    return 0;
}

/* ------------------------------------------------- hand disturbance forces --- */
/* tasks/allegro_hand.py:532 (rb_forces[env_ids] = 0 in reset_idx), :558-559 (log-uniform re-draw of
 * random_force_prob on reset), :617-625 (decay, Bernoulli selection, randn * mass * force_scale into
 * the object's body rows); tasks/shadow_hand.py:616, :642-643, :700-708 are the same lines. */
int orc_hand_random_forces(const igx_hand_forces *f, int64_t n)
{
    if (!f || n < 0) return -1;
    const int nb = f->num_bodies;
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < n; ++e) {
        float u_prob, u_sel, z[3];
        if (f->rand) {
            const float *r = f->rand + e * IGX_HAND_FORCE_RAND_PER_ENV;
            u_prob = r[0]; u_sel = r[1]; z[0] = r[2]; z[1] = r[3]; z[2] = r[4];
        } else {  /* the library's own stream: uniforms 0..5 of the environment, Box-Muller on (2,3) and (4,5) */
            const float u2 = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 2), u3 = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 3);
            const float v0 = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 4), v1 = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 5);
            u_prob = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 0);
            u_sel = orc_philox_uniform(f->seed, f->rng_offset, e + f->env_offset, 1);
            const float r0 = sqrtf(-2.0f * logf(1.0f - u2)), a0 = 6.2831855f * u3;
            const float r1 = sqrtf(-2.0f * logf(1.0f - v0)), a1 = 6.2831855f * v1;
            z[0] = r0 * cosf(a0); z[1] = r0 * sinf(a0); z[2] = r1 * cosf(a1);
        }
        float *rows = f->rb_forces + e * nb * 3;
        if (f->reset[e] != 0) {
            for (int i = 0; i < nb * 3; ++i) rows[i] = 0.0f;
            f->random_force_prob[e] = expf((f->log_prob_lo - f->log_prob_hi) * u_prob + f->log_prob_hi);
        }
        for (int i = 0; i < nb * 3; ++i) rows[i] = rows[i] * f->decay;
        if (u_sel < f->random_force_prob[e])
            for (int c = 0; c < 3; ++c) rows[f->object_rb_handle * 3 + c] = (z[c] * f->object_rb_mass) * f->force_scale;
    }
    return 0;
}

/* --------------------------------------------- windows of finished episodes --- */
/* utils/rlgames_utils.py:159-174 (RLGPUAlgoObserver.process_infos): cumulative += value; every done
 * environment, in index order, appends its cumulative value to a deque(maxlen=games_to_track) and is
 * zeroed.  The deque is the ring described in igx.h; header = {finished_total, base, last_count, pad}. */
int orc_episode_window_step(const igx_episode_window *w, int64_t n)
{
    if (!w || n < 0 || w->games_to_track < 1) return -1;
    uint64_t *h = (uint64_t *)w->workspace;
    const uint64_t G = (uint64_t)w->games_to_track;
    uint64_t total = h[0];
    h[1] = total;
    for (int k = 0; k < w->num_keys; ++k)
        for (int64_t e = 0; e < n; ++e) w->cumulative[k][e] = w->cumulative[k][e] + w->values[k][e];
    uint64_t count = 0;
    for (int64_t e = 0; e < n; ++e) {
        if (w->done[e] == 0) continue;
        for (int k = 0; k < w->num_keys; ++k) {
            w->window[(uint64_t)k * G + (total % G)] = w->cumulative[k][e];  /* append: overwrites the oldest */
            w->cumulative[k][e] = 0.0f;
        }
        ++total;
        ++count;
    }
    h[0] = total;
    h[2] = count;
    return 0;
}
