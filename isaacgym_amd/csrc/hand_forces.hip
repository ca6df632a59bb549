// Random disturbance forces on the manipulated object of the in-hand tasks: the tensor part of
// AllegroHand / ShadowHand.pre_physics_step that surrounds apply_rigid_body_force_tensors.
//
// Reference (IsaacGymEnvs/isaacgymenvs/tasks/allegro_hand.py; shadow_hand.py has the same lines
// at :616, :642-643, :700-708), in the order pre_physics_step executes them:
//   reset_idx, for the flagged environments (:532, :558-559):
//       rb_forces[env, :, :] = 0
//       random_force_prob[env] = exp((log(lo) - log(hi)) * rand + log(hi))      log-uniform in [lo, hi]
//   then, when forceScale > 0 (:617-625):
//       rb_forces *= pow(force_decay, dt / force_decay_interval)
//       force_indices = (rand(N) < random_force_prob).nonzero()
//       rb_forces[force_indices, object_rb_handles, :] = randn * object_rb_masses * force_scale
// One thread per four consecutive floats of rb_forces; every value an element needs from its
// environment (flag, probability, draws) is recomputed by the element's thread, so there is no
// exchange between threads and no second launch.
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

struct ForceDraws { float u_prob, u_sel, z[3]; };

__device__ __forceinline__ ForceDraws force_draws(const igx_hand_forces &f, int64_t e)
{
    ForceDraws d;
    if (f.rand) {
        const float *r = f.rand + e * IGX_HAND_FORCE_RAND_PER_ENV;
        d.u_prob = r[0]; d.u_sel = r[1]; d.z[0] = r[2]; d.z[1] = r[3]; d.z[2] = r[4];
    } else {
        float u[4], v[4];
        const uint64_t off = rng_offset_of(f.rng_offset, f.rng_offset_dev);
        philox_uniform4(f.seed, off, e + f.env_offset, 0, u);
        philox_uniform4(f.seed, off, e + f.env_offset, 1, v);
        d.u_prob = u[0]; d.u_sel = u[1];
        // Box-Muller on (u2, u3) and (v0, v1): three of the four normals
        const float r0 = sqrtf(-2.0f * logf(1.0f - u[2])), a0 = 6.2831855f * u[3];
        const float r1 = sqrtf(-2.0f * logf(1.0f - v[0])), a1 = 6.2831855f * v[1];
        d.z[0] = r0 * cosf(a0); d.z[1] = r0 * sinf(a0); d.z[2] = r1 * cosf(a1);
    }
    return d;
}

// what environment e decides this step: its (possibly re-drawn) probability and whether a new force is applied
__device__ __forceinline__ void force_decision(const igx_hand_forces &f, int64_t e, const ForceDraws &d, bool &rst, float &prob,
                                               bool &selected)
{
    rst = f.reset[e] != 0;
    prob = rst ? expf((f.log_prob_lo - f.log_prob_hi) * d.u_prob + f.log_prob_hi) : f.random_force_prob[e];
    selected = d.u_sel < prob;
}

__global__ __launch_bounds__(256) void hand_forces_kernel(const igx_hand_forces f, int64_t n, int64_t total)
{
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 >= total) return;
    const int per_env = 3 * f.num_bodies;
    const bool vec = i0 + 4 <= total;
    float v[4];
    if (vec) {
        const v4f x = *reinterpret_cast<const v4f *>(f.rb_forces + i0);
        v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w;
    } else {
        for (int k = 0; k < 4; ++k) v[k] = i0 + k < total ? f.rb_forces[i0 + k] : 0.0f;
    }
    int64_t e = i0 / per_env;
    int r = (int)(i0 - e * per_env);
    int64_t e_cached = -1;
    bool rst = false, selected = false;
    float prob = 0.0f;
    ForceDraws d{};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (i0 + k < total) {
            if (e != e_cached) {  // at most two environments per thread
                d = force_draws(f, e);
                force_decision(f, e, d, rst, prob, selected);
                e_cached = e;
            }
            const int body = r / 3, comp = r - 3 * body;
            float x = rst ? 0.0f : v[k];
            x = x * f.decay;
            if (selected && body == f.object_rb_handle) x = (d.z[comp] * f.object_rb_mass) * f.force_scale;
            v[k] = x;
            if (r == 0 && rst) f.random_force_prob[e] = prob;  // the element that owns the environment's scalar
        }
        if (++r == per_env) {
            r = 0;
            ++e;
        }
    }
    if (vec) {
        *reinterpret_cast<v4f *>(f.rb_forces + i0) = v4f{v[0], v[1], v[2], v[3]};
    } else {
        for (int k = 0; k < 4; ++k)
            if (i0 + k < total) f.rb_forces[i0 + k] = v[k];
    }
}

}  // namespace igx

extern "C" int igx_hand_random_forces(const igx_hand_forces *f, int64_t n_envs, igx_stream_t stream)
{
    using namespace igx;
    const char *who = "igx_hand_random_forces";
    if (!f) return fail(IGX_E_NULL, "%s: argument struct is NULL", who);
    if (n_envs < 0 || n_envs > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: n_envs %lld out of range", who, (long long)n_envs);
    if (f->num_bodies < 1 || f->num_bodies > 4096) return fail(IGX_E_PARAM, "%s: num_bodies %d out of range", who, f->num_bodies);
    if (f->object_rb_handle < 0 || f->object_rb_handle >= f->num_bodies)
        return fail(IGX_E_PARAM, "%s: object_rb_handle %d outside [0, %d)", who, f->object_rb_handle, f->num_bodies);
    if (n_envs == 0) return IGX_OK;
    if (!f->rb_forces || !f->random_force_prob || !f->reset)
        return fail(IGX_E_NULL, "%s: rb_forces / random_force_prob / reset is NULL", who);
    if (!aligned16(f->rb_forces)) return fail(IGX_E_ALIGN, "%s: rb_forces must be 16-byte aligned", who);
    const int64_t total = n_envs * 3 * f->num_bodies;
    const int64_t blocks = (total + 1023) / 1024;
    if (blocks > 0x7fffffffLL) return fail(IGX_E_SIZE, "%s: too many elements for one launch", who);
    hipLaunchKernelGGL(hand_forces_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), *f, n_envs,
                       total);
    return launch_status(who);
}
