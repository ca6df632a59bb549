// Device-side math shared by all libigx kernels (gfx950).
//
// Numerics contract: the reference runs every ATen op as its own CPU kernel,
// so every intermediate is rounded to fp32 once per operation.  These helpers
// therefore use one IEEE operation per source operation (build with
// -ffp-contract=off; '/' and sqrtf are correctly rounded under hipcc's default
// -fhip-fp32-correctly-rounded-divide-sqrt) and an explicit fused multiply-add
// only where the ATen CPU kernel fuses (torch.cross, vector norm).
// Quaternions are xyzw.  File:line citations are into
// IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py of the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace igx {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

__device__ __forceinline__ float fma_rn(float a, float b, float c) { return __builtin_fmaf(a, b, c); }

// :41-62 quat_mul, 8-multiply form
__device__ __forceinline__ f4 quat_mul(f4 a, f4 b)
{
    float ww = (a.z + a.x) * (b.x + b.y);
    float yy = (a.w - a.y) * (b.w + b.z);
    float zz = (a.w + a.y) * (b.w - b.z);
    float xx = ww + yy + zz;
    float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
    f4 o;
    o.w = qq - ww + (a.z - a.y) * (b.y - b.z);
    o.x = qq - xx + (a.x + a.w) * (b.x + b.w);
    o.y = qq - yy + (a.w - a.x) * (b.y + b.z);
    o.z = qq - zz + (a.z + a.y) * (b.w - b.x);
    return o;
}

// :106-110
__device__ __forceinline__ f4 quat_conjugate(f4 a) { return f4{-a.x, -a.y, -a.z, a.w}; }

// torch.cross on CPU: one fused multiply-add per component
__device__ __forceinline__ f3 cross3(f3 a, f3 b)
{
    return f3{fma_rn(a.y, b.z, -(a.z * b.y)), fma_rn(a.z, b.x, -(a.x * b.z)), fma_rn(a.x, b.y, -(a.y * b.x))};
}

// torch.bmm([N,1,3],[N,3,1]) on CPU: unfused dot accumulated onto +0
__device__ __forceinline__ float dot3(f3 a, f3 b) { return ((0.0f + a.x * b.x) + a.y * b.y) + a.z * b.z; }

// torch.norm(p=2, dim=-1) of a 3-vector on CPU
__device__ __forceinline__ float norm3(f3 v)
{
    float acc = v.x * v.x;
    acc = fma_rn(v.y, v.y, acc);
    acc = fma_rn(v.z, v.z, acc);
    return sqrtf(acc);
}

__device__ __forceinline__ float norm4(f4 v)
{
    float acc = v.x * v.x;
    acc = fma_rn(v.y, v.y, acc);
    acc = fma_rn(v.z, v.z, acc);
    acc = fma_rn(v.w, v.w, acc);
    return sqrtf(acc);
}

// :65-67 normalize
__device__ __forceinline__ f3 normalize3(f3 v)
{
    float n = norm3(v);
    n = (n < 1e-9f) ? 1e-9f : n;
    return f3{v.x / n, v.y / n, v.z / n};
}

// :80-90 quat_rotate / :411-421 my_quat_rotate (INVERSE=false),
// :93-103 quat_rotate_inverse (INVERSE=true)
template <bool INVERSE>
__device__ __forceinline__ f3 quat_rotate(f4 q, f3 v)
{
    f3 qv{q.x, q.y, q.z};
    float s = 2.0f * (q.w * q.w) - 1.0f;
    f3 cr = cross3(qv, v);
    float d = dot3(qv, v);
    f3 a{v.x * s, v.y * s, v.z * s};
    f3 b{cr.x * q.w * 2.0f, cr.y * q.w * 2.0f, cr.z * q.w * 2.0f};
    f3 c{qv.x * d * 2.0f, qv.y * d * 2.0f, qv.z * d * 2.0f};
    if (INVERSE) return f3{a.x - b.x + c.x, a.y - b.y + c.y, a.z - b.z + c.z};
    return f3{a.x + b.x + c.x, a.y + b.y + c.y, a.z + b.z + c.z};
}

#define IGX_TWO_PI_F 6.2831855f

// torch.remainder(atan2(y, x), float32(2*pi)).  |atan2| <= pi, so fmod leaves the value alone and what is
// left of remainder is its sign fix-up: add 2*pi to a negative NON-ZERO value.  A non-zero result carries the
// sign of y (-pi behind y = -0 and a negative x included), so its own sign decides.  A zero result is left
// alone by the reference when its atan2 gave +-0 as well: y = +-0, or a negative y whose quotient y / x
// underflows to zero (y = -1e-40 against x = 1e30, or x = +Inf).  The one case in which a zero result must
// still become 2*pi is a library that rounds or flushes a tiny negative quotient to -0 where the reference's
// keeps the subnormal: that is decided on the quotient itself, one IEEE division, the same everywhere.
// NaN stays NaN.
__device__ __forceinline__ float atan2_mod_2pi(float y, float x)
{
    const float a = atan2f(y, x);
    const bool negative = (a < 0.0f) || (a == 0.0f && y < 0.0f && y / x != 0.0f);
    return negative ? a + IGX_TWO_PI_F : a;
}

// the same for asin(s) (the pitch of get_euler_xyz): negative exactly when s < 0
__device__ __forceinline__ float mod_2pi_signed_by(float a, float s) { return (s < 0.0f) ? a + IGX_TWO_PI_F : a; }

// :126-128
__device__ __forceinline__ float normalize_angle(float x) { return atan2f(sinf(x), cosf(x)); }

// :175-195 get_euler_xyz (roll and yaw; pitch is unused on this path)
__device__ __forceinline__ void euler_roll_yaw(f4 q, float &roll, float &yaw)
{
    float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
    float cosr_cosp = q.w * q.w - q.x * q.x - q.y * q.y + q.z * q.z;
    float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
    float cosy_cosp = q.w * q.w + q.x * q.x - q.y * q.y - q.z * q.z;
    roll = atan2_mod_2pi(sinr_cosp, cosr_cosp);
    yaw = atan2_mod_2pi(siny_cosp, cosy_cosp);
}

// :239-241 unscale
__device__ __forceinline__ float unscale(float x, float lo, float hi) { return (2.0f * x - hi - lo) / (hi - lo); }
// :234-236 scale
__device__ __forceinline__ float scale(float x, float lo, float hi) { return 0.5f * (x + 1.0f) * (hi - lo) + lo; }
// torch.min / torch.max of two tensors: a NaN on either side gives NaN (fminf / fmaxf would drop it)
__device__ __forceinline__ float torch_min(float a, float b) { return (a != a || b != b) ? a + b : (b < a ? b : a); }
__device__ __forceinline__ float torch_max(float a, float b) { return (a != a || b != b) ? a + b : (a < b ? b : a); }
// :228-230 tensor_clamp = torch.max(torch.min(t, hi), lo): a NaN target stays NaN
__device__ __forceinline__ float tensor_clamp(float t, float lo, float hi) { return torch_max(torch_min(t, hi), lo); }

// :65-67 normalize applied to a quaternion (quat_unit, :113-115)
__device__ __forceinline__ f4 quat_unit(f4 q)
{
    float n = norm4(q);
    n = (n < 1e-9f) ? 1e-9f : n;
    return f4{q.x / n, q.y / n, q.z / n, q.w / n};
}

// :118-123 quat_from_angle_axis: normalises the axis and the result
__device__ __forceinline__ f4 quat_from_angle_axis(float angle, f3 axis)
{
    const float theta = angle / 2.0f;
    const f3 a = normalize3(axis);
    const float s = sinf(theta);
    return quat_unit(f4{a.x * s, a.y * s, a.z * s, cosf(theta)});
}

// :570-587 exp_map_to_angle_axis + :589-593 exp_map_to_quat.  The reference
// divides by a possibly zero angle and masks the result afterwards; the select
// below reproduces the masked value without ever consuming the inf / nan.
__device__ __forceinline__ f4 exp_map_to_quat(f3 e)
{
    const float n = norm3(e);
    f3 axis{e.x / n, e.y / n, e.z / n};
    float angle = normalize_angle(n);
    const bool mask = angle > 1e-5f;
    angle = mask ? angle : 0.0f;
    axis = mask ? axis : f3{0.0f, 0.0f, 1.0f};
    return quat_from_angle_axis(angle, axis);
}

// :423-444 quat_to_angle_axis
__device__ __forceinline__ void quat_to_angle_axis(f4 q, float &angle, f3 &axis)
{
    const float sin_theta = sqrtf(1.0f - q.w * q.w);
    angle = normalize_angle(2.0f * acosf(q.w));
    axis = f3{q.x / sin_theta, q.y / sin_theta, q.z / sin_theta};
    const bool mask = sin_theta > 1e-5f;
    angle = mask ? angle : 0.0f;
    axis = mask ? axis : f3{0.0f, 0.0f, 1.0f};
}

// :454-461 quat_to_exp_map
__device__ __forceinline__ f3 quat_to_exp_map(f4 q)
{
    float angle;
    f3 axis;
    quat_to_angle_axis(q, angle, axis);
    return f3{angle * axis.x, angle * axis.y, angle * axis.z};
}

// :595-628 slerp; t may lie outside [0,1] (the motion library extrapolates)
__device__ __forceinline__ f4 slerp(f4 q0, f4 q1, float t)
{
    float c = q0.w * q1.w + q0.x * q1.x + q0.y * q1.y + q0.z * q1.z;
    if (c < 0.0f) q1 = f4{-q1.x, -q1.y, -q1.z, -q1.w};
    c = fabsf(c);
    const float half_theta = acosf(c);
    const float s = sqrtf(1.0f - c * c);
    const float ra = sinf((1.0f - t) * half_theta) / s;
    const float rb = sinf(t * half_theta) / s;
    f4 o{ra * q0.x + rb * q1.x, ra * q0.y + rb * q1.y, ra * q0.z + rb * q1.z, ra * q0.w + rb * q1.w};
    if (fabsf(s) < 0.001f) o = f4{0.5f * q0.x + 0.5f * q1.x, 0.5f * q0.y + 0.5f * q1.y, 0.5f * q0.z + 0.5f * q1.z,
                                  0.5f * q0.w + 0.5f * q1.w};
    if (fabsf(c) >= 1.0f) o = q0;
    return o;
}

// :630-641 calc_heading
__device__ __forceinline__ float calc_heading(f4 q)
{
    const f3 d = quat_rotate<false>(q, f3{1.0f, 0.0f, 0.0f});
    return atan2f(d.y, d.x);
}

// :656-667 calc_heading_quat_inv
__device__ __forceinline__ f4 calc_heading_quat_inv(f4 q)
{
    return quat_from_angle_axis(-calc_heading(q), f3{0.0f, 0.0f, 1.0f});
}

// :70-77 quat_apply: b + w * t + xyz x t with t = 2 (xyz x b)
__device__ __forceinline__ f3 quat_apply(f4 a, f3 b)
{
    const f3 xyz{a.x, a.y, a.z};
    const f3 c = cross3(xyz, b);
    const f3 t{c.x * 2.0f, c.y * 2.0f, c.z * 2.0f};
    const f3 u = cross3(xyz, t);
    return f3{b.x + a.w * t.x + u.x, b.y + a.w * t.y + u.y, b.z + a.w * t.z + u.z};
}

// :168-172 copysign(a, b) = |a| * sign(b); torch.sign gives 0 for +-0 and for NaN
__device__ __forceinline__ float copysign_sign(float a, float b)
{
    return fabsf(a) * (float)((int)(0.0f < b) - (int)(b < 0.0f));
}

// :184-187, :195 the pitch of get_euler_xyz
__device__ __forceinline__ float euler_pitch(f4 q)
{
    const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
    const float pitch = (fabsf(sinp) >= 1.0f) ? copysign_sign(1.5707964f, sinp) : asinf(sinp);
    return mod_2pi_signed_by(pitch, sinp);
}

// :198-212 quat_from_euler_xyz
__device__ __forceinline__ f4 quat_from_euler_xyz(float roll, float pitch, float yaw)
{
    const float cy = cosf(yaw * 0.5f), sy = sinf(yaw * 0.5f);
    const float cr = cosf(roll * 0.5f), sr = sinf(roll * 0.5f);
    const float cp = cosf(pitch * 0.5f), sp = sinf(pitch * 0.5f);
    return f4{cy * sr * cp - sy * cr * sp, cy * cr * sp + sy * sr * cp, sy * cr * cp - cy * sr * sp,
              cy * cr * cp + sy * sr * sp};
}

// :570-587 exp_map_to_angle_axis (masked like the reference's where)
__device__ __forceinline__ void exp_map_to_angle_axis(f3 e, float &angle, f3 &axis)
{
    const float n = norm3(e);
    axis = f3{e.x / n, e.y / n, e.z / n};
    angle = normalize_angle(n);
    const bool mask = angle > 1e-5f;
    angle = mask ? angle : 0.0f;
    axis = mask ? axis : f3{0.0f, 0.0f, 1.0f};
}

// torch.clamp(x, lo, hi): NaN stays NaN (lo <= hi)
__device__ __forceinline__ float clamp1(float x, float lo, float hi)
{
    x = x < lo ? lo : x;
    return x > hi ? hi : x;
}

// Philox4x32-10 counter-based generator (Salmon et al., SC'11).  Used for the
// reset draws when the caller supplies no pre-drawn uniforms.  Counter layout:
// (env lo, env hi, offset lo, offset hi[23:0] << 8 | block), key = seed.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1)
{
#pragma unroll
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

// Step counter of the Philox stream: the by-value offset plus an optional device-resident
// counter.  A recorded graph replays the same argument values; the counter, advanced on the
// stream between replays, gives every replay fresh draws.
__device__ __forceinline__ uint64_t rng_offset_of(uint64_t offset, const uint64_t *dev) { return dev ? offset + *dev : offset; }

// Uniforms 4*blk .. 4*blk+3 of environment `env` at step `offset`, in [0,1).
__device__ __forceinline__ void philox_uniform4(uint64_t seed, uint64_t offset, int64_t env, int blk, float (&u)[4])
{
    uint32_t c[4];
    c[0] = (uint32_t)(uint64_t)env;
    c[1] = (uint32_t)((uint64_t)env >> 32);
    c[2] = (uint32_t)offset;
    c[3] = (((uint32_t)(offset >> 32) & 0xFFFFFFu) << 8) | (uint32_t)blk;
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int i = 0; i < 4; ++i) u[i] = (float)(c[i] >> 8) * (1.0f / 16777216.0f);
}

}  // namespace igx
