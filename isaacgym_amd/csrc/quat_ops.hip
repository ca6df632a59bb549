// Batched torch_jit_utils math behind the C ABI: one kernel template, one
// small functor per reference function.
//
// Rows are short (1..6 floats), so every array of a 256-item tile is moved
// between global memory and LDS as one contiguous slab with coalesced accesses
// and each lane works on its own item out of LDS; [n,3] rows never turn into
// strided 4-byte global accesses.
//
// Reference: IsaacGymEnvs/isaacgymenvs/utils/torch_jit_utils.py (line numbers
// at each functor).
#include "igx_device.hpp"
#include "igx_host.hpp"

namespace igx {

constexpr int QTILE = 256;

// Functor interface: widths of up to three per-item input rows (0 = unused),
// width of the output row, and compute(in0, in1, in2, out) on LDS rows.
struct OpQuatMul {  // :41-62
    static constexpr int A = 4, B = 4, C = 0, O = 4;
    __device__ static void compute(const float *a, const float *b, const float *, float *o)
    {
        const f4 r = quat_mul(f4{a[0], a[1], a[2], a[3]}, f4{b[0], b[1], b[2], b[3]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
};
struct OpQuatConjugate {  // :106-110
    static constexpr int A = 4, B = 0, C = 0, O = 4;
    __device__ static void compute(const float *a, const float *, const float *, float *o)
    {
        o[0] = -a[0]; o[1] = -a[1]; o[2] = -a[2]; o[3] = a[3];
    }
};
template <bool INVERSE>
struct OpQuatRotate {  // :80-90 / :93-103
    static constexpr int A = 4, B = 3, C = 0, O = 3;
    __device__ static void compute(const float *q, const float *v, const float *, float *o)
    {
        const f3 r = quat_rotate<INVERSE>(f4{q[0], q[1], q[2], q[3]}, f3{v[0], v[1], v[2]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    }
};
struct OpSlerp {  // :595-628
    static constexpr int A = 4, B = 4, C = 1, O = 4;
    __device__ static void compute(const float *a, const float *b, const float *t, float *o)
    {
        const f4 r = slerp(f4{a[0], a[1], a[2], a[3]}, f4{b[0], b[1], b[2], b[3]}, t[0]);
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
};
struct OpQuatFromAngleAxis {  // :118-123
    static constexpr int A = 1, B = 3, C = 0, O = 4;
    __device__ static void compute(const float *ang, const float *ax, const float *, float *o)
    {
        const f4 r = quat_from_angle_axis(ang[0], f3{ax[0], ax[1], ax[2]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
};
struct OpNormalizeAngle {  // :126-128
    static constexpr int A = 1, B = 0, C = 0, O = 1;
    __device__ static void compute(const float *x, const float *, const float *, float *o) { o[0] = normalize_angle(x[0]); }
};
struct OpQuatToExpMap {  // :454-461
    static constexpr int A = 4, B = 0, C = 0, O = 3;
    __device__ static void compute(const float *q, const float *, const float *, float *o)
    {
        const f3 r = quat_to_exp_map(f4{q[0], q[1], q[2], q[3]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z;
    }
};
struct OpExpMapToQuat {  // :589-593
    static constexpr int A = 3, B = 0, C = 0, O = 4;
    __device__ static void compute(const float *e, const float *, const float *, float *o)
    {
        const f4 r = exp_map_to_quat(f3{e[0], e[1], e[2]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
};
struct OpQuatToTanNorm {  // :548-561
    static constexpr int A = 4, B = 0, C = 0, O = 6;
    __device__ static void compute(const float *q, const float *, const float *, float *o)
    {
        const f4 qq{q[0], q[1], q[2], q[3]};
        const f3 t = quat_rotate<false>(qq, f3{1.0f, 0.0f, 0.0f});
        const f3 n = quat_rotate<false>(qq, f3{0.0f, 0.0f, 1.0f});
        o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = n.x; o[4] = n.y; o[5] = n.z;
    }
};
struct OpCalcHeadingQuatInv {  // :656-667
    static constexpr int A = 4, B = 0, C = 0, O = 4;
    __device__ static void compute(const float *q, const float *, const float *, float *o)
    {
        const f4 r = calc_heading_quat_inv(f4{q[0], q[1], q[2], q[3]});
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
    }
};

template <int W>
__device__ __forceinline__ void slab_in(const float *__restrict__ g, float *s, int64_t base, int nvalid, int tid)
{
    if (W == 0) return;
    const float *src = g + base * W;
    const int count = nvalid * W;
    if (nvalid == QTILE && (W * QTILE) % 4 == 0 && ((reinterpret_cast<uintptr_t>(src) & 15u) == 0)) {
        using v4f = float __attribute__((ext_vector_type(4)));
        for (int k = tid; k < count / 4; k += QTILE)
            *reinterpret_cast<v4f *>(s + 4 * k) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + k);
    } else {
        for (int k = tid; k < count; k += QTILE) s[k] = src[k];
    }
}

template <class Op>
__global__ __launch_bounds__(QTILE) void rowwise_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        const float *__restrict__ c, float *__restrict__ out, int64_t n)
{
    __shared__ __align__(16) float sa[QTILE * (Op::A ? Op::A : 1)];
    __shared__ __align__(16) float sb[QTILE * (Op::B ? Op::B : 1)];
    __shared__ __align__(16) float sc[QTILE * (Op::C ? Op::C : 1)];
    __shared__ __align__(16) float so[QTILE * Op::O];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * QTILE;
    const int nvalid = (n - base >= QTILE) ? QTILE : (int)(n - base);
    slab_in<Op::A>(a, sa, base, nvalid, tid);
    slab_in<Op::B>(b, sb, base, nvalid, tid);
    slab_in<Op::C>(c, sc, base, nvalid, tid);
    __syncthreads();
    if (tid < nvalid) Op::compute(sa + tid * Op::A, sb + tid * Op::B, sc + tid * Op::C, so + tid * Op::O);
    __syncthreads();
    float *dst = out + base * Op::O;
    const int count = nvalid * Op::O;
    if (nvalid == QTILE && (Op::O * QTILE) % 4 == 0 && ((reinterpret_cast<uintptr_t>(dst) & 15u) == 0)) {
        using v4f = float __attribute__((ext_vector_type(4)));
        for (int k = tid; k < count / 4; k += QTILE) reinterpret_cast<v4f *>(dst)[k] = *reinterpret_cast<const v4f *>(so + 4 * k);
    } else {
        for (int k = tid; k < count; k += QTILE) dst[k] = so[k];
    }
}

template <class Op>
static int run_rowwise(const char *who, const float *a, const float *b, const float *c, float *out, int64_t n,
                       igx_stream_t stream)
{
    if (n < 0 || n > ((int64_t)1 << 38)) return fail(IGX_E_SIZE, "%s: n %lld out of range", who, (long long)n);
    if (n == 0) return IGX_OK;
    if (!a || (Op::B && !b) || (Op::C && !c) || !out) return fail(IGX_E_NULL, "%s: NULL argument", who);
    const int64_t blocks = (n + QTILE - 1) / QTILE;
    hipLaunchKernelGGL((rowwise_kernel<Op>), dim3((unsigned)blocks), dim3(QTILE), 0, static_cast<hipStream_t>(stream), a,
                       b, c, out, n);
    return launch_status(who);
}

// scale / unscale / tensor_clamp: x [n,d] against per-column lower / upper [d]
enum { AFF_SCALE = 0, AFF_UNSCALE = 1, AFF_CLAMP = 2 };

template <int KIND>
__global__ __launch_bounds__(256) void affine_kernel(const float *__restrict__ x, const float *__restrict__ lower,
                                                     const float *__restrict__ upper, float *__restrict__ out,
                                                     int64_t total, int d)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % d);
    const float lo = lower[col], hi = upper[col], v = x[i];
    float r;
    if (KIND == AFF_SCALE) r = scale(v, lo, hi);            // :234-236
    else if (KIND == AFF_UNSCALE) r = unscale(v, lo, hi);   // :239-241
    else r = tensor_clamp(v, lo, hi);                       // :228-230
    out[i] = r;
}

template <int KIND>
static int run_affine(const char *who, const float *x, const float *lower, const float *upper, float *out, int64_t n,
                      int32_t d, igx_stream_t stream)
{
    if (n < 0 || d <= 0 || n > ((int64_t)1 << 38) / d) return fail(IGX_E_SIZE, "%s: bad shape [%lld,%d]", who, (long long)n, d);
    if (n == 0) return IGX_OK;
    if (!x || !lower || !upper || !out) return fail(IGX_E_NULL, "%s: NULL argument", who);
    const int64_t total = n * d;
    hipLaunchKernelGGL((affine_kernel<KIND>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, lower, upper, out, total, (int)d);
    return launch_status(who);
}

}  // namespace igx

using namespace igx;

extern "C" int igx_quat_mul(const float *a, const float *b, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatMul>("igx_quat_mul", a, b, nullptr, out, n, s);
}
extern "C" int igx_quat_conjugate(const float *a, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatConjugate>("igx_quat_conjugate", a, nullptr, nullptr, out, n, s);
}
extern "C" int igx_quat_rotate(const float *q, const float *v, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatRotate<false>>("igx_quat_rotate", q, v, nullptr, out, n, s);
}
extern "C" int igx_quat_rotate_inverse(const float *q, const float *v, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatRotate<true>>("igx_quat_rotate_inverse", q, v, nullptr, out, n, s);
}
extern "C" int igx_slerp(const float *q0, const float *q1, const float *t, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpSlerp>("igx_slerp", q0, q1, t, out, n, s);
}
extern "C" int igx_scale(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d, igx_stream_t s)
{
    return run_affine<AFF_SCALE>("igx_scale", x, lo, hi, out, n, d, s);
}
extern "C" int igx_unscale(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d, igx_stream_t s)
{
    return run_affine<AFF_UNSCALE>("igx_unscale", x, lo, hi, out, n, d, s);
}
extern "C" int igx_tensor_clamp(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d,
                                igx_stream_t s)
{
    return run_affine<AFF_CLAMP>("igx_tensor_clamp", x, lo, hi, out, n, d, s);
}
extern "C" int igx_quat_from_angle_axis(const float *angle, const float *axis, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatFromAngleAxis>("igx_quat_from_angle_axis", angle, axis, nullptr, out, n, s);
}
extern "C" int igx_normalize_angle(const float *x, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpNormalizeAngle>("igx_normalize_angle", x, nullptr, nullptr, out, n, s);
}
extern "C" int igx_quat_to_exp_map(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatToExpMap>("igx_quat_to_exp_map", q, nullptr, nullptr, out, n, s);
}
extern "C" int igx_exp_map_to_quat(const float *e, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpExpMapToQuat>("igx_exp_map_to_quat", e, nullptr, nullptr, out, n, s);
}
extern "C" int igx_quat_to_tan_norm(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpQuatToTanNorm>("igx_quat_to_tan_norm", q, nullptr, nullptr, out, n, s);
}
extern "C" int igx_calc_heading_quat_inv(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_rowwise<OpCalcHeadingQuatInv>("igx_calc_heading_quat_inv", q, nullptr, nullptr, out, n, s);
}
