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
#include <initializer_list>

#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

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

// One slab of RT rows of W floats, global -> registers -> LDS in two steps, so that a block has the
// loads of all its input arrays in flight before it waits for the first.
template <int W, int RT>
struct Slab {
    static constexpr int N4 = RT * W / 4;
    static constexpr int IT = W ? (N4 + QTILE - 1) / QTILE : 1;
    using v4f = float __attribute__((ext_vector_type(4)));
    v4f r[IT];
    __device__ __forceinline__ void load(const float *__restrict__ src, int tid)
    {
        if (W == 0) return;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int k = tid + it * QTILE;
            if ((it + 1) * QTILE <= N4 || k < N4) r[it] = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + k);
        }
    }
    __device__ __forceinline__ void store(float *s, int tid) const
    {
        if (W == 0) return;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int k = tid + it * QTILE;
            if ((it + 1) * QTILE <= N4 || k < N4) *reinterpret_cast<v4f *>(s + 4 * k) = r[it];
        }
    }
};

__device__ __forceinline__ bool dev_aligned16_or_null(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// TPB tiles of QTILE rows per block: a block's lifetime is one memory round trip, so the bytes it
// moves per trip set the rate; more rows per block as long as the LDS still holds several blocks per CU.
template <class Op>
struct RowwiseShape {
    static constexpr int WSUM = (Op::A ? Op::A : 1) + (Op::B ? Op::B : 1) + (Op::C ? Op::C : 1) + Op::O;
    static constexpr int TPB = WSUM <= 8 ? 4 : 2;
    static constexpr int RT = QTILE * TPB;
};

template <class Op>
__global__ __launch_bounds__(QTILE) void rowwise_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                        const float *__restrict__ c, float *__restrict__ out, int64_t n)
{
    constexpr int RT = RowwiseShape<Op>::RT, TPB = RowwiseShape<Op>::TPB;
    __shared__ __align__(16) float sa[RT * (Op::A ? Op::A : 1)];
    __shared__ __align__(16) float sb[RT * (Op::B ? Op::B : 1)];
    __shared__ __align__(16) float sc[RT * (Op::C ? Op::C : 1)];
    __shared__ __align__(16) float so[RT * Op::O];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * RT;
    const int nvalid = (n - base >= RT) ? RT : (int)(n - base);
    const float *ga = a + base * Op::A, *gb = Op::B ? b + base * Op::B : nullptr, *gc = Op::C ? c + base * Op::C : nullptr;
    float *dst = out + base * Op::O;
    const bool fast = nvalid == RT && dev_aligned16_or_null(ga) && dev_aligned16_or_null(gb) && dev_aligned16_or_null(gc);
    if (fast) {
        Slab<Op::A, RT> ra;
        Slab<Op::B, RT> rb;
        Slab<Op::C, RT> rc;
        ra.load(ga, tid);
        rb.load(gb, tid);
        rc.load(gc, tid);
        ra.store(sa, tid);
        rb.store(sb, tid);
        rc.store(sc, tid);
    } else {
        for (int k = tid; k < nvalid * Op::A; k += QTILE) sa[k] = ga[k];
        for (int k = tid; k < nvalid * Op::B; k += QTILE) sb[k] = gb[k];
        for (int k = tid; k < nvalid * Op::C; k += QTILE) sc[k] = gc[k];
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPB; ++t) {
        const int row = tid + t * QTILE;
        if (row < nvalid) Op::compute(sa + row * Op::A, sb + row * Op::B, sc + row * Op::C, so + row * Op::O);
    }
    __syncthreads();
    const int count = nvalid * Op::O;
    if (nvalid == RT && dev_aligned16_or_null(dst)) {
        using v4f = float __attribute__((ext_vector_type(4)));
        constexpr int N4 = RT * Op::O / 4;
#pragma unroll
        for (int k0 = 0; k0 < N4; k0 += QTILE) {
            const int k = k0 + tid;
            if (k0 + QTILE <= N4 || k < N4) reinterpret_cast<v4f *>(dst)[k] = *reinterpret_cast<const v4f *>(so + 4 * k);
        }
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
    constexpr int RT = RowwiseShape<Op>::RT;
    const int64_t blocks = (n + RT - 1) / RT;
    hipLaunchKernelGGL((rowwise_kernel<Op>), dim3((unsigned)blocks), dim3(QTILE), 0, static_cast<hipStream_t>(stream), a,
                       b, c, out, n);
    return launch_status(who);
}

// ---- functions with more arrays: up to five per-item input rows, six output rows ----
constexpr int MAX_IN = 5, MAX_OUT = 6;
struct RowPtrs {
    const float *in[MAX_IN];
    float *out[MAX_OUT];
};
struct OpArg {  // the scalar arguments some functions take
    float f;
    int i;
};

template <int N>
constexpr int width_sum(const int (&w)[N])
{
    int t = 0;
    for (int k = 0; k < N; ++k) t += w[k];
    return t;
}

__device__ __forceinline__ void put3(float *o, f3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
__device__ __forceinline__ void put4(float *o, f4 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }
__device__ __forceinline__ f3 get3(const float *p) { return f3{p[0], p[1], p[2]}; }
__device__ __forceinline__ f4 get4(const float *p) { return f4{p[0], p[1], p[2], p[3]}; }

// Functor interface: IW / OW = widths of the input / output rows (0 = unused slot),
// compute(in, out, arg) on this item's LDS rows.
struct OpNormalize3 {  // :65-67
    static constexpr int IW[MAX_IN] = {3, 0, 0, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg) { put3(out[0], normalize3(get3(in[0]))); }
};
struct OpQuatUnit {  // :113-115
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {4, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg) { put4(out[0], quat_unit(get4(in[0]))); }
};
struct OpQuatApply {  // :70-77 (tf_vector :142-144 is the same function)
    static constexpr int IW[MAX_IN] = {4, 3, 0, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        put3(out[0], quat_apply(get4(in[0]), get3(in[1])));
    }
};
struct OpTfInverse {  // :131-134
    static constexpr int IW[MAX_IN] = {4, 3, 0, 0, 0}, OW[MAX_OUT] = {4, 3, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 qi = quat_conjugate(get4(in[0]));
        const f3 t = quat_apply(qi, get3(in[1]));
        put4(out[0], qi);
        put3(out[1], f3{-t.x, -t.y, -t.z});
    }
};
struct OpTfApply {  // :137-139
    static constexpr int IW[MAX_IN] = {4, 3, 3, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f3 r = quat_apply(get4(in[0]), get3(in[2]));
        const f3 t = get3(in[1]);
        put3(out[0], f3{r.x + t.x, r.y + t.y, r.z + t.z});
    }
};
struct OpTfCombine {  // :147-149
    static constexpr int IW[MAX_IN] = {4, 3, 4, 3, 0}, OW[MAX_OUT] = {4, 3, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 q1 = get4(in[0]);
        const f3 t1 = get3(in[1]);
        put4(out[0], quat_mul(q1, get4(in[2])));
        const f3 r = quat_apply(q1, get3(in[3]));
        put3(out[1], f3{r.x + t1.x, r.y + t1.y, r.z + t1.z});
    }
};
struct OpCopysign {  // :168-172
    static constexpr int IW[MAX_IN] = {1, 0, 0, 0, 0}, OW[MAX_OUT] = {1, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg a) { out[0][0] = copysign_sign(a.f, in[0][0]); }
};
struct OpGetEulerXyz {  // :175-195
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {1, 1, 1, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 q = get4(in[0]);
        float roll, yaw;
        euler_roll_yaw(q, roll, yaw);
        out[0][0] = roll;
        out[1][0] = euler_pitch(q);
        out[2][0] = yaw;
    }
};
struct OpQuatFromEulerXyz {  // :198-212
    static constexpr int IW[MAX_IN] = {1, 1, 1, 0, 0}, OW[MAX_OUT] = {4, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        put4(out[0], quat_from_euler_xyz(in[0][0], in[1][0], in[2][0]));
    }
};
struct OpEulerXyzToExpMap {  // :563-568
    static constexpr int IW[MAX_IN] = {1, 1, 1, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        put3(out[0], quat_to_exp_map(quat_from_euler_xyz(in[0][0], in[1][0], in[2][0])));
    }
};
struct OpHeadingAndUp {  // :248-263; arg.i = up_idx
    static constexpr int IW[MAX_IN] = {4, 4, 3, 3, 3}, OW[MAX_OUT] = {4, 1, 1, 3, 3, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg a)
    {
        const f3 target_dirs = normalize3(get3(in[2]));
        const f4 tq = quat_mul(get4(in[0]), get4(in[1]));
        const f3 up = quat_rotate<false>(tq, get3(in[4]));
        const f3 heading = quat_rotate<false>(tq, get3(in[3]));
        put4(out[0], tq);
        out[1][0] = a.i == 0 ? up.x : a.i == 1 ? up.y : up.z;
        out[2][0] = dot3(heading, target_dirs);
        put3(out[3], up);
        put3(out[4], heading);
    }
};
struct OpComputeRot {  // :266-277
    static constexpr int IW[MAX_IN] = {4, 3, 3, 3, 3}, OW[MAX_OUT] = {3, 3, 1, 1, 1, 1};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 tq = get4(in[0]);
        put3(out[0], quat_rotate<true>(tq, get3(in[1])));
        put3(out[1], quat_rotate<true>(tq, get3(in[2])));
        float roll, yaw;
        euler_roll_yaw(tq, roll, yaw);
        out[2][0] = roll;
        out[3][0] = euler_pitch(tq);
        out[4][0] = yaw;
        out[5][0] = atan2f(in[3][2] - in[4][2], in[3][0] - in[4][0]) - yaw;
    }
};
struct OpQuatAxis {  // :280-285; arg.i = axis
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg a)
    {
        const f3 basis{a.i == 0 ? 1.0f : 0.0f, a.i == 1 ? 1.0f : 0.0f, a.i == 2 ? 1.0f : 0.0f};
        put3(out[0], quat_rotate<false>(get4(in[0]), basis));
    }
};
struct OpQuatDiffRad {  // :354-373
    static constexpr int IW[MAX_IN] = {4, 4, 0, 0, 0}, OW[MAX_OUT] = {1, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 m = quat_mul(get4(in[0]), quat_conjugate(get4(in[1])));
        const float nrm = norm3(f3{m.x, m.y, m.z});
        out[0][0] = 2.0f * asinf(nrm > 1.0f ? 1.0f : nrm);  // torch.clamp(max=1): NaN stays NaN
    }
};
struct OpLocalToWorldSpace {  // :376-395; pose_global rows are (pos 3, quat 4)
    static constexpr int IW[MAX_IN] = {3, 7, 0, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const f4 local{in[0][0], in[0][1], in[0][2], 0.0f};
        const f4 qg{in[1][3], in[1][4], in[1][5], in[1][6]};
        const f4 r = quat_mul(qg, quat_mul(local, quat_conjugate(qg)));
        put3(out[0], f3{r.x + in[1][0], r.y + in[1][1], r.z + in[1][2]});
    }
};
struct OpQuatToAngleAxis {  // :423-444
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {1, 3, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        float angle;
        f3 axis;
        quat_to_angle_axis(get4(in[0]), angle, axis);
        out[0][0] = angle;
        put3(out[1], axis);
    }
};
struct OpAngleAxisToExpMap {  // :446-452
    static constexpr int IW[MAX_IN] = {1, 3, 0, 0, 0}, OW[MAX_OUT] = {3, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        const float a = in[0][0];
        put3(out[0], f3{a * in[1][0], a * in[1][1], a * in[1][2]});
    }
};
struct OpExpMapToAngleAxis {  // :570-587
    static constexpr int IW[MAX_IN] = {3, 0, 0, 0, 0}, OW[MAX_OUT] = {1, 3, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        float angle;
        f3 axis;
        exp_map_to_angle_axis(get3(in[0]), angle, axis);
        out[0][0] = angle;
        put3(out[1], axis);
    }
};
struct OpCalcHeading {  // :630-641
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {1, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg) { out[0][0] = calc_heading(get4(in[0])); }
};
struct OpCalcHeadingQuat {  // :643-654
    static constexpr int IW[MAX_IN] = {4, 0, 0, 0, 0}, OW[MAX_OUT] = {4, 0, 0, 0, 0, 0};
    __device__ static void compute(const float *const *in, float *const *out, OpArg)
    {
        put4(out[0], quat_from_angle_axis(calc_heading(get4(in[0])), f3{0.0f, 0.0f, 1.0f}));
    }
};

// one contiguous slab of `width`-float rows between global memory and LDS (the slab of a
// full tile is 256 * width floats: always whole 16-byte pieces)
__device__ __forceinline__ void slab_load(const float *__restrict__ g, float *s, int width, int64_t base, int nvalid, int tid)
{
    using v4f = float __attribute__((ext_vector_type(4)));
    const float *src = g + base * width;
    const int count = nvalid * width;
    if (nvalid == QTILE && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        for (int k = tid; k < count / 4; k += QTILE)
            *reinterpret_cast<v4f *>(s + 4 * k) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(src) + k);
    } else {
        for (int k = tid; k < count; k += QTILE) s[k] = src[k];
    }
}

__device__ __forceinline__ void slab_store(float *__restrict__ g, const float *s, int width, int64_t base, int nvalid, int tid)
{
    using v4f = float __attribute__((ext_vector_type(4)));
    float *dst = g + base * width;
    const int count = nvalid * width;
    if (nvalid == QTILE && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        for (int k = tid; k < count / 4; k += QTILE) reinterpret_cast<v4f *>(dst)[k] = *reinterpret_cast<const v4f *>(s + 4 * k);
    } else {
        for (int k = tid; k < count; k += QTILE) dst[k] = s[k];
    }
}

template <class Op>
__global__ __launch_bounds__(QTILE) void multi_kernel(const RowPtrs p, int64_t n, OpArg arg)
{
    constexpr int WI = width_sum(Op::IW), WO = width_sum(Op::OW);
    __shared__ __align__(16) float s_in[QTILE * WI];
    __shared__ __align__(16) float s_out[QTILE * WO];
    const int tid = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * QTILE;
    const int nvalid = (n - base >= QTILE) ? QTILE : (int)(n - base);
    const float *in[MAX_IN];
    float *out[MAX_OUT];
    static_assert(MAX_IN == 5, "the register staging below names five slabs");
    constexpr int O1 = Op::IW[0], O2 = O1 + Op::IW[1], O3 = O2 + Op::IW[2], O4 = O3 + Op::IW[3];
    bool fast = nvalid == QTILE && Op::IW[1] != 0;  // a single input array has nothing to overlap with
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k)
        if (Op::IW[k]) fast = fast && dev_aligned16_or_null(p.in[k] + base * Op::IW[k]);
    int off = 0;
    if (fast) {  // every array's loads in flight before the first wait
        Slab<Op::IW[0], QTILE> r0;
        Slab<Op::IW[1], QTILE> r1;
        Slab<Op::IW[2], QTILE> r2;
        Slab<Op::IW[3], QTILE> r3;
        Slab<Op::IW[4], QTILE> r4;
        r0.load(p.in[0] + base * Op::IW[0], tid);
        r1.load(p.in[1] + base * Op::IW[1], tid);
        r2.load(p.in[2] + base * Op::IW[2], tid);
        r3.load(p.in[3] + base * Op::IW[3], tid);
        r4.load(p.in[4] + base * Op::IW[4], tid);
        r0.store(s_in, tid);
        r1.store(s_in + QTILE * O1, tid);
        r2.store(s_in + QTILE * O2, tid);
        r3.store(s_in + QTILE * O3, tid);
        r4.store(s_in + QTILE * O4, tid);
    }
#pragma unroll
    for (int k = 0; k < MAX_IN; ++k) {
        in[k] = s_in + QTILE * off + tid * Op::IW[k];
        if (Op::IW[k] && !fast) slab_load(p.in[k], s_in + QTILE * off, Op::IW[k], base, nvalid, tid);
        off += Op::IW[k];
    }
    off = 0;
#pragma unroll
    for (int k = 0; k < MAX_OUT; ++k) {
        out[k] = s_out + QTILE * off + tid * Op::OW[k];
        off += Op::OW[k];
    }
    __syncthreads();
    if (tid < nvalid) Op::compute(in, out, arg);
    __syncthreads();
    off = 0;
#pragma unroll
    for (int k = 0; k < MAX_OUT; ++k) {
        if (Op::OW[k]) slab_store(p.out[k], s_out + QTILE * off, Op::OW[k], base, nvalid, tid);
        off += Op::OW[k];
    }
}

template <class Op>
static int run_multi(const char *who, const RowPtrs &p, int64_t n, OpArg arg, igx_stream_t stream)
{
    if (n < 0 || n > ((int64_t)1 << 38)) return fail(IGX_E_SIZE, "%s: n %lld out of range", who, (long long)n);
    if (n == 0) return IGX_OK;
    for (int k = 0; k < MAX_IN; ++k)
        if (Op::IW[k] && !p.in[k]) return fail(IGX_E_NULL, "%s: NULL input", who);
    for (int k = 0; k < MAX_OUT; ++k)
        if (Op::OW[k] && !p.out[k]) return fail(IGX_E_NULL, "%s: NULL output", who);
    const int64_t blocks = (n + QTILE - 1) / QTILE;
    hipLaunchKernelGGL((multi_kernel<Op>), dim3((unsigned)blocks), dim3(QTILE), 0, static_cast<hipStream_t>(stream), p, n,
                       arg);
    return launch_status(who);
}

// :215-218 torch_rand_float: span * u + lower over `count` elements; u from the caller or
// from the Philox stream (element i is lane i & 3 of counter block i >> 2)
__global__ __launch_bounds__(256) void rand_float_kernel(float span, float lower, const float *__restrict__ u, uint64_t seed,
                                                         uint64_t offset, float *__restrict__ out, int64_t count)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    float r;
    if (u) {
        r = u[i];
    } else {
        float q[4];
        philox_uniform4(seed, offset, i >> 2, 0, q);
        const int a = (int)(i & 3);
        r = a == 0 ? q[0] : a == 1 ? q[1] : a == 2 ? q[2] : q[3];
    }
    out[i] = span * r + lower;
}

// scale / unscale / tensor_clamp: x [n,d] against per-column lower / upper [d]
enum { AFF_SCALE = 0, AFF_UNSCALE = 1, AFF_CLAMP = 2, AFF_SCALE_TRANSFORM = 3, AFF_UNSCALE_TRANSFORM = 4 };

template <int KIND>
__device__ __forceinline__ float affine1(float v, float lo, float hi)
{
    if (KIND == AFF_SCALE) return scale(v, lo, hi);            // :234-236
    if (KIND == AFF_UNSCALE) return unscale(v, lo, hi);        // :239-241
    if (KIND == AFF_SCALE_TRANSFORM) return 2.0f * (v - (lo + hi) * 0.5f) / (hi - lo);    // :292-310
    if (KIND == AFF_UNSCALE_TRANSFORM) return v * (hi - lo) * 0.5f + (lo + hi) * 0.5f;    // :313-331
    return tensor_clamp(v, lo, hi);                            // :228-230, saturate :333-348
}

// VEC: AFF_U x four consecutive elements per lane through 16-byte loads and stores (total % 4 == 0,
// x and out 16-byte aligned), the per-column bounds staged in LDS when they fit; otherwise one
// element per lane.
constexpr int AFF_U = 4;
constexpr int AFF_LDS_COLS = 1024;

template <int KIND, bool VEC>
__global__ __launch_bounds__(256) void affine_kernel(const float *__restrict__ x, const float *__restrict__ lower,
                                                     const float *__restrict__ upper, float *__restrict__ out,
                                                     int64_t total, int d)
{
    if (VEC) {
        __shared__ float s_lo[AFF_LDS_COLS], s_hi[AFF_LDS_COLS];
        const int64_t t0 = (int64_t)blockIdx.x * (256 * AFF_U) + threadIdx.x;
        v4f v[AFF_U];
#pragma unroll
        for (int u = 0; u < AFF_U; ++u) {
            const int64_t t = t0 + u * 256;
            v[u] = ld_stream(reinterpret_cast<const v4f *>(x) + (4 * t < total ? t : 0));
        }
        const bool staged = d <= AFF_LDS_COLS;
        if (staged) {
            for (int c = threadIdx.x; c < d; c += 256) {
                s_lo[c] = lower[c];
                s_hi[c] = upper[c];
            }
            __syncthreads();
        }
        const float *lo = staged ? s_lo : lower, *hi = staged ? s_hi : upper;
#pragma unroll
        for (int u = 0; u < AFF_U; ++u) {
            const int64_t t = t0 + u * 256, i = 4 * t;
            if (i >= total) break;
            int col = total <= 0x7fffffffLL ? (int)((unsigned)i % (unsigned)d) : (int)(i % d);
            float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                e[k] = affine1<KIND>(e[k], lo[col], hi[col]);
                if (++col == d) col = 0;
            }
            st_stream(reinterpret_cast<v4f *>(out) + t, v4f{e[0], e[1], e[2], e[3]});
        }
    } else {
        const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
        if (t >= total) return;
        const int col = (int)(t % d);
        out[t] = affine1<KIND>(x[t], lower[col], upper[col]);
    }
}

template <int KIND>
static int run_affine(const char *who, const float *x, const float *lower, const float *upper, float *out, int64_t n,
                      int32_t d, igx_stream_t stream)
{
    if (n < 0 || d <= 0 || n > ((int64_t)1 << 38) / d) return fail(IGX_E_SIZE, "%s: bad shape [%lld,%d]", who, (long long)n, d);
    if (n == 0) return IGX_OK;
    if (!x || !lower || !upper || !out) return fail(IGX_E_NULL, "%s: NULL argument", who);
    const int64_t total = n * d;
    if (total % 4 == 0 && aligned16(x) && aligned16(out)) {
        hipLaunchKernelGGL((affine_kernel<KIND, true>), dim3((unsigned)((total / 4 + 256 * AFF_U - 1) / (256 * AFF_U))), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, lower, upper, out, total, (int)d);
    } else {
        hipLaunchKernelGGL((affine_kernel<KIND, false>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), x, lower, upper, out, total, (int)d);
    }
    return launch_status(who);
}

// :670-681 extract_positions_from_rigid_body_states: columns 13 i + {0, 1, 2}, i < bodies, of every row
// (the position of each body out of its 13-float state).  Every 64-byte line of a row holds part of a
// position, so the whole row is delivered either way: a block streams EXT_ROWS whole rows through LDS
// with 16-byte loads and writes the picked columns as one contiguous run.
constexpr int EXT_ROWS = 16;
constexpr int EXT_MAX_BODIES = 32;

__global__ __launch_bounds__(256) void extract_positions_tile_kernel(const float *__restrict__ x, float *__restrict__ out,
                                                                     int64_t n, int bodies)
{
    __shared__ __align__(16) float s_rows[EXT_ROWS * EXT_MAX_BODIES * 13];
    const int w = bodies * 13, ow = bodies * 3;
    const int64_t r0 = (int64_t)blockIdx.x * EXT_ROWS;
    const int rows = (n - r0 >= EXT_ROWS) ? EXT_ROWS : (int)(n - r0);
    const int chunks = rows * w / 4;  // rows * w is a multiple of 4 for a full tile; the last tile's tail is scalar
    const v4f *src = reinterpret_cast<const v4f *>(x + r0 * w);
    for (int k = threadIdx.x; k < chunks; k += 256) *reinterpret_cast<v4f *>(s_rows + 4 * k) = ld_stream(src + k);
    for (int k = 4 * chunks + threadIdx.x; k < rows * w; k += 256) s_rows[k] = x[r0 * w + k];
    __syncthreads();
    float *dst = out + r0 * ow;
    for (int k = threadIdx.x; k < rows * ow; k += 256) {
        const int r = k / ow, c = k - r * ow, body = c / 3;
        dst[k] = s_rows[r * w + body * 13 + (c - body * 3)];
    }
}

__global__ __launch_bounds__(256) void extract_positions_kernel(const float *__restrict__ x, int64_t row_stride,
                                                                float *__restrict__ out, int64_t total, int bodies)
{
    const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (k >= total) return;
    const int ow = bodies * 3;
    const int64_t r = k / ow;
    const int c = (int)(k - r * ow), body = c / 3;
    out[k] = x[r * row_stride + body * 13 + (c - body * 3)];
}

}  // namespace igx

using namespace igx;

extern "C" int igx_extract_positions_from_rigid_body_states(const float *states, int64_t row_stride, int32_t num_bodies,
                                                            float *out, int64_t n, igx_stream_t s)
{
    const char *who = "igx_extract_positions_from_rigid_body_states";
    if (num_bodies < 1 || num_bodies > EXT_MAX_BODIES) return fail(IGX_E_PARAM, "%s: num_bodies %d outside [1, %d]", who, num_bodies, EXT_MAX_BODIES);
    if (row_stride < (int64_t)num_bodies * 13) return fail(IGX_E_SIZE, "%s: rows of %lld floats do not hold %d bodies", who, (long long)row_stride, num_bodies);
    if (n < 0 || n > ((int64_t)1 << 36)) return fail(IGX_E_SIZE, "%s: n out of range", who);
    if (n == 0) return IGX_OK;
    if (!states || !out) return fail(IGX_E_NULL, "%s: NULL argument", who);
    hipStream_t st = static_cast<hipStream_t>(s);
    if (row_stride == (int64_t)num_bodies * 13 && aligned16(states) && (EXT_ROWS * num_bodies * 13) % 4 == 0) {
        hipLaunchKernelGGL(extract_positions_tile_kernel, dim3((unsigned)((n + EXT_ROWS - 1) / EXT_ROWS)), dim3(256), 0, st,
                           states, out, n, (int)num_bodies);
    } else {
        const int64_t total = n * num_bodies * 3;
        hipLaunchKernelGGL(extract_positions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, states,
                           row_stride, out, total, (int)num_bodies);
    }
    return launch_status(who);
}

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

// ---- the rest of utils/torch_jit_utils.py ----
static RowPtrs rows(std::initializer_list<const float *> in, std::initializer_list<float *> out)
{
    RowPtrs p{};
    int k = 0;
    for (const float *x : in) p.in[k++] = x;
    k = 0;
    for (float *x : out) p.out[k++] = x;
    return p;
}

extern "C" int igx_normalize(const float *x, float *out, int64_t n, int32_t d, igx_stream_t s)
{
    if (d == 3) return run_multi<OpNormalize3>("igx_normalize", rows({x}, {out}), n, OpArg{}, s);
    if (d == 4) return run_multi<OpQuatUnit>("igx_normalize", rows({x}, {out}), n, OpArg{}, s);
    return fail(IGX_E_PARAM, "igx_normalize: rows of 3 or 4 floats only (got %d)", d);
}
extern "C" int igx_quat_unit(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpQuatUnit>("igx_quat_unit", rows({q}, {out}), n, OpArg{}, s);
}
extern "C" int igx_quat_apply(const float *a, const float *b, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpQuatApply>("igx_quat_apply", rows({a, b}, {out}), n, OpArg{}, s);
}
extern "C" int igx_tf_inverse(const float *q, const float *t, float *q_out, float *t_out, int64_t n, igx_stream_t s)
{
    return run_multi<OpTfInverse>("igx_tf_inverse", rows({q, t}, {q_out, t_out}), n, OpArg{}, s);
}
extern "C" int igx_tf_apply(const float *q, const float *t, const float *v, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpTfApply>("igx_tf_apply", rows({q, t, v}, {out}), n, OpArg{}, s);
}
extern "C" int igx_tf_combine(const float *q1, const float *t1, const float *q2, const float *t2, float *q_out, float *t_out,
                              int64_t n, igx_stream_t s)
{
    return run_multi<OpTfCombine>("igx_tf_combine", rows({q1, t1, q2, t2}, {q_out, t_out}), n, OpArg{}, s);
}
extern "C" int igx_copysign(float a, const float *b, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpCopysign>("igx_copysign", rows({b}, {out}), n, OpArg{a, 0}, s);
}
extern "C" int igx_get_euler_xyz(const float *q, float *roll, float *pitch, float *yaw, int64_t n, igx_stream_t s)
{
    return run_multi<OpGetEulerXyz>("igx_get_euler_xyz", rows({q}, {roll, pitch, yaw}), n, OpArg{}, s);
}
extern "C" int igx_quat_from_euler_xyz(const float *roll, const float *pitch, const float *yaw, float *out, int64_t n,
                                       igx_stream_t s)
{
    return run_multi<OpQuatFromEulerXyz>("igx_quat_from_euler_xyz", rows({roll, pitch, yaw}, {out}), n, OpArg{}, s);
}
extern "C" int igx_euler_xyz_to_exp_map(const float *roll, const float *pitch, const float *yaw, float *out, int64_t n,
                                        igx_stream_t s)
{
    return run_multi<OpEulerXyzToExpMap>("igx_euler_xyz_to_exp_map", rows({roll, pitch, yaw}, {out}), n, OpArg{}, s);
}
extern "C" int igx_torch_rand_float(double lower, double upper, const float *uniforms, uint64_t seed, uint64_t rng_offset,
                                    float *out, int64_t count, igx_stream_t s)
{
    if (count < 0 || count > ((int64_t)1 << 40)) return fail(IGX_E_SIZE, "igx_torch_rand_float: count %lld out of range", (long long)count);
    if (count == 0) return IGX_OK;
    if (!out) return fail(IGX_E_NULL, "igx_torch_rand_float: NULL output");
    // (upper - lower) is a Python float subtraction; the product and the sum are fp32 tensor ops
    hipLaunchKernelGGL(rand_float_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(s),
                       (float)(upper - lower), (float)lower, uniforms, seed, rng_offset, out, count);
    return launch_status("igx_torch_rand_float");
}
extern "C" int igx_compute_heading_and_up(const float *torso_rotation, const float *inv_start_rot, const float *to_target,
                                          const float *vec0, const float *vec1, int32_t up_idx, float *torso_quat,
                                          float *up_proj, float *heading_proj, float *up_vec, float *heading_vec, int64_t n,
                                          igx_stream_t s)
{
    if (up_idx < 0 || up_idx > 2) return fail(IGX_E_PARAM, "igx_compute_heading_and_up: up_idx %d", up_idx);
    return run_multi<OpHeadingAndUp>("igx_compute_heading_and_up", rows({torso_rotation, inv_start_rot, to_target, vec0, vec1},
                                                                         {torso_quat, up_proj, heading_proj, up_vec, heading_vec}),
                                     n, OpArg{0.0f, up_idx}, s);
}
extern "C" int igx_compute_rot(const float *torso_quat, const float *velocity, const float *ang_velocity, const float *targets,
                               const float *torso_positions, float *vel_loc, float *angvel_loc, float *roll, float *pitch,
                               float *yaw, float *angle_to_target, int64_t n, igx_stream_t s)
{
    return run_multi<OpComputeRot>("igx_compute_rot", rows({torso_quat, velocity, ang_velocity, targets, torso_positions},
                                                           {vel_loc, angvel_loc, roll, pitch, yaw, angle_to_target}),
                                   n, OpArg{}, s);
}
extern "C" int igx_quat_axis(const float *q, int32_t axis, float *out, int64_t n, igx_stream_t s)
{
    if (axis < 0 || axis > 2) return fail(IGX_E_PARAM, "igx_quat_axis: axis %d", axis);
    return run_multi<OpQuatAxis>("igx_quat_axis", rows({q}, {out}), n, OpArg{0.0f, axis}, s);
}
extern "C" int igx_scale_transform(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d,
                                   igx_stream_t s)
{
    return run_affine<AFF_SCALE_TRANSFORM>("igx_scale_transform", x, lo, hi, out, n, d, s);
}
extern "C" int igx_unscale_transform(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d,
                                     igx_stream_t s)
{
    return run_affine<AFF_UNSCALE_TRANSFORM>("igx_unscale_transform", x, lo, hi, out, n, d, s);
}
extern "C" int igx_saturate(const float *x, const float *lo, const float *hi, float *out, int64_t n, int32_t d, igx_stream_t s)
{
    return run_affine<AFF_CLAMP>("igx_saturate", x, lo, hi, out, n, d, s);
}
extern "C" int igx_quat_diff_rad(const float *a, const float *b, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpQuatDiffRad>("igx_quat_diff_rad", rows({a, b}, {out}), n, OpArg{}, s);
}
extern "C" int igx_local_to_world_space(const float *pos_offset_local, const float *pose_global, float *out, int64_t n,
                                        igx_stream_t s)
{
    return run_multi<OpLocalToWorldSpace>("igx_local_to_world_space", rows({pos_offset_local, pose_global}, {out}), n, OpArg{}, s);
}
extern "C" int igx_quat_to_angle_axis(const float *q, float *angle, float *axis, int64_t n, igx_stream_t s)
{
    return run_multi<OpQuatToAngleAxis>("igx_quat_to_angle_axis", rows({q}, {angle, axis}), n, OpArg{}, s);
}
extern "C" int igx_angle_axis_to_exp_map(const float *angle, const float *axis, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpAngleAxisToExpMap>("igx_angle_axis_to_exp_map", rows({angle, axis}, {out}), n, OpArg{}, s);
}
extern "C" int igx_exp_map_to_angle_axis(const float *exp_map, float *angle, float *axis, int64_t n, igx_stream_t s)
{
    return run_multi<OpExpMapToAngleAxis>("igx_exp_map_to_angle_axis", rows({exp_map}, {angle, axis}), n, OpArg{}, s);
}
extern "C" int igx_calc_heading(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpCalcHeading>("igx_calc_heading", rows({q}, {out}), n, OpArg{}, s);
}
extern "C" int igx_calc_heading_quat(const float *q, float *out, int64_t n, igx_stream_t s)
{
    return run_multi<OpCalcHeadingQuat>("igx_calc_heading_quat", rows({q}, {out}), n, OpArg{}, s);
}
