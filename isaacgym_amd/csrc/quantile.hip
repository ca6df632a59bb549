// Exact quantile of n doubles on the device (torch.quantile, 'linear' interpolation):
// radix select of the two neighbouring order statistics, most significant digit
// first, 6 passes over the data (9 + 5 x 11 bits).  Per pass one kernel: every
// workgroup first derives the prefix found so far from the previous pass's complete
// histogram (all workgroups redundantly - 2048 bins, cheaper than a launch of its own),
// builds LDS histograms of this pass's digit among the elements that match the prefix,
// and flushes the non-empty bins with atomics (integer adds: the result does not depend
// on their order).  Each pass has its own histogram, zeroed at the start, so an
// environment-sharded run only has to all-reduce histogram `pass` between kernels
// `pass` and `pass + 1`, which makes every rank pick the same bins.
//
// Reference: torch.quantile as called at IsaacGymEnvs/isaacgymenvs/tasks/ant_imitate.py:529
// (sort, ranks = q * (n - 1), lerp between floor and ceil ranks).
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int QBINS = 2048;
constexpr int QPASSES = IGX_QUANTILE_PASSES;

struct QuantileState {  // 64 bytes; states[p] = after p digits; states[0].result also receives the final value
    unsigned long long prefix[2];
    unsigned long long rank[2];
    double weight;   // fractional part of q * (n - 1)
    double result;
    unsigned long long n_total;
    unsigned long long pad;
};
static_assert(sizeof(QuantileState) == 64, "state header");
static_assert(IGX_QUANTILE_HIST_OFFSET(0) == 8 * sizeof(QuantileState), "room for the states of all passes");
static_assert(IGX_QUANTILE_HIST_OFFSET(1) - IGX_QUANTILE_HIST_OFFSET(0) == 2 * QBINS * 8, "two histograms per pass");
static_assert(IGX_QUANTILE_WORKSPACE_BYTES == IGX_QUANTILE_HIST_OFFSET(QPASSES), "workspace size is part of the ABI");

__host__ __device__ inline int digit_bits(int pass) { return pass == 0 ? 9 : 11; }
__host__ __device__ inline int digit_shift(int pass) { return pass == 0 ? 55 : 55 - 11 * pass; }

__device__ __forceinline__ unsigned long long *hist_of(QuantileState *states, int pass)
{
    return reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(states) + IGX_QUANTILE_HIST_OFFSET(pass));
}

// order-preserving map of a double's bits to an unsigned integer
__device__ __forceinline__ unsigned long long key_of(double x)
{
    const unsigned long long u = (unsigned long long)__double_as_longlong(x);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double value_of(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__global__ __launch_bounds__(256) void quantile_begin_kernel(QuantileState *states, double q, unsigned long long n_total)
{
    unsigned long long *hist = hist_of(states, 0);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < QPASSES * 2 * QBINS; i += gridDim.x * 256) hist[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const double ranks = q * (double)(n_total - 1);
        const long long below = (long long)ranks;  // toType(kLong): truncation
        const long long above = (long long)ceil(ranks);
        QuantileState st;
        st.prefix[0] = st.prefix[1] = 0;
        st.rank[0] = (unsigned long long)below;
        st.rank[1] = (unsigned long long)above;
        st.weight = ranks - (double)below;
        st.result = 0.0;
        st.n_total = n_total;
        st.pad = 0;
        states[0] = st;
    }
}

// exclusive prefix sum over the 256 threads of the block
__device__ __forceinline__ unsigned long long block_exclusive_scan(unsigned long long v, unsigned long long *s_wave)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    unsigned long long inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long t = __shfl_up(inc, off);
        if (lane >= off) inc += t;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    unsigned long long base = 0;
    for (int i = 0; i < w; ++i) base += s_wave[i];
    __syncthreads();
    return base + inc - v;
}

// *out (LDS) = `in` advanced by digit `pass`, whose complete histogram is `hist`: the bin that
// holds each rank.  Called by all 256 threads of a block; ends with a barrier.
__device__ __forceinline__ void advance_state(const QuantileState *in, const unsigned long long *hist, int pass,
                                              QuantileState *out, unsigned long long *s_wave)
{
    if (threadIdx.x == 0) *out = *in;
    __syncthreads();
    const int bits = digit_bits(pass), nb = 1 << bits;
    const bool same = in->prefix[0] == in->prefix[1];
    for (int r = 0; r < 2; ++r) {
        const unsigned long long *h = hist + ((r == 1 && !same) ? QBINS : 0);
        // 256 threads x 8 consecutive bins
        unsigned long long loc[8], tot = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bin = threadIdx.x * 8 + i;
            loc[i] = bin < nb ? h[bin] : 0;
            tot += loc[i];
        }
        unsigned long long run = block_exclusive_scan(tot, s_wave);
        const unsigned long long rank = in->rank[r];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (rank >= run && rank < run + loc[i]) {  // exactly one (thread, i) satisfies this
                out->prefix[r] = (in->prefix[r] << bits) | (unsigned long long)(threadIdx.x * 8 + i);
                out->rank[r] = rank - run;
            }
            run += loc[i];
        }
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void quantile_hist_kernel(const double *__restrict__ x, int64_t n, QuantileState *states,
                                                            int pass)
{
    __shared__ unsigned s_hist[2][QBINS];
    __shared__ unsigned long long s_wave[4];
    __shared__ QuantileState s_st;
    for (int i = threadIdx.x; i < 2 * QBINS; i += 256) (&s_hist[0][0])[i] = 0;
    if (pass == 0) {
        if (threadIdx.x == 0) s_st = states[0];
        __syncthreads();
    } else {
        advance_state(&states[pass - 1], hist_of(states, pass - 1), pass - 1, &s_st, s_wave);
        if (blockIdx.x == 0 && threadIdx.x == 0) states[pass] = s_st;
    }
    unsigned long long *hist = hist_of(states, pass);
    const int shift = digit_shift(pass), bits = digit_bits(pass);
    const unsigned long long p0 = s_st.prefix[0], p1 = s_st.prefix[1];
    const bool same = p0 == p1;
    auto count = [&](double v) {
        // NaN: torch.quantile answers NaN as soon as one element is NaN.  Pass 0 uses 512 of its 2048 bins;
        // the last bin of its first histogram counts the NaNs (and is all-reduced with it in sharded runs).
        if (pass == 0 && v != v) atomicAdd(&s_hist[0][QBINS - 1], 1u);
        const unsigned long long k = key_of(v);
        const unsigned long long hi = pass == 0 ? 0ull : (k >> (shift + bits));
        const unsigned d = (unsigned)((k >> shift) & ((1u << bits) - 1u));
        if (hi == p0) atomicAdd(&s_hist[0][d], 1u);
        if (!same && hi == p1) atomicAdd(&s_hist[1][d], 1u);
    };
    // eight loads in flight per thread: one load per trip would pay the memory latency once per element
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * 256;
    int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld_narrow(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) count(v[u]);
    }
    for (; i < n; i += stride) count(ld_narrow(x + i));
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * QBINS; i += 256) {
        const unsigned c = (&s_hist[0][0])[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// one block: the last digit, then torch's lerp between the two order statistics
__global__ __launch_bounds__(256) void quantile_finish_kernel(QuantileState *states, double *out)
{
    __shared__ unsigned long long s_wave[4];
    __shared__ QuantileState s_st;
    advance_state(&states[QPASSES - 1], hist_of(states, QPASSES - 1), QPASSES - 1, &s_st, s_wave);
    if (threadIdx.x == 0) {
        const double a = value_of(s_st.prefix[0]), b = value_of(s_st.prefix[1]), w = s_st.weight;
        // at::native lerp as torch's CPU kernel evaluates it: one fused multiply-add onto the nearer end
        const double diff = b - a;
        double r = (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
        if (hist_of(states, 0)[QBINS - 1] != 0) r = __longlong_as_double(0x7ff8000000000000ll);  // some element was NaN
        s_st.result = r;
        states[QPASSES] = s_st;
        states[0].result = r;
        if (out) *out = r;
    }
}

static int check_ws(void *ws, const char *who)
{
    if (!ws) return fail(IGX_E_NULL, "%s: workspace is NULL", who);
    if (!aligned16(ws)) return fail(IGX_E_ALIGN, "%s: workspace must be 16-byte aligned", who);
    return IGX_OK;
}

}  // namespace igx

using namespace igx;

extern "C" int igx_quantile_begin(void *workspace, double q, int64_t n_total, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_begin");
    if (rc != IGX_OK) return rc;
    if (!(q >= 0.0 && q <= 1.0)) return fail(IGX_E_PARAM, "igx_quantile_begin: q must lie in [0, 1]");
    if (n_total <= 0) return fail(IGX_E_SIZE, "igx_quantile_begin: n_total must be positive");
    hipLaunchKernelGGL(quantile_begin_kernel, dim3(8), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<QuantileState *>(workspace), q, (unsigned long long)n_total);
    return launch_status("igx_quantile_begin");
}

extern "C" int igx_quantile_histogram(const double *x, int64_t n, void *workspace, int32_t pass, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_histogram");
    if (rc != IGX_OK) return rc;
    if (pass < 0 || pass >= IGX_QUANTILE_PASSES) return fail(IGX_E_PARAM, "igx_quantile_histogram: pass %d", pass);
    if (n < 0) return fail(IGX_E_SIZE, "igx_quantile_histogram: negative n");
    if (n > 0 && !x) return fail(IGX_E_NULL, "igx_quantile_histogram: x is NULL");
    // an empty shard contributes nothing but still advances its copy of the state
    static const int per_thread = tuning_int("IGX_QUANTILE_PER_THREAD", 16);
    int64_t blocks = (n + 256 * (int64_t)per_thread - 1) / (256 * (int64_t)per_thread);
    blocks = blocks < 1 ? 1 : blocks > 2048 ? 2048 : blocks;
    hipLaunchKernelGGL(quantile_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n,
                       static_cast<QuantileState *>(workspace), (int)pass);
    return launch_status("igx_quantile_histogram");
}

extern "C" int igx_quantile_finish(void *workspace, double *out, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_finish");
    if (rc != IGX_OK) return rc;
    hipLaunchKernelGGL(quantile_finish_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<QuantileState *>(workspace), out);
    return launch_status("igx_quantile_finish");
}

extern "C" int igx_quantile_f64(const double *x, int64_t n, double q, double *out, void *workspace, igx_stream_t stream)
{
    if (!out) return fail(IGX_E_NULL, "igx_quantile_f64: out is NULL");
    int rc = igx_quantile_begin(workspace, q, n, stream);
    for (int pass = 0; rc == IGX_OK && pass < IGX_QUANTILE_PASSES; ++pass)
        rc = igx_quantile_histogram(x, n, workspace, pass, stream);
    if (rc != IGX_OK) return rc;
    return igx_quantile_finish(workspace, out, stream);
}
