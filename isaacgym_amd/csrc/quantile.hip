// Exact quantile of n doubles on the device (torch.quantile, 'linear' interpolation):
// radix select of the two neighbouring order statistics, most significant digit
// first, 6 passes over the data (9 + 5 x 11 bits).  Per pass: every workgroup
// builds LDS histograms of the digit among the elements that match the prefix
// found so far, flushes the non-empty bins with atomics (integer adds: the result
// does not depend on their order), and a one-block kernel picks the bin that
// holds each rank.  For environment-sharded runs the caller all-reduces the
// histograms between the two kernels, which makes every rank pick the same bins.
//
// Reference: torch.quantile as called at IsaacGymEnvs/isaacgymenvs/tasks/ant_imitate.py:529
// (sort, ranks = q * (n - 1), lerp between floor and ceil ranks).
#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

namespace igx {

constexpr int QBINS = 2048;

struct QuantileState {  // 64 bytes, followed by hist[2][QBINS]
    unsigned long long prefix[2];
    unsigned long long rank[2];
    double weight;   // fractional part of q * (n - 1)
    double result;
    unsigned long long n_total;
    unsigned long long pad;
};
static_assert(sizeof(QuantileState) == 64, "state header");
static_assert(IGX_QUANTILE_WORKSPACE_BYTES == 64 + 2 * QBINS * 8, "workspace size is part of the ABI");

__host__ __device__ inline int digit_bits(int pass) { return pass == 0 ? 9 : 11; }
__host__ __device__ inline int digit_shift(int pass) { return pass == 0 ? 55 : 55 - 11 * pass; }

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

__global__ void quantile_begin_kernel(QuantileState *st, unsigned long long *hist, double q, unsigned long long n_total)
{
    for (int i = threadIdx.x; i < 2 * QBINS; i += blockDim.x) hist[i] = 0;
    if (threadIdx.x == 0) {
        const double ranks = q * (double)(n_total - 1);
        const long long below = (long long)ranks;  // toType(kLong): truncation
        const long long above = (long long)ceil(ranks);
        st->prefix[0] = st->prefix[1] = 0;
        st->rank[0] = (unsigned long long)below;
        st->rank[1] = (unsigned long long)above;
        st->weight = ranks - (double)below;
        st->result = 0.0;
        st->n_total = n_total;
    }
}

__global__ __launch_bounds__(256) void quantile_hist_kernel(const double *__restrict__ x, int64_t n, const QuantileState *st,
                                                            unsigned long long *hist, int pass)
{
    __shared__ unsigned s_hist[2][QBINS];
    for (int i = threadIdx.x; i < 2 * QBINS; i += 256) (&s_hist[0][0])[i] = 0;
    __syncthreads();
    const int shift = digit_shift(pass), bits = digit_bits(pass);
    const unsigned long long p0 = st->prefix[0], p1 = st->prefix[1];
    const bool same = p0 == p1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const unsigned long long k = key_of(ld_stream(x + i));
        const unsigned long long hi = pass == 0 ? 0ull : (k >> (shift + bits));
        const unsigned d = (unsigned)((k >> shift) & ((1u << bits) - 1u));
        if (hi == p0) atomicAdd(&s_hist[0][d], 1u);
        if (!same && hi == p1) atomicAdd(&s_hist[1][d], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * QBINS; i += 256) {
        const unsigned c = (&s_hist[0][0])[i];
        if (c) atomicAdd(&hist[i], (unsigned long long)c);
    }
}

// one block: the bin that holds each rank; histograms re-zeroed for the next pass
__global__ __launch_bounds__(256) void quantile_advance_kernel(QuantileState *st, unsigned long long *hist, int pass)
{
    __shared__ unsigned long long s_sum[256];
    const int bits = digit_bits(pass), nb = 1 << bits;
    const bool same = st->prefix[0] == st->prefix[1];
    for (int r = 0; r < 2; ++r) {
        const unsigned long long *h = hist + ((r == 1 && !same) ? QBINS : 0);
        // 256 threads x 8 consecutive bins
        unsigned long long loc[8], tot = 0;
        for (int i = 0; i < 8; ++i) {
            const int bin = threadIdx.x * 8 + i;
            loc[i] = bin < nb ? h[bin] : 0;
            tot += loc[i];
        }
        s_sum[threadIdx.x] = tot;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long run = 0;
            for (int t = 0; t < 256; ++t) {
                const unsigned long long v = s_sum[t];
                s_sum[t] = run;
                run += v;
            }
        }
        __syncthreads();
        const unsigned long long rank = st->rank[r];
        unsigned long long run = s_sum[threadIdx.x];
        __syncthreads();
        for (int i = 0; i < 8; ++i) {
            if (rank >= run && rank < run + loc[i]) {  // exactly one (thread, i) satisfies this
                st->prefix[r] = (st->prefix[r] << bits) | (unsigned long long)(threadIdx.x * 8 + i);
                st->rank[r] = rank - run;
            }
            run += loc[i];
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < 2 * QBINS; i += 256) hist[i] = 0;
    if (pass == IGX_QUANTILE_PASSES - 1 && threadIdx.x == 0) {
        const double a = value_of(st->prefix[0]), b = value_of(st->prefix[1]), w = st->weight;
        const double diff = b - a;  // at::native::lerp
        st->result = (fabs(w) < 0.5) ? a + w * diff : b - diff * (1.0 - w);
    }
}

__global__ void quantile_result_kernel(const QuantileState *st, double *out) { *out = st->result; }

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
    QuantileState *st = static_cast<QuantileState *>(workspace);
    hipLaunchKernelGGL(quantile_begin_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), st,
                       reinterpret_cast<unsigned long long *>(st + 1), q, (unsigned long long)n_total);
    return launch_status("igx_quantile_begin");
}

extern "C" int igx_quantile_histogram(const double *x, int64_t n, void *workspace, int32_t pass, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_histogram");
    if (rc != IGX_OK) return rc;
    if (pass < 0 || pass >= IGX_QUANTILE_PASSES) return fail(IGX_E_PARAM, "igx_quantile_histogram: pass %d", pass);
    if (n < 0) return fail(IGX_E_SIZE, "igx_quantile_histogram: negative n");
    if (n == 0) return IGX_OK;  // an empty shard contributes nothing
    if (!x) return fail(IGX_E_NULL, "igx_quantile_histogram: x is NULL");
    QuantileState *st = static_cast<QuantileState *>(workspace);
    int64_t blocks = (n + 256 * 16 - 1) / (256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(quantile_hist_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, n,
                       st, reinterpret_cast<unsigned long long *>(st + 1), (int)pass);
    return launch_status("igx_quantile_histogram");
}

extern "C" int igx_quantile_advance(void *workspace, int32_t pass, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_advance");
    if (rc != IGX_OK) return rc;
    if (pass < 0 || pass >= IGX_QUANTILE_PASSES) return fail(IGX_E_PARAM, "igx_quantile_advance: pass %d", pass);
    QuantileState *st = static_cast<QuantileState *>(workspace);
    hipLaunchKernelGGL(quantile_advance_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), st,
                       reinterpret_cast<unsigned long long *>(st + 1), (int)pass);
    return launch_status("igx_quantile_advance");
}

extern "C" int igx_quantile_f64(const double *x, int64_t n, double q, double *out, void *workspace, igx_stream_t stream)
{
    if (!out) return fail(IGX_E_NULL, "igx_quantile_f64: out is NULL");
    int rc = igx_quantile_begin(workspace, q, n, stream);
    for (int pass = 0; rc == IGX_OK && pass < IGX_QUANTILE_PASSES; ++pass) {
        rc = igx_quantile_histogram(x, n, workspace, pass, stream);
        if (rc == IGX_OK) rc = igx_quantile_advance(workspace, pass, stream);
    }
    if (rc != IGX_OK) return rc;
    hipLaunchKernelGGL(quantile_result_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream),
                       static_cast<const QuantileState *>(workspace), out);
    return launch_status("igx_quantile_f64");
}
