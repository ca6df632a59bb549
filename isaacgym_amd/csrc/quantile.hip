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

// ascending bitonic sort of P (a power of two, <= 4096) keys in LDS by one block of 1024 threads;
// the keys must be in place (barrier before the call); ends with a barrier
__device__ __forceinline__ void bitonic_sort_1024(unsigned long long *s_keys, unsigned P)
{
    const unsigned t = threadIdx.x;
    for (unsigned size = 2; size <= P; size <<= 1)
        for (unsigned str = size >> 1; str > 0; str >>= 1) {
            for (unsigned i = t; i < P / 2; i += 1024) {
                const unsigned a = 2 * i - (i & (str - 1)), b = a + str;  // the pair this compare-exchange owns
                const bool up = (a & size) == 0;
                const unsigned long long ka = s_keys[a], kb = s_keys[b];
                if ((ka > kb) == up) {
                    s_keys[a] = kb;
                    s_keys[b] = ka;
                }
            }
            __syncthreads();
        }
}

// The whole select by ONE workgroup of 1024 threads: eight 8-bit digits with LDS histograms, the same two order
// statistics as the multi-pass kernels.  REG: n <= 4096 and every thread keeps its four keys in registers;
// otherwise the values are streamed from memory once per digit.  Leaves the two keys in s_prefix and the NaN
// count in *s_nan (all in LDS, valid after the call's final barrier).
constexpr int QSMALL = 4096;
struct BlockSelect {
    unsigned hist[2][256];
    unsigned long long prefix[2];
    unsigned rank[2];
    unsigned nan;
};

template <bool REG>
__device__ __forceinline__ void block_select(const double *__restrict__ x, int64_t n, long long below, long long above,
                                             BlockSelect &s)
{
    const unsigned t = threadIdx.x, lane = t & 63, w = t >> 6;
    unsigned long long k[4] = {0, 0, 0, 0};
    bool valid[4] = {false, false, false, false};
    unsigned nans = 0;
    if (REG) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned i = t + c * 1024;
            valid[c] = (int64_t)i < n;
            const double v = valid[c] ? ld_narrow(x + i) : 0.0;
            nans += (valid[c] && v != v) ? 1u : 0u;
            k[c] = key_of(v);
        }
    }
    if (t == 0) {
        s.prefix[0] = s.prefix[1] = 0;
        s.rank[0] = (unsigned)below;
        s.rank[1] = (unsigned)above;
        s.nan = 0;
    }
    if (t < 512) (&s.hist[0][0])[t] = 0;
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        const unsigned long long p0 = s.prefix[0], p1 = s.prefix[1];
        const bool same = p0 == p1;
        auto count = [&](unsigned long long key) {
            const unsigned long long hi = pass == 0 ? 0ull : (key >> (shift + 8));
            const unsigned d = (unsigned)(key >> shift) & 255u;
            if (hi == p0) atomicAdd(&s.hist[0][d], 1u);
            if (!same && hi == p1) atomicAdd(&s.hist[1][d], 1u);
        };
        if (REG) {
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (valid[c]) count(k[c]);
        } else {
            constexpr int U = 8;
            int64_t i = t;
            for (; i + (U - 1) * 1024 < n; i += U * 1024) {
                double v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld_narrow(x + i + u * 1024);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (pass == 0) nans += v[u] != v[u] ? 1u : 0u;
                    count(key_of(v[u]));
                }
            }
            for (; i < n; i += 1024) {
                const double v = ld_narrow(x + i);
                if (pass == 0) nans += v != v ? 1u : 0u;
                count(key_of(v));
            }
        }
        if (pass == 0 && nans) atomicAdd(&s.nan, nans);
        __syncthreads();
        if (w < 2) {  // wave r finds the bin of rank r: four bins per lane, a scan over the wave
            const unsigned *h = s.hist[(w == 1 && !same) ? 1 : 0];
            unsigned c4[4], tot = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                c4[i] = h[lane * 4 + i];
                tot += c4[i];
            }
            unsigned inc = tot;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(inc, off);
                if (lane >= off) inc += v;
            }
            unsigned run = inc - tot;
            const unsigned rank = s.rank[w];
            const unsigned long long pw = w == 0 ? p0 : p1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (rank >= run && rank < run + c4[i]) {  // exactly one (lane, i)
                    s.prefix[w] = (pw << 8) | (unsigned long long)(lane * 4 + i);
                    s.rank[w] = rank - run;
                }
                run += c4[i];
            }
        }
        __syncthreads();
        if (t < 512) (&s.hist[0][0])[t] = 0;
        __syncthreads();
    }
}

// torch's lerp between the two order statistics (at::native lerp: one fused multiply-add onto the nearer end)
__device__ __forceinline__ double quantile_lerp(unsigned long long k0, unsigned long long k1, double w, bool any_nan)
{
    const double a = value_of(k0), b = value_of(k1), diff = b - a;
    const double r = (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
    return any_nan ? __longlong_as_double(0x7ff8000000000000ll) : r;
}

// n <= QSMALL values: one launch instead of eight (the select of a task at the reference's default 4 096
// environments).
__global__ __launch_bounds__(1024) void quantile_small_kernel(const double *__restrict__ x, int n, double q,
                                                              QuantileState *states, double *out)
{
    __shared__ BlockSelect s;
    const double ranks = q * (double)(n - 1);
    const long long below = (long long)ranks;  // toType(kLong): truncation
    const long long above = (long long)ceil(ranks);
    block_select<true>(x, n, below, above, s);
    if (threadIdx.x == 0) {
        const double wgt = ranks - (double)below;
        const double r = quantile_lerp(s.prefix[0], s.prefix[1], wgt, s.nan != 0);
        QuantileState st;
        st.prefix[0] = s.prefix[0];
        st.prefix[1] = s.prefix[1];
        st.rank[0] = (unsigned long long)below;
        st.rank[1] = (unsigned long long)above;
        st.weight = wgt;
        st.result = r;
        st.n_total = (unsigned long long)n;
        st.pad = 0;
        states[0] = st;
        if (out) *out = r;
    }
}

// ---- the same select in two collectives: histogram over a shared window of the key space ----------------
//
// Bin 0 holds the keys below the window, bins 1 .. WBINS - 2 are 2^shift keys wide starting at `lo`, bin
// WBINS - 1 holds everything above.  The located bin is the one that contains rank[0] of the all-reduced
// histogram; when it holds at most `capacity` keys the ranks exchange those keys and sort them, otherwise
// the window shrinks to that bin (or moves to the part below / above it) and the shards are counted again.
constexpr int WBINS = IGX_QUANTILE_WINDOW_BINS;
constexpr unsigned WMAGIC = 0x51574e44u;
constexpr unsigned WSHIFT_MAX = 50;  // (WBINS - 2) << 50 still fits 64 bits
constexpr unsigned WNONE = 0xffffffffu;
constexpr unsigned IGX_QW_BAD_GATHER = 5;
constexpr unsigned IGX_QW_FELL_BACK = 6;  // host-free sequence: the value came from the one-workgroup select

struct WindowState {  // the first 128 bytes of the workspace
    unsigned long long lo;
    unsigned long long rank[2];
    double weight;
    unsigned long long n_total;
    unsigned long long below;  // keys in the bins before the located one
    double result;
    unsigned shift, magic;
    unsigned bin, count;       // the located bin and what it holds (all ranks together)
    unsigned rounds, status;
    unsigned nan, inside;      // some element was NaN; rank[1] lies in the located bin too
    unsigned capacity, pad[9];
};
static_assert(sizeof(WindowState) == IGX_QUANTILE_WINDOW_HIST_OFFSET, "state header");
static_assert(IGX_QUANTILE_WINDOW_MAX_CANDIDATES == 4096, "the finishing sort holds the candidates in LDS");

__device__ __forceinline__ unsigned *window_hist(void *ws)
{
    return reinterpret_cast<unsigned *>(static_cast<char *>(ws) + IGX_QUANTILE_WINDOW_HIST_OFFSET);
}

__device__ __forceinline__ unsigned window_bin(unsigned long long k, unsigned long long lo, unsigned shift)
{
    if (k < lo) return 0u;
    const unsigned long long d = (k - lo) >> shift;
    return d >= (unsigned long long)(WBINS - 2) ? (unsigned)(WBINS - 1) : (unsigned)d + 1u;
}

// smallest shift (at most WSHIFT_MAX) whose window spans `span` keys
__device__ __forceinline__ unsigned shift_for_span(unsigned long long span)
{
    unsigned s = 0;
    while (s < WSHIFT_MAX && ((unsigned long long)(WBINS - 2) << s) < span) ++s;
    return s;
}

__global__ __launch_bounds__(256) void window_begin_kernel(WindowState *st, double q, unsigned long long n_total, int reset,
                                                           unsigned long long *slots)
{
    unsigned *hist = window_hist(st);
    if (slots && blockIdx.x == 0 && threadIdx.x < 2) slots[threadIdx.x] = threadIdx.x == 0 ? 0ull : ~0ull;  // count, minimum
    for (int i = blockIdx.x * 256 + threadIdx.x; i < IGX_QUANTILE_WINDOW_HIST_WORDS; i += gridDim.x * 256) hist[i] = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        WindowState s = *st;
        if (reset || s.magic != WMAGIC || s.shift > WSHIFT_MAX) {
            s.lo = 0;
            s.shift = WSHIFT_MAX;
        }
        const double ranks = q * (double)(n_total - 1);
        const long long below = (long long)ranks;
        const long long above = (long long)ceil(ranks);
        s.rank[0] = (unsigned long long)below;
        s.rank[1] = (unsigned long long)above;
        s.weight = ranks - (double)below;
        s.n_total = n_total;
        s.below = 0;
        s.result = 0.0;
        s.magic = WMAGIC;
        s.bin = WNONE;
        s.count = 0;
        s.rounds = 0;
        s.status = IGX_QW_REFINED;
        s.nan = 0;
        s.inside = 0;
        s.capacity = 0;
        for (int i = 0; i < 9; ++i) s.pad[i] = 0;
        *st = s;
    }
}

__global__ __launch_bounds__(1024) void window_hist_kernel(const double *__restrict__ x, int64_t n, WindowState *st)
{
    __shared__ unsigned s_hist[WBINS];  // 64 KB
    if (st->status != IGX_QW_REFINED) return;  // a round of the host-free sequence that the round before made unnecessary
    for (int i = threadIdx.x; i < WBINS; i += 1024) s_hist[i] = 0;
    const unsigned long long lo = st->lo;
    const unsigned shift = st->shift;
    __syncthreads();
    unsigned nans = 0;
    auto count = [&](double v) {
        nans += v != v ? 1u : 0u;
        atomicAdd(&s_hist[window_bin(key_of(v), lo, shift)], 1u);
    };
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld_narrow(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) count(v[u]);
    }
    for (; i < n; i += stride) count(ld_narrow(x + i));
    __syncthreads();
    unsigned *hist = window_hist(st);
    for (int b = threadIdx.x; b < WBINS; b += 1024) {
        const unsigned c = s_hist[b];
        if (c) atomicAdd(&hist[b], c);
    }
    if (nans) atomicAdd(&hist[WBINS], nans);
}

// one block of 1024 threads x 16 bins: which bin holds rank[0], and what to do about it
__global__ __launch_bounds__(1024) void window_locate_kernel(WindowState *st, unsigned capacity, long long *decision)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned long long s_below;
    __shared__ unsigned s_bin, s_count, s_status;
    unsigned *hist = window_hist(st);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (st->status != IGX_QW_REFINED) return;  // already located (block-uniform: one block, read before any write)
    if (t == 0) {
        s_bin = WNONE;
        s_count = 0;
        s_below = 0;
    }
    unsigned loc[16];
    unsigned long long tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        loc[i] = hist[t * 16 + i];
        tot += loc[i];
    }
    unsigned long long inc = tot;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned long long v = __shfl_up(inc, off);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[w] = inc;
    __syncthreads();
    unsigned long long run = inc - tot;
    for (int i = 0; i < w; ++i) run += s_wave[i];
    const unsigned long long r0 = st->rank[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (r0 >= run && r0 < run + loc[i]) {  // at most one (thread, i)
            s_bin = (unsigned)(t * 16 + i);
            s_count = loc[i];
            s_below = run;
        }
        run += loc[i];
    }
    __syncthreads();
    if (t == 0) {
        WindowState s = *st;
        const unsigned b = s_bin, c = s_count;
        s.rounds += 1;
        s.bin = b;
        s.count = c;
        s.below = s_below;
        s.inside = (b != WNONE && s.rank[1] - s_below < (unsigned long long)c) ? 1u : 0u;
        s.nan |= hist[WBINS] != 0 ? 1u : 0u;
        s.capacity = capacity;
        unsigned status;
        const bool interior = b >= 1u && b <= (unsigned)(WBINS - 2);
        if (b == WNONE) {
            status = IGX_QW_GIVE_UP;  // the histogram does not hold n_total elements
        } else if (c <= capacity) {
            status = IGX_QW_COLLECT;
        } else if (interior && s.shift == 0) {
            status = IGX_QW_VALUE;    // more than `capacity` copies of one key
        } else if (s.rounds >= IGX_QUANTILE_WINDOW_MAX_ROUNDS) {
            status = IGX_QW_GIVE_UP;
        } else {
            status = IGX_QW_REFINED;
            if (interior) {
                s.lo = s.lo + ((unsigned long long)(b - 1u) << s.shift);
                s.shift = s.shift > 13u ? s.shift - 13u : 0u;  // (WBINS - 2) << (shift - 13) >= 1 << shift
            } else if (b == 0u) {
                s.shift = shift_for_span(s.lo);
                s.lo = 0;
            } else {
                s.lo = s.lo + ((unsigned long long)(WBINS - 2) << s.shift);
                s.shift = shift_for_span(0ull - s.lo);
            }
        }
        if (status == IGX_QW_GIVE_UP) s.magic = 0;  // the next call starts from the whole key space
        s.status = status;
        *st = s;
        s_status = status;
        if (decision) {
            decision[0] = (long long)status;
            decision[1] = status == IGX_QW_COLLECT ? (long long)c : 0ll;
            decision[2] = (long long)b;
            decision[3] = (long long)s.rounds;
        }
    }
    __syncthreads();
    if (s_status == IGX_QW_REFINED) {
#pragma unroll
        for (int i = 0; i < 16; ++i) hist[t * 16 + i] = 0;
        if (t < IGX_QUANTILE_WINDOW_HIST_WORDS - WBINS) hist[WBINS + t] = 0;
    }
}

__global__ void window_slots_init_kernel(unsigned long long *slots)
{
    if (threadIdx.x == 0) slots[0] = 0ull;
    if (threadIdx.x == 1) slots[1] = ~0ull;
}

// slots = {number of this shard's keys in the located bin, smallest key of this shard above that bin, the keys}.
// Same-address atomics are what this kernel would otherwise spend its time on (one per candidate and one
// minimum per wave, a few nanoseconds each, serialised): a workgroup gathers its candidates and its minimum in
// LDS and goes to memory with one add and one min.
__global__ __launch_bounds__(1024) void window_collect_kernel(const double *__restrict__ x, int64_t n, const WindowState *st,
                                                              unsigned long long *slots, long long num_slots)
{
    __shared__ unsigned long long s_keys[IGX_QUANTILE_WINDOW_MAX_CANDIDATES];
    __shared__ unsigned long long s_min, s_base;
    __shared__ unsigned s_cnt;
    const unsigned long long lo = st->lo;
    const unsigned shift = st->shift, b0 = st->bin;
    const bool emit = st->status == IGX_QW_COLLECT;
    if (!emit && st->status != IGX_QW_VALUE) return;  // host-free sequence: nothing was located, the fallback selects
    if (threadIdx.x == 0) {
        s_min = ~0ull;
        s_cnt = 0;
    }
    __syncthreads();
    unsigned long long mn = ~0ull;
    auto take = [&](double v) {
        const unsigned long long k = key_of(v);
        const unsigned b = window_bin(k, lo, shift);
        if (b == b0) {
            if (emit) {  // at most `capacity` keys of all shards together lie in the bin, so the LDS list cannot overflow
                const unsigned pos = atomicAdd(&s_cnt, 1u);
                if (pos < IGX_QUANTILE_WINDOW_MAX_CANDIDATES) s_keys[pos] = k;
            }
        } else if (b > b0) {
            mn = k < mn ? k : mn;
        }
    };
    // eight loads in flight per thread, as in the histogram kernels
    constexpr int U = 8;
    const int64_t stride = (int64_t)gridDim.x * 1024;
    int64_t i = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    for (; i + (U - 1) * stride < n; i += U * stride) {
        double v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ld_narrow(x + i + u * stride);
#pragma unroll
        for (int u = 0; u < U; ++u) take(v[u]);
    }
    for (; i < n; i += stride) take(ld_narrow(x + i));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = __shfl_xor(mn, off);
        mn = o < mn ? o : mn;
    }
    if ((threadIdx.x & 63) == 0 && mn != ~0ull) atomicMin(&s_min, mn);
    __syncthreads();
    const unsigned cnt = s_cnt < IGX_QUANTILE_WINDOW_MAX_CANDIDATES ? s_cnt : IGX_QUANTILE_WINDOW_MAX_CANDIDATES;
    if (threadIdx.x == 0) {
        if (s_min != ~0ull) atomicMin(&slots[1], s_min);
        s_base = s_cnt ? atomicAdd(&slots[0], (unsigned long long)s_cnt) : 0ull;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    for (unsigned j = threadIdx.x; j < cnt; j += 1024)
        if ((long long)(base + j) + 2 < num_slots) slots[2 + base + j] = s_keys[j];
}

// one block: sort the gathered keys of the located bin, pick the two order statistics, interpolate like
// quantile_finish_kernel, centre the window on the value for the next call
__global__ __launch_bounds__(1024) void window_finish_kernel(WindowState *st, const unsigned long long *__restrict__ g,
                                                             int num_ranks, long long num_slots, double *out,
                                                             const double *__restrict__ x, int64_t n)
{
    __shared__ unsigned long long s_keys[IGX_QUANTILE_WINDOW_MAX_CANDIDATES];
    __shared__ unsigned s_off[IGX_QUANTILE_WINDOW_MAX_RANKS + 1];
    __shared__ unsigned long long s_min;
    __shared__ unsigned s_ok;
    const int t = threadIdx.x;
    const WindowState s = *st;
    // host-free sequence (x given): the rounds that were enqueued did not resolve the value (a cold or outrun window
    // on data that needs more than two rounds, or more than 4 096 copies of the value) - this workgroup runs the whole
    // select itself: slow (it streams the values eight times through one CU) but exact, and rare: the window the
    // rounds left behind is kept, so the next call continues from it.
    if (x && s.status != IGX_QW_COLLECT && s.status != IGX_QW_VALUE) {
        __shared__ BlockSelect sel;
        block_select<false>(x, n, (long long)s.rank[0], (long long)s.rank[1], sel);
        if (t == 0) {
            const double r = quantile_lerp(sel.prefix[0], sel.prefix[1], s.weight, sel.nan != 0);
            st->result = r;
            st->rounds = 0;
            st->status = IGX_QW_FELL_BACK;
            if (out) *out = r;
        }
        return;
    }
    if (t == 0) {
        unsigned long long total = 0, mn = ~0ull;
        bool ok = s.status == IGX_QW_COLLECT || s.status == IGX_QW_VALUE;
        for (int r = 0; r < num_ranks; ++r) {
            const unsigned long long c = g[(long long)r * num_slots];
            const unsigned long long m = g[(long long)r * num_slots + 1];
            s_off[r] = (unsigned)(total < IGX_QUANTILE_WINDOW_MAX_CANDIDATES ? total : IGX_QUANTILE_WINDOW_MAX_CANDIDATES);
            if ((long long)c + 2 > num_slots) ok = false;
            total += c;
            mn = m < mn ? m : mn;
        }
        const unsigned long long want = s.status == IGX_QW_COLLECT ? s.count : 0ull;
        if (total != want || total > IGX_QUANTILE_WINDOW_MAX_CANDIDATES) ok = false;
        s_off[num_ranks] = ok ? (unsigned)total : 0u;
        s_min = mn;
        s_ok = ok ? 1u : 0u;
    }
    __syncthreads();
    const bool ok = s_ok != 0;
    const unsigned total = s_off[num_ranks];
    unsigned P = 2;
    while (P < total) P <<= 1;
    for (unsigned i = t; i < P; i += 1024) s_keys[i] = ~0ull;
    __syncthreads();
    if (ok)
        for (int r = 0; r < num_ranks; ++r) {
            const unsigned o = s_off[r], c = s_off[r + 1] - o;
            for (unsigned j = t; j < c; j += 1024) s_keys[o + j] = g[(long long)r * num_slots + 2 + j];
        }
    __syncthreads();
    bitonic_sort_1024(s_keys, P);
    if (t == 0) {
        WindowState n = s;
        unsigned long long k0, k1;
        if (s.status == IGX_QW_VALUE) {
            k0 = s.lo + (unsigned long long)(s.bin - 1u);
            k1 = (s.rank[1] == s.rank[0] || s.inside) ? k0 : s_min;
        } else {
            const unsigned long long i0 = s.rank[0] - s.below;
            k0 = i0 < total ? s_keys[i0] : 0ull;
            k1 = s.rank[1] == s.rank[0] ? k0 : (s.inside ? s_keys[i0 + 1 < total ? i0 + 1 : i0] : s_min);
        }
        const double a = value_of(k0), b = value_of(k1), w = s.weight;
        const double diff = b - a;
        double r = (fabs(w) < 0.5) ? fma(w, diff, a) : fma(w - 1.0, diff, b);
        if (s.nan || !ok) r = __longlong_as_double(0x7ff8000000000000ll);
        n.result = r;
        n.rounds = 0;
        if (!ok) {
            n.status = IGX_QW_BAD_GATHER;
            n.magic = 0;  // the next call starts from the whole key space
        } else {
            if (s.status == IGX_QW_COLLECT) {  // keep the located bin between capacity / 64 and capacity / 2 keys
                if (s.count > s.capacity / 2u && n.shift > 0u) n.shift -= 1u;
                else if (s.count < s.capacity / 64u && n.shift < WSHIFT_MAX) n.shift += 1u;
            }
            const unsigned long long half = (unsigned long long)(WBINS / 2 - 1) << n.shift;
            n.lo = k0 > half ? k0 - half : 0ull;
        }
        *st = n;
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
    static const int small_max = tuning_int("IGX_QUANTILE_SMALL_MAX", QSMALL);
    if (n > 0 && n <= small_max && n <= QSMALL) {
        int rc0 = check_ws(workspace, "igx_quantile_f64");
        if (rc0 != IGX_OK) return rc0;
        if (!x) return fail(IGX_E_NULL, "igx_quantile_f64: x is NULL");
        if (!(q >= 0.0 && q <= 1.0)) return fail(IGX_E_PARAM, "igx_quantile_f64: q must lie in [0, 1]");
        hipLaunchKernelGGL(quantile_small_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), x, (int)n, q,
                           static_cast<QuantileState *>(workspace), out);
        return launch_status("igx_quantile_f64");
    }
    int rc = igx_quantile_begin(workspace, q, n, stream);
    for (int pass = 0; rc == IGX_OK && pass < IGX_QUANTILE_PASSES; ++pass)
        rc = igx_quantile_histogram(x, n, workspace, pass, stream);
    if (rc != IGX_OK) return rc;
    return igx_quantile_finish(workspace, out, stream);
}

extern "C" int igx_quantile_window_begin(void *workspace, double q, int64_t n_total, int32_t reset_window, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_window_begin");
    if (rc != IGX_OK) return rc;
    if (!(q >= 0.0 && q <= 1.0)) return fail(IGX_E_PARAM, "igx_quantile_window_begin: q must lie in [0, 1]");
    if (n_total <= 0 || n_total >= (int64_t(1) << 31))
        return fail(IGX_E_SIZE, "igx_quantile_window_begin: n_total must lie in [1, 2^31)");
    hipLaunchKernelGGL(window_begin_kernel, dim3(8), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<WindowState *>(workspace), q, (unsigned long long)n_total, (int)reset_window,
                       static_cast<unsigned long long *>(nullptr));
    return launch_status("igx_quantile_window_begin");
}

extern "C" int igx_quantile_window_histogram(const double *x, int64_t n, void *workspace, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_window_histogram");
    if (rc != IGX_OK) return rc;
    if (n < 0) return fail(IGX_E_SIZE, "igx_quantile_window_histogram: negative n");
    if (n > 0 && !x) return fail(IGX_E_NULL, "igx_quantile_window_histogram: x is NULL");
    static const int per_thread = tuning_int("IGX_QUANTILE_WINDOW_PER_THREAD", 16);
    int64_t blocks = (n + 1024 * (int64_t)per_thread - 1) / (1024 * (int64_t)per_thread);
    blocks = blocks < 1 ? 1 : blocks > 512 ? 512 : blocks;
    hipLaunchKernelGGL(window_hist_kernel, dim3((unsigned)blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), x, n,
                       static_cast<WindowState *>(workspace));
    return launch_status("igx_quantile_window_histogram");
}

extern "C" int igx_quantile_window_locate(void *workspace, int32_t capacity, int64_t *decision, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_window_locate");
    if (rc != IGX_OK) return rc;
    if (!decision) return fail(IGX_E_NULL, "igx_quantile_window_locate: decision is NULL");
    if (capacity < 1 || capacity > IGX_QUANTILE_WINDOW_MAX_CANDIDATES)
        return fail(IGX_E_PARAM, "igx_quantile_window_locate: capacity %d outside [1, %d]", capacity,
                    IGX_QUANTILE_WINDOW_MAX_CANDIDATES);
    hipLaunchKernelGGL(window_locate_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       static_cast<WindowState *>(workspace), (unsigned)capacity, reinterpret_cast<long long *>(decision));
    return launch_status("igx_quantile_window_locate");
}

extern "C" int igx_quantile_window_collect(const double *x, int64_t n, const void *workspace, uint64_t *slots,
                                           int64_t num_slots, igx_stream_t stream)
{
    int rc = check_ws(const_cast<void *>(workspace), "igx_quantile_window_collect");
    if (rc != IGX_OK) return rc;
    if (!slots) return fail(IGX_E_NULL, "igx_quantile_window_collect: slots is NULL");
    if (num_slots < 2 || num_slots > IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2)
        return fail(IGX_E_SIZE, "igx_quantile_window_collect: num_slots %lld outside [2, %d]", (long long)num_slots,
                    IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2);
    if (n < 0) return fail(IGX_E_SIZE, "igx_quantile_window_collect: negative n");
    if (n > 0 && !x) return fail(IGX_E_NULL, "igx_quantile_window_collect: x is NULL");
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned long long *sl = reinterpret_cast<unsigned long long *>(slots);
    hipLaunchKernelGGL(window_slots_init_kernel, dim3(1), dim3(64), 0, st, sl);
    int64_t blocks = (n + 1024 * 16 - 1) / (1024 * 16);
    blocks = blocks < 1 ? 1 : blocks > 512 ? 512 : blocks;
    hipLaunchKernelGGL(window_collect_kernel, dim3((unsigned)blocks), dim3(1024), 0, st, x, n,
                       static_cast<const WindowState *>(workspace), sl, (long long)num_slots);
    return launch_status("igx_quantile_window_collect");
}

extern "C" int igx_quantile_window_finish(void *workspace, const uint64_t *gathered, int32_t num_ranks, int64_t num_slots,
                                          double *out, igx_stream_t stream)
{
    int rc = check_ws(workspace, "igx_quantile_window_finish");
    if (rc != IGX_OK) return rc;
    if (!gathered) return fail(IGX_E_NULL, "igx_quantile_window_finish: gathered is NULL");
    if (num_ranks < 1 || num_ranks > IGX_QUANTILE_WINDOW_MAX_RANKS)
        return fail(IGX_E_PARAM, "igx_quantile_window_finish: num_ranks %d outside [1, %d]", num_ranks,
                    IGX_QUANTILE_WINDOW_MAX_RANKS);
    if (num_slots < 2 || num_slots > IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2)
        return fail(IGX_E_SIZE, "igx_quantile_window_finish: num_slots %lld outside [2, %d]", (long long)num_slots,
                    IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2);
    hipLaunchKernelGGL(window_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream),
                       static_cast<WindowState *>(workspace), reinterpret_cast<const unsigned long long *>(gathered),
                       (int)num_ranks, (long long)num_slots, out, static_cast<const double *>(nullptr), (int64_t)0);
    return launch_status("igx_quantile_window_finish");
}

// layout of the host-free select inside a workspace of IGX_QUANTILE_WORKSPACE_BYTES: the 512-byte header of the
// multi-pass select (its result field is where igx_ant_imitate_finish reads), the window state and histogram, the slots
constexpr size_t HF_WINDOW = 512, HF_SLOTS = HF_WINDOW + IGX_QUANTILE_WINDOW_WORKSPACE_BYTES;
constexpr long long HF_NUM_SLOTS = IGX_QUANTILE_WINDOW_MAX_CANDIDATES + 2;
static_assert(HF_SLOTS % 16 == 0 && HF_SLOTS + 8 * HF_NUM_SLOTS <= IGX_QUANTILE_WORKSPACE_BYTES, "fits the select's workspace");

extern "C" int igx_quantile_f64_windowed(const double *x, int64_t n, double q, double *out, void *workspace,
                                         igx_stream_t stream)
{
    const char *who = "igx_quantile_f64_windowed";
    if (!out) return fail(IGX_E_NULL, "%s: out is NULL", who);
    if (n <= QSMALL || n >= (int64_t(1) << 31)) return igx_quantile_f64(x, n, q, out, workspace, stream);
    int rc = check_ws(workspace, who);
    if (rc != IGX_OK) return rc;
    if (!x) return fail(IGX_E_NULL, "%s: x is NULL", who);
    if (!(q >= 0.0 && q <= 1.0)) return fail(IGX_E_PARAM, "%s: q must lie in [0, 1]", who);
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *base = static_cast<char *>(workspace);
    WindowState *ws = reinterpret_cast<WindowState *>(base + HF_WINDOW);
    unsigned long long *slots = reinterpret_cast<unsigned long long *>(base + HF_SLOTS);
    static const int per_thread = tuning_int("IGX_QUANTILE_WINDOW_PER_THREAD", 16);
    int64_t hb = (n + 1024 * (int64_t)per_thread - 1) / (1024 * (int64_t)per_thread);
    hb = hb < 1 ? 1 : hb > 512 ? 512 : hb;
    int64_t cb = (n + 1024 * 16 - 1) / (1024 * 16);
    cb = cb < 1 ? 1 : cb > 512 ? 512 : cb;
    hipLaunchKernelGGL(window_begin_kernel, dim3(8), dim3(256), 0, st, ws, q, (unsigned long long)n, 0, slots);
    for (int round = 0; round < 2; ++round) {  // the second round returns at once when the first one located the bin
        hipLaunchKernelGGL(window_hist_kernel, dim3((unsigned)hb), dim3(1024), 0, st, x, n, ws);
        hipLaunchKernelGGL(window_locate_kernel, dim3(1), dim3(1024), 0, st, ws,
                           (unsigned)IGX_QUANTILE_WINDOW_MAX_CANDIDATES, static_cast<long long *>(nullptr));
    }
    hipLaunchKernelGGL(window_collect_kernel, dim3((unsigned)cb), dim3(1024), 0, st, x, n, ws, slots, HF_NUM_SLOTS);
    hipLaunchKernelGGL(window_finish_kernel, dim3(1), dim3(1024), 0, st, ws, slots, 1, HF_NUM_SLOTS, out, x, n);
    return launch_status(who);
}
