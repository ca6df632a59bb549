// Building blocks shared by the per-task step kernels: a wavefront owns a
// tile of 64 consecutive environments, streams the tile's contiguous slice of
// every AoS input with 16-byte non-temporal accesses, and assembles the tile's
// observation rows in LDS so that they leave as whole contiguous lines.
#pragma once
#include "igx_device.hpp"

namespace igx {

constexpr int WAVE = 64;
// native vectors: copies of these are real vector loads / stores (a struct
// copy would become a memcpy through a private-memory temporary)
using v4f = float __attribute__((ext_vector_type(4)));
using v2f = float __attribute__((ext_vector_type(2)));

// 16-byte vector at a 4-byte-aligned address (global memory takes dword-aligned
// multi-dword accesses): rows whose length is not a multiple of 4 floats
using v4f_u = v4f __attribute__((aligned(4)));

// Streaming (non-temporal) global accesses: every byte of this path is touched
// exactly once per step, so nothing is worth keeping in L2; on MI355X the nt
// forms stream ~20 % faster than default-policy loads / stores here.
template <class T>
__device__ __forceinline__ T ld_stream(const T *p) { return __builtin_nontemporal_load(p); }
template <class T>
__device__ __forceinline__ void st_stream(T *p, T v) { __builtin_nontemporal_store(v, p); }

// Per-environment scalars (one 4- or 8-byte element per lane: flags, potentials, rewards) take the
// default cache policy.  Measured on MI355X: the nt forms pay off for the 16-byte-per-lane streams only;
// on these narrow accesses they cost the Ant step 4-6 us of 102 and the Humanoid step 9 us of 181.
template <class T>
__device__ __forceinline__ T ld_narrow(const T *p) { return *p; }
template <class T>
__device__ __forceinline__ void st_narrow(T *p, T v) { *p = v; }

__device__ __forceinline__ v4f ld_stream_u(const float *p)
{
    // (16-byte gathers keep the nt policy: plain loads measured 5-12 % slower on the hand, AMP and imitation steps)
    return __builtin_nontemporal_load(reinterpret_cast<const v4f_u *>(p));
}

__device__ __forceinline__ bool dev_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Compiler-level ordering point for the wave-synchronous LDS exchange.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int N4, int ITERS>
__device__ __forceinline__ void load_chunks(const float *__restrict__ g, v4f (&v)[ITERS], int lane)
{
    const v4f *g4 = reinterpret_cast<const v4f *>(g);
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        // the last, partly filled round re-reads the final chunk in its idle
        // lanes (same cache line, no extra traffic) so every load is unconditional
        int k = lane + it * WAVE;
        if ((it + 1) * WAVE > N4) k = (k < N4) ? k : N4 - 1;
        v[it] = ld_stream(g4 + k);
    }
}

// Rows of ROW floats -> columns C0.. of the LDS observation rows (optionally scaled).
template <int ROW, int C0, int NOBS, bool SCALE, int T = WAVE, int ITERS = 1>
__device__ __forceinline__ void stage_rows(const v4f (&v)[ITERS], float *s_tile, float scl, int lane)
{
    constexpr int N4 = T * ROW / 4;  // T rows per tile, WAVE lanes
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int k = lane + it * WAVE;
        if ((it + 1) * WAVE <= N4 || k < N4) {
            v4f x = v[it];
            if (SCALE) { x.x *= scl; x.y *= scl; x.z *= scl; x.w *= scl; }
            if constexpr (ROW % 4 == 0 && C0 % 4 == 0 && NOBS % 4 == 0) {
                constexpr int R4 = ROW / 4;
                const int env = k / R4, part = k - env * R4;
                *reinterpret_cast<v4f *>(s_tile + env * NOBS + C0 + 4 * part) = x;
            } else {
                // one division per chunk; the row boundary inside the chunk is a compare
                const float e[4] = {x.x, x.y, x.z, x.w};
                int env = (4 * k) / ROW, col = 4 * k - env * ROW;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    s_tile[env * NOBS + C0 + col] = e[i];
                    if (++col == ROW) {
                        col = 0;
                        ++env;
                    }
                }
            }
        }
    }
}

// Interleaved (pos, vel) rows: position -> unscale -> column C_POS+j,
// velocity * dof_vel_scale -> column C_VEL+j.
template <int D, int NOBS, int C_POS, int C_VEL, int T = WAVE, int ITERS = 1>
__device__ __forceinline__ void stage_dof_state(const v4f (&v)[ITERS], float *s_tile, const float *s_lim,
                                                float dof_vel_scale, int lane)
{
    constexpr int N4 = T * D * 2 / 4;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int k = lane + it * WAVE;
        if ((it + 1) * WAVE <= N4 || k < N4) {
            const v4f x = v[it];
            const float pos[2] = {x.x, x.z}, vel[2] = {x.y, x.w};
            if constexpr (D % 2 == 0 && C_POS % 2 == 0 && C_VEL % 2 == 0 && NOBS % 2 == 0) {
                // both dofs of this chunk belong to one environment
                const int pr = 2 * k;
                const int env = pr / D, j = pr - env * D;
                const v2f lo = *reinterpret_cast<const v2f *>(s_lim + j);
                const v2f hi = *reinterpret_cast<const v2f *>(s_lim + D + j);
                *reinterpret_cast<v2f *>(s_tile + env * NOBS + C_POS + j) =
                    v2f{unscale(pos[0], lo.x, hi.x), unscale(pos[1], lo.y, hi.y)};
                *reinterpret_cast<v2f *>(s_tile + env * NOBS + C_VEL + j) =
                    v2f{vel[0] * dof_vel_scale, vel[1] * dof_vel_scale};
            } else {
                int env = (2 * k) / D, j = 2 * k - env * D;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    s_tile[env * NOBS + C_POS + j] = unscale(pos[h], s_lim[j], s_lim[D + j]);
                    s_tile[env * NOBS + C_VEL + j] = vel[h] * dof_vel_scale;
                    if (++j == D) {
                        j = 0;
                        ++env;
                    }
                }
            }
        }
    }
}

// torch.sum(dim=-1) over a row of D floats as the ATen CPU kernel orders it:
// 8 vector lanes over the full blocks of 8, tail first, lanes left to right.
template <int D>
__device__ __forceinline__ float row_sum(const float (&x)[D])
{
    constexpr int NB = D / 8;
    if constexpr (NB == 0) {
        float r = x[0];
#pragma unroll
        for (int i = 1; i < D; ++i) r += x[i];
        return r;
    } else {
        float lane[8];
#pragma unroll
        for (int l = 0; l < 8; ++l) {
            float a = x[l];
#pragma unroll
            for (int b = 1; b < NB; ++b) a += x[8 * b + l];
            lane[l] = a;
        }
        constexpr int T = 8 * NB;
        float r;
        if constexpr (T < D) {
            r = x[T];
#pragma unroll
            for (int i = T + 1; i < D; ++i) r += x[i];
            r += lane[0];
        } else {
            r = lane[0];
        }
#pragma unroll
        for (int l = 1; l < 8; ++l) r += lane[l];
        return r;
    }
}

template <int LEN, int C0, int NOBS>
__device__ __forceinline__ void read_cols(const float *row, float (&out)[LEN])
{
    if constexpr (LEN % 4 == 0 && C0 % 4 == 0 && NOBS % 4 == 0) {
#pragma unroll
        for (int i = 0; i < LEN / 4; ++i) {
            const v4f v = *reinterpret_cast<const v4f *>(row + C0 + 4 * i);
            out[4 * i] = v.x; out[4 * i + 1] = v.y; out[4 * i + 2] = v.z; out[4 * i + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int i = 0; i < LEN; ++i) out[i] = row[C0 + i];
    }
}

// Write the first COUNT floats of the LDS tile to global memory as whole
// contiguous lines (full tile) or element by element (partial tile).
template <int ROWW, int T = WAVE>
__device__ __forceinline__ void drain_rows(float *__restrict__ g, const float *s_tile, int nvalid, int lane)
{
    if (nvalid == T) {
        constexpr int N4 = T * ROWW / 4;
        v4f *o4 = reinterpret_cast<v4f *>(g);
#pragma unroll
        for (int it = 0; it < (N4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= N4 || k < N4) st_stream(o4 + k, *reinterpret_cast<const v4f *>(s_tile + 4 * k));
        }
    } else {
        for (int i = lane; i < nvalid * ROWW; i += WAVE) g[i] = s_tile[i];
    }
}

// drain_rows with torch.clamp(x, -clip, clip) applied on the way out
template <int ROWW, int T = WAVE>
__device__ __forceinline__ void drain_rows_clamped(float *__restrict__ g, const float *s_tile, int nvalid, int lane, float clip)
{
    if (nvalid == T && dev_aligned16(g)) {
        constexpr int N4 = T * ROWW / 4;
        v4f *o4 = reinterpret_cast<v4f *>(g);
#pragma unroll
        for (int it = 0; it < (N4 + WAVE - 1) / WAVE; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= N4 || k < N4) {
                v4f v = *reinterpret_cast<const v4f *>(s_tile + 4 * k);
                v.x = clamp1(v.x, -clip, clip); v.y = clamp1(v.y, -clip, clip);
                v.z = clamp1(v.z, -clip, clip); v.w = clamp1(v.w, -clip, clip);
                st_stream(o4 + k, v);
            }
        }
    } else {
        for (int i = lane; i < nvalid * ROWW; i += WAVE) g[i] = clamp1(s_tile[i], -clip, clip);
    }
}

// Park a tile's ROWW-float rows (chunks already in registers for a full tile,
// or straight from global memory for a partial one) linearly at s.
template <int ROWW, int T = WAVE, int ITERS = 1>
__device__ __forceinline__ void park_rows(const v4f (&v)[ITERS], const float *__restrict__ g, float *s, int nvalid,
                                          int lane)
{
    if (nvalid == T) {
        constexpr int N4 = T * ROWW / 4;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int k = lane + it * WAVE;
            if ((it + 1) * WAVE <= N4 || k < N4) *reinterpret_cast<v4f *>(s + 4 * k) = v[it];
        }
    } else {
        for (int i = lane; i < nvalid * ROWW; i += WAVE) s[i] = g[i];
    }
}

}  // namespace igx
