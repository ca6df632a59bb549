// Access-pattern skeleton of the Ant step (same arrays, same bytes per env,
// no math): what launch shape streams this multi-array pattern fastest?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string>
#include <vector>
using v4f = float __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Bufs { const float *root, *dof, *sens, *act; float *pot, *prev, *up, *head, *rew, *obs; int64_t *reset, *prog, *rnd; };

// TPB threads per block, each wave handles its own 64-env tile(s); PERSIST: grid-stride over tiles with prefetch
template <int TPB, bool PERSIST, bool NT>
__global__ __launch_bounds__(TPB) void skel(Bufs b, int64_t ntiles)
{
    const int lane = threadIdx.x & 63;
    const int wib = threadIdx.x >> 6;
    constexpr int WPB = TPB / 64;
    int64_t tile = (int64_t)blockIdx.x * WPB + wib;
    const int64_t tstride = (int64_t)gridDim.x * WPB;
    auto LD = [&](const v4f *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ST = [&](v4f *p, v4f v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    v4f r[16]; float pot; int64_t rs, pg, rn;
    auto load = [&](int64_t t) {
        const int64_t base = t * 64;
        const v4f *g;
        g = (const v4f *)(b.root + base * 13);
        r[0] = LD(g + lane); r[1] = LD(g + 64 + lane); r[2] = LD(g + 128 + lane); r[3] = LD(g + 192 + (lane & 15));
        g = (const v4f *)(b.dof + base * 16);
#pragma unroll
        for (int i = 0; i < 4; ++i) r[4 + i] = LD(g + 64 * i + lane);
        g = (const v4f *)(b.sens + base * 24);
#pragma unroll
        for (int i = 0; i < 6; ++i) r[8 + i] = LD(g + 64 * i + lane);
        g = (const v4f *)(b.act + base * 8);
        r[14] = LD(g + lane); r[15] = LD(g + 64 + lane);
        pot = b.pot[base + lane]; rs = b.reset[base + lane]; pg = b.prog[base + lane]; rn = b.rnd[base + lane];
    };
    if (tile < ntiles) load(tile);
    while (tile < ntiles) {
        const int64_t base = tile * 64;
        v4f o[15];
#pragma unroll
        for (int i = 0; i < 15; ++i) o[i] = r[i] + r[15];
        float p0 = pot; int64_t a = rs, c = pg, d = rn;
        v4f u = r[3], h = r[7];
        if (PERSIST) { int64_t nx = tile + tstride; if (nx < ntiles) load(nx); }
        v4f *og = (v4f *)(b.obs + base * 60);
#pragma unroll
        for (int i = 0; i < 15; ++i) ST(og + 64 * i + lane, o[i]);
        b.pot[base + lane] = p0 + 1.0f; b.prev[base + lane] = p0; b.rew[base + lane] = p0 * 2.0f;
        b.reset[base + lane] = a ^ 1; b.prog[base + lane] = c + 1; b.rnd[base + lane] = d + 1;
        if (lane < 48) { ST((v4f *)(b.up + base * 3) + lane, u); ST((v4f *)(b.head + base * 3) + lane, h); }
        if (!PERSIST) break;
        tile += tstride;
    }
}

// one-shot, obs rows written as three partial-row segments per env
// (cols 0..27 | sensors 28..51 | actions 52..59) instead of linear full rows
template <int TPB, bool NT>
__global__ __launch_bounds__(TPB) void skel_split(Bufs b, int64_t ntiles)
{
    const int lane = threadIdx.x & 63;
    const int64_t tile = (int64_t)blockIdx.x * (TPB / 64) + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    auto LD = [&](const v4f *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto ST = [&](v4f *p, v4f v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    v4f r[16];
    const int64_t base = tile * 64;
    const v4f *g;
    g = (const v4f *)(b.root + base * 13);
    r[0] = LD(g + lane); r[1] = LD(g + 64 + lane); r[2] = LD(g + 128 + lane); r[3] = LD(g + 192 + (lane & 15));
    g = (const v4f *)(b.dof + base * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[4 + i] = LD(g + 64 * i + lane);
    g = (const v4f *)(b.sens + base * 24);
#pragma unroll
    for (int i = 0; i < 6; ++i) r[8 + i] = LD(g + 64 * i + lane);
    g = (const v4f *)(b.act + base * 8);
    r[14] = LD(g + lane); r[15] = LD(g + 64 + lane);
    float pot = b.pot[base + lane]; int64_t rs = b.reset[base + lane], pg = b.prog[base + lane], rn = b.rnd[base + lane];
    v4f *og = (v4f *)(b.obs + base * 60);
    // sensors: chunk k = lane + 64 i -> env k/6, part k%6, at float4 index env*15 + 7 + part
#pragma unroll
    for (int i = 0; i < 6; ++i) { int k = lane + 64 * i; int env = k / 6, part = k - env * 6; ST(og + env * 15 + 7 + part, r[8 + i] * 0.1f); }
    // cols 0..27: 7 float4 per env = 448 chunks
#pragma unroll
    for (int i = 0; i < 7; ++i) { int k = lane + 64 * i; int env = k / 7, part = k - env * 7; ST(og + env * 15 + part, r[i] + r[7]); }
    // actions: 2 float4 per env
#pragma unroll
    for (int i = 0; i < 2; ++i) { int k = lane + 64 * i; int env = k >> 1, part = k & 1; ST(og + env * 15 + 13 + part, r[14 + i]); }
    b.pot[base + lane] = pot + 1.0f; b.prev[base + lane] = pot; b.rew[base + lane] = pot * 2.0f;
    b.reset[base + lane] = rs ^ 1; b.prog[base + lane] = pg + 1; b.rnd[base + lane] = rn + 1;
    if (lane < 48) { ST((v4f *)(b.up + base * 3) + lane, r[3]); ST((v4f *)(b.head + base * 3) + lane, r[7]); }
}

// one-shot, 64-thread blocks, the sixteen 16-byte loads issued by inline assembly with a given
// cache-policy suffix (POL: 0 nt, 1 sc1 nt, 2 sc0 sc1 nt, 3 sc1, 4 sc0 sc1, 5 none); stores nt
#define LOAD_POL(dst, ptr, pol) asm volatile("global_load_dwordx4 %0, %1, off " pol : "=v"(dst) : "v"(ptr) : "memory")
template <int POL>
__global__ __launch_bounds__(64) void skel_pol(Bufs b, int64_t ntiles)
{
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t base = tile * 64;
    v4f r[16];
    const v4f *p[16];
    const v4f *g = (const v4f *)(b.root + base * 13);
    p[0] = g + lane; p[1] = g + 64 + lane; p[2] = g + 128 + lane; p[3] = g + 192 + (lane & 15);
    g = (const v4f *)(b.dof + base * 16);
    for (int i = 0; i < 4; ++i) p[4 + i] = g + 64 * i + lane;
    g = (const v4f *)(b.sens + base * 24);
    for (int i = 0; i < 6; ++i) p[8 + i] = g + 64 * i + lane;
    g = (const v4f *)(b.act + base * 8);
    p[14] = g + lane; p[15] = g + 64 + lane;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (POL == 0) LOAD_POL(r[i], p[i], "nt");
        else if (POL == 1) LOAD_POL(r[i], p[i], "sc1 nt");
        else if (POL == 2) LOAD_POL(r[i], p[i], "sc0 sc1 nt");
        else if (POL == 3) LOAD_POL(r[i], p[i], "sc1");
        else if (POL == 4) LOAD_POL(r[i], p[i], "sc0 sc1");
        else LOAD_POL(r[i], p[i], "");
    }
    float pot = b.pot[base + lane]; int64_t rs = b.reset[base + lane], pg = b.prog[base + lane], rn = b.rnd[base + lane];
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]),
                 "+v"(r[8]), "+v"(r[9]), "+v"(r[10]), "+v"(r[11]), "+v"(r[12]), "+v"(r[13]), "+v"(r[14]), "+v"(r[15]));
    v4f *og = (v4f *)(b.obs + base * 60);
#pragma unroll
    for (int i = 0; i < 15; ++i) __builtin_nontemporal_store(r[i] + r[15], og + 64 * i + lane);
    b.pot[base + lane] = pot + 1.0f; b.prev[base + lane] = pot; b.rew[base + lane] = pot * 2.0f;
    b.reset[base + lane] = rs ^ 1; b.prog[base + lane] = pg + 1; b.rnd[base + lane] = rn + 1;
    if (lane < 48) { __builtin_nontemporal_store(r[3], (v4f *)(b.up + base * 3) + lane); __builtin_nontemporal_store(r[7], (v4f *)(b.head + base * 3) + lane); }
}

// Humanoid access pattern (root 13, dof_state 42, dof_force 21, sensors 12, actions 21 floats in; 108-wide
// rows out; the same per-env scalars), T rows per 64-lane wave, no math and no LDS: every input chunk is
// loaded, folded into one value and the output chunks are stored.
struct HBufs { const float *root, *dof, *force, *sens, *act; float *pot, *prev, *up, *head, *rew, *obs; int64_t *reset, *prog, *rnd; };
// the same streams, but the loaded chunks go through an LDS image of the output tile (vector writes, a
// wave barrier, vector reads) before they are stored: the staging structure of the real kernel without its math
template <int T, int VAR = 0>
__global__ __launch_bounds__(64) void skel_hum_lds(HBufs b, int64_t ntiles)
{
    __shared__ __attribute__((aligned(16))) float s_tile[T * 108];
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t base = tile * T;
    constexpr int O4 = T * 108 / 4;
    int slot = lane;
    float pot1 = 0; int64_t rs1 = 0, pg1 = 0, rn1 = 0;
    if (VAR & 1) {
        const int e1 = lane < T ? lane : 0;
        pg1 = __builtin_nontemporal_load(b.prog + base + e1); rs1 = __builtin_nontemporal_load(b.reset + base + e1);
        pot1 = __builtin_nontemporal_load(b.pot + base + e1); rn1 = __builtin_nontemporal_load(b.rnd + base + e1);
    }
    if (VAR & 4) {  // first, default policy
        const int e1 = lane < T ? lane : 0;
        pg1 = b.prog[base + e1]; rs1 = b.reset[base + e1]; pot1 = b.pot[base + e1]; rn1 = b.rnd[base + e1];
    }
    auto stream_in = [&](const float *g, int row) {
        const v4f *g4 = (const v4f *)(g + base * row);
        const int n4 = T * row / 4;
        v4f r[11];
        const int iters = (n4 + 63) / 64;
        for (int k = 0; k < iters; ++k) { const int i = lane + 64 * k; r[k] = __builtin_nontemporal_load(g4 + (i < n4 ? i : n4 - 1)); }
        for (int k = 0; k < iters; ++k) { if (slot < O4) *(v4f *)(s_tile + 4 * slot) = r[k]; slot += 64; }
    };
    stream_in(b.root, 13); stream_in(b.dof, 42); stream_in(b.force, 21); stream_in(b.sens, 12); stream_in(b.act, 21);
    const int e = lane < T ? lane : 0;
    float pot; int64_t rs, pg, rn;
    if (VAR & 5) { pot = pot1; rs = rs1; pg = pg1; rn = rn1; }
    else if (VAR & 8) {  // last, non-temporal
        pot = __builtin_nontemporal_load(b.pot + base + e); rs = __builtin_nontemporal_load(b.reset + base + e);
        pg = __builtin_nontemporal_load(b.prog + base + e); rn = __builtin_nontemporal_load(b.rnd + base + e);
    } else { pot = b.pot[base + e]; rs = b.reset[base + e]; pg = b.prog[base + e]; rn = b.rnd[base + e]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    v4f *og = (v4f *)(b.obs + base * 108);
    const v4f acc = *(const v4f *)(s_tile + 4 * lane);
    if ((VAR & 16) && lane < T) {  // the per-environment scalars stored ahead of the wide drain
        __builtin_nontemporal_store(pot + acc.x, b.pot + base + lane); __builtin_nontemporal_store(pot, b.prev + base + lane);
        __builtin_nontemporal_store(pot * 2.0f, b.rew + base + lane); __builtin_nontemporal_store(rs ^ 1, b.reset + base + lane);
        __builtin_nontemporal_store(pg + 1, b.prog + base + lane); __builtin_nontemporal_store(rn + 1, b.rnd + base + lane);
    }
#pragma unroll
    for (int k = 0; k < (O4 + 63) / 64; ++k) {
        const int i = lane + 64 * k;
        if (i < O4) __builtin_nontemporal_store(*(const v4f *)(s_tile + 4 * i), og + i);
    }
    if (VAR & 16) {
    } else if (lane < T) {
        if (VAR & 2) {
            __builtin_nontemporal_store(pot + acc.x, b.pot + base + lane); __builtin_nontemporal_store(pot, b.prev + base + lane);
            __builtin_nontemporal_store(pot * 2.0f, b.rew + base + lane); __builtin_nontemporal_store(rs ^ 1, b.reset + base + lane);
            __builtin_nontemporal_store(pg + 1, b.prog + base + lane); __builtin_nontemporal_store(rn + 1, b.rnd + base + lane);
        } else {
            b.pot[base + lane] = pot + acc.x; b.prev[base + lane] = pot; b.rew[base + lane] = pot * 2.0f;
            b.reset[base + lane] = rs ^ 1; b.prog[base + lane] = pg + 1; b.rnd[base + lane] = rn + 1;
        }
    }
    if (lane < T * 3 / 4) { __builtin_nontemporal_store(acc, (v4f *)(b.up + base * 3) + lane); __builtin_nontemporal_store(acc, (v4f *)(b.head + base * 3) + lane); }
}

template <int T, int DELAY = 0>
__global__ __launch_bounds__(64) void skel_hum(HBufs b, int64_t ntiles)
{
    const int lane = threadIdx.x;
    const int64_t tile = blockIdx.x;
    if (tile >= ntiles) return;
    const int64_t base = tile * T;
    v4f acc = {0, 0, 0, 0};
    auto stream_in = [&](const float *g, int row) {
        const v4f *g4 = (const v4f *)(g + base * row);
        const int n4 = T * row / 4;
#pragma unroll
        for (int k = 0; k < (n4 + 63) / 64; ++k) {
            const int i = lane + 64 * k;
            acc += __builtin_nontemporal_load(g4 + (i < n4 ? i : n4 - 1));
        }
    };
    stream_in(b.root, 13); stream_in(b.dof, 42); stream_in(b.force, 21); stream_in(b.sens, 12); stream_in(b.act, 21);
    if (DELAY) {  // a dependent chain of DELAY fused multiply-adds between the loads and the stores: the compute phase
        float x = acc.x;
#pragma unroll 8
        for (int i = 0; i < DELAY; ++i) x = __builtin_fmaf(x, 1.0000001f, 0.5f);
        acc.x = x;
    }
    const int e = lane < T ? lane : 0;
    const float pot = b.pot[base + e];
    const int64_t rs = b.reset[base + e], pg = b.prog[base + e], rn = b.rnd[base + e];
    v4f *og = (v4f *)(b.obs + base * 108);
    constexpr int O4 = T * 108 / 4;
#pragma unroll
    for (int k = 0; k < (O4 + 63) / 64; ++k) {
        const int i = lane + 64 * k;
        if (i < O4) __builtin_nontemporal_store(acc, og + i);
    }
    if (lane < T) {
        b.pot[base + lane] = pot + acc.x; b.prev[base + lane] = pot; b.rew[base + lane] = pot * 2.0f;
        b.reset[base + lane] = rs ^ 1; b.prog[base + lane] = pg + 1; b.rnd[base + lane] = rn + 1;
    }
    if (lane < T * 3 / 4) { __builtin_nontemporal_store(acc, (v4f *)(b.up + base * 3) + lane); __builtin_nontemporal_store(acc, (v4f *)(b.head + base * 3) + lane); }
}

template <class F>
static double time_ms(F f, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f(i);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters;
}

// argv[1] = "random": fill every buffer with random finite floats instead of zeros
// (the data pattern changes what the memory system sustains)
int main(int argc, char **argv)
{
    const int64_t N = 1 << 20, ntiles = N / 64;
    const int NS = 4;
    const bool random = argc > 1 && std::string(argv[1]) == "random";
    std::vector<uint32_t> host;
    if (random) {
        host.resize((size_t)N * 60 + 1024);
        uint64_t x = 0x9E3779B97F4A7C15ull;
        for (auto &w : host) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            w = ((uint32_t)(x >> 32) & 0x007FFFFFu) | 0x3F000000u | ((uint32_t)x & 0x80000000u);  // +-[0.5, 1)
        }
    }
    Bufs s[NS];
    for (int i = 0; i < NS; ++i) {
        auto A = [&](size_t bytes) {
            void *p; CHECK(hipMalloc(&p, bytes));
            if (random) CHECK(hipMemcpy(p, (const char *)host.data() + 4 * ((i * 977 + bytes / 64) % 1024), bytes, hipMemcpyHostToDevice));
            else CHECK(hipMemset(p, 0, bytes));
            return p; };
        s[i].root = (float *)A(N * 13 * 4); s[i].dof = (float *)A(N * 16 * 4); s[i].sens = (float *)A(N * 24 * 4); s[i].act = (float *)A(N * 8 * 4);
        s[i].pot = (float *)A(N * 4); s[i].prev = (float *)A(N * 4); s[i].up = (float *)A(N * 12); s[i].head = (float *)A(N * 12); s[i].rew = (float *)A(N * 4);
        s[i].obs = (float *)A(N * 240); s[i].reset = (int64_t *)A(N * 8); s[i].prog = (int64_t *)A(N * 8); s[i].rnd = (int64_t *)A(N * 8);
    }
    const double bytes = 556.0 * N;
    printf("{\"data\": \"%s\", ", random ? "random" : "zeros");
    bool first = true;
    auto emit = [&](const std::string &name, double ms) { printf("%s\"%s\": [%.1f, %.0f]", first ? "" : ", ", name.c_str(), ms * 1e3, bytes / ms / 1e6); first = false; fflush(stdout); };
#define RUN(TPB, PERSIST, NT, GRID, NAME) emit(NAME, time_ms([&](int i) { hipLaunchKernelGGL((skel<TPB, PERSIST, NT>), dim3(GRID), dim3(TPB), 0, 0, s[i % NS], ntiles); }, 30))
    RUN(64, false, false, (unsigned)ntiles, "oneshot_w64");
    RUN(64, false, true, (unsigned)ntiles, "oneshot_w64_nt");
    RUN(256, false, false, (unsigned)(ntiles / 4), "oneshot_b256");
    RUN(256, false, true, (unsigned)(ntiles / 4), "oneshot_b256_nt");
    RUN(512, false, false, (unsigned)(ntiles / 8), "oneshot_b512");
    RUN(1024, false, false, (unsigned)(ntiles / 16), "oneshot_b1024");
    for (int lds : {8000, 16000, 28000, 40000})  // fewer resident waves through an LDS reservation
        emit("oneshot_w64_nt_lds" + std::to_string(lds), time_ms([&](int i) { hipLaunchKernelGGL((skel<64, false, true>), dim3((unsigned)ntiles), dim3(64), lds, 0, s[i % NS], ntiles); }, 30));
#define RUNS(TPB, NT, GRID, NAME) emit(NAME, time_ms([&](int i) { hipLaunchKernelGGL((skel_split<TPB, NT>), dim3(GRID), dim3(TPB), 0, 0, s[i % NS], ntiles); }, 30))
    RUNS(64, true, (unsigned)ntiles, "split_w64_nt");
    RUNS(256, true, (unsigned)(ntiles / 4), "split_b256_nt");
    RUNS(256, false, (unsigned)(ntiles / 4), "split_b256");
#define RUNP(POL, NAME) emit(NAME, time_ms([&](int i) { hipLaunchKernelGGL((skel_pol<POL>), dim3((unsigned)ntiles), dim3(64), 0, 0, s[i % NS], ntiles); }, 30))
    RUNP(0, "loads_nt"); RUNP(1, "loads_sc1_nt"); RUNP(2, "loads_sc0_sc1_nt"); RUNP(3, "loads_sc1"); RUNP(4, "loads_sc0_sc1"); RUNP(5, "loads_default");
    RUNP(0, "loads_nt_again");
    RUN(64, true, false, 2048, "persist_w64_g2048");
    RUN(64, true, false, 4096, "persist_w64_g4096");
    RUN(64, true, false, 8192, "persist_w64_g8192");
    RUN(256, true, false, 512, "persist_b256_g512");
    RUN(256, true, false, 1024, "persist_b256_g1024");
    RUN(256, true, false, 2048, "persist_b256_g2048");
    RUN(256, true, true, 2048, "persist_b256_g2048_nt");
    {   // Humanoid pattern: 960 B per env with randomize both ways (940 in the model)
        HBufs h[NS];
        for (int i = 0; i < NS; ++i) {
            auto A = [&](size_t bytes) {
                void *p; CHECK(hipMalloc(&p, bytes));
                if (random) CHECK(hipMemcpy(p, (const char *)host.data() + 4 * ((i * 977 + bytes / 64) % 1024), bytes < host.size() * 4 - 4096 ? bytes : host.size() * 4 - 4096, hipMemcpyHostToDevice));
                else CHECK(hipMemset(p, 0, bytes));
                return p; };
            h[i].root = (float *)A(N * 13 * 4); h[i].dof = (float *)A(N * 42 * 4); h[i].force = (float *)A(N * 21 * 4);
            h[i].sens = (float *)A(N * 12 * 4); h[i].act = (float *)A(N * 21 * 4);
            h[i].pot = (float *)A(N * 4); h[i].prev = (float *)A(N * 4); h[i].up = (float *)A(N * 12); h[i].head = (float *)A(N * 12);
            h[i].rew = (float *)A(N * 4); h[i].obs = (float *)A(N * 432); h[i].reset = (int64_t *)A(N * 8); h[i].prog = (int64_t *)A(N * 8);
            h[i].rnd = (int64_t *)A(N * 8);
        }
        const double hbytes = 940.0 * N;
        auto emith = [&](const std::string &name, double ms) { printf(", \"%s\": [%.1f, %.0f]", name.c_str(), ms * 1e3, hbytes / ms / 1e6); fflush(stdout); };
        emith("humanoid_T32", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<32>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        for (int lds : {7000, 10000, 14000, 20000, 28000})
            emith("humanoid_T32_lds" + std::to_string(lds), time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<32>), dim3((unsigned)(N / 32)), dim3(64), lds, 0, h[i % NS], N / 32); }, 30));
        for (int lds : {0, 10000, 14000, 20000})
            emith("humanoid_T32_delay2000_lds" + std::to_string(lds), time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<32, 2000>), dim3((unsigned)(N / 32)), dim3(64), lds, 0, h[i % NS], N / 32); }, 30));
        for (int lds : {0, 14000})
            emith("humanoid_T32_delay4000_lds" + std::to_string(lds), time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<32, 4000>), dim3((unsigned)(N / 32)), dim3(64), lds, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 0>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_flags_first_nt", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 1>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_scalar_stores_nt", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 2>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_both", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 3>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_flags_first_plain", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 4>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_flags_last_nt", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 8>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        {   // outputs and per-environment state shared by all input sets, as a task that steps in place has them
            HBufs g[NS];
            for (int i = 0; i < NS; ++i) { g[i] = h[i]; g[i].pot = h[0].pot; g[i].prev = h[0].prev; g[i].up = h[0].up; g[i].head = h[0].head;
                                           g[i].rew = h[0].rew; g[i].obs = h[0].obs; g[i].reset = h[0].reset; g[i].prog = h[0].prog; g[i].rnd = h[0].rnd; }
            emith("humanoid_T32_through_lds_shared_outputs", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 0>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, g[i % NS], N / 32); }, 30));
        }
        emith("humanoid_T32_through_lds_scalars_before_drain", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 16>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_scalars_before_drain_flags_first", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 17>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T32_through_lds_again", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum_lds<32, 0>), dim3((unsigned)(N / 32)), dim3(64), 0, 0, h[i % NS], N / 32); }, 30));
        emith("humanoid_T64", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<64>), dim3((unsigned)(N / 64)), dim3(64), 0, 0, h[i % NS], N / 64); }, 30));
        emith("humanoid_T16", time_ms([&](int i) { hipLaunchKernelGGL((skel_hum<16>), dim3((unsigned)(N / 16)), dim3(64), 0, 0, h[i % NS], N / 16); }, 30));
    }
    printf("}\n");
    return 0;
}
