// Which difference between the no-math skeleton of the Ant step and the real kernel costs time?
// One-shot 64-lane workgroups, 64 environments each, 16-byte nt loads / stores for the wide
// streams; flags switch on, one at a time, what the real kernel does differently.
//   bit 0: per-environment scalar loads non-temporal          bit 1: per-environment scalar stores non-temporal
//   bit 2: a true_objective stream (4 B / env out)            bit 3: scalar loads issued after the root rows, before the wide streams
//   bit 4: a 640-byte by-value parameter block in the kernel arguments, a few words of it read
//   bit 5: the tile goes through LDS (write, wave barrier, read) before it is stored
// argv[1]: "shared" = every input set writes the same output / state buffers (a task stepping in place)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string>
#include <vector>
using v4f = float __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

struct Bufs { const float *root, *dof, *sens, *act; float *pot, *prev, *up, *head, *rew, *obs, *tobj; int64_t *reset, *prog, *rnd; };
struct Big { float w[160]; };

template <int F>
__global__ __launch_bounds__(64) void skel(Bufs b, Big big, int64_t ntiles)
{
    __shared__ __attribute__((aligned(16))) float s_tile[(F & 32) ? 64 * 60 : 4];
    const int lane = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * 64;
    auto LD = [&](const v4f *p) { return __builtin_nontemporal_load(p); };
    v4f r[16]; float pot; int64_t rs, pg, rn;
    auto scal = [&]() {
        if (F & 1) { pot = __builtin_nontemporal_load(b.pot + base + lane); rs = __builtin_nontemporal_load(b.reset + base + lane);
                     pg = __builtin_nontemporal_load(b.prog + base + lane); rn = __builtin_nontemporal_load(b.rnd + base + lane); }
        else { pot = b.pot[base + lane]; rs = b.reset[base + lane]; pg = b.prog[base + lane]; rn = b.rnd[base + lane]; }
    };
    const v4f *g = (const v4f *)(b.root + base * 13);
    r[0] = LD(g + lane); r[1] = LD(g + 64 + lane); r[2] = LD(g + 128 + lane); r[3] = LD(g + 192 + (lane & 15));
    if (F & 8) { scal(); __builtin_amdgcn_sched_barrier(0); }
    g = (const v4f *)(b.dof + base * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) r[4 + i] = LD(g + 64 * i + lane);
    g = (const v4f *)(b.sens + base * 24);
#pragma unroll
    for (int i = 0; i < 6; ++i) r[8 + i] = LD(g + 64 * i + lane);
    g = (const v4f *)(b.act + base * 8);
    r[14] = LD(g + lane); r[15] = LD(g + 64 + lane);
    if (!(F & 8)) scal();
    v4f o[15];
#pragma unroll
    for (int i = 0; i < 15; ++i) o[i] = r[i] + r[15];
    if (F & 16) o[0].x += big.w[lane & 7] + big.w[100];
    if (F & 32) {
#pragma unroll
        for (int i = 0; i < 15; ++i) *(v4f *)(s_tile + 4 * (lane + 64 * i)) = o[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int i = 0; i < 15; ++i) o[i] = *(const v4f *)(s_tile + 4 * ((lane * 15 + i) % 960));
    }
    v4f *og = (v4f *)(b.obs + base * 60);
#pragma unroll
    for (int i = 0; i < 15; ++i) __builtin_nontemporal_store(o[i], og + 64 * i + lane);
    if (F & 2) {
        __builtin_nontemporal_store(pot + 1.0f, b.pot + base + lane); __builtin_nontemporal_store(pot, b.prev + base + lane);
        __builtin_nontemporal_store(pot * 2.0f, b.rew + base + lane); __builtin_nontemporal_store(rs ^ 1, b.reset + base + lane);
        __builtin_nontemporal_store(pg + 1, b.prog + base + lane); __builtin_nontemporal_store(rn + 1, b.rnd + base + lane);
        if (F & 4) __builtin_nontemporal_store(pot * 3.0f, b.tobj + base + lane);
    } else {
        b.pot[base + lane] = pot + 1.0f; b.prev[base + lane] = pot; b.rew[base + lane] = pot * 2.0f;
        b.reset[base + lane] = rs ^ 1; b.prog[base + lane] = pg + 1; b.rnd[base + lane] = rn + 1;
        if (F & 4) b.tobj[base + lane] = pot * 3.0f;
    }
    if (lane < 48) { __builtin_nontemporal_store(r[3], (v4f *)(b.up + base * 3) + lane); __builtin_nontemporal_store(r[7], (v4f *)(b.head + base * 3) + lane); }
}

template <class F>
static double time_us(F f, int iters)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 5; ++i) f(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) f(i);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms / iters * 1e3;
}

int main(int argc, char **argv)
{
    const int64_t N = 1 << 20, ntiles = N / 64;
    const int NS = 4;
    const bool shared = argc > 1 && std::string(argv[1]) == "shared";
    std::vector<uint32_t> host((size_t)N * 60 + 1024);
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (auto &w : host) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        w = ((uint32_t)(x >> 32) & 0x007FFFFFu) | 0x3F000000u | ((uint32_t)x & 0x80000000u);
    }
    Bufs s[NS];
    for (int i = 0; i < NS; ++i) {
        auto A = [&](size_t bytes) {
            void *p; CHECK(hipMalloc(&p, bytes));
            CHECK(hipMemcpy(p, (const char *)host.data() + 4 * ((i * 977 + bytes / 64) % 1024), bytes, hipMemcpyHostToDevice));
            return p; };
        s[i].root = (float *)A(N * 13 * 4); s[i].dof = (float *)A(N * 16 * 4); s[i].sens = (float *)A(N * 24 * 4); s[i].act = (float *)A(N * 8 * 4);
        s[i].pot = (float *)A(N * 4); s[i].prev = (float *)A(N * 4); s[i].up = (float *)A(N * 12); s[i].head = (float *)A(N * 12); s[i].rew = (float *)A(N * 4);
        s[i].tobj = (float *)A(N * 4);
        s[i].obs = (float *)A(N * 240); s[i].reset = (int64_t *)A(N * 8); s[i].prog = (int64_t *)A(N * 8); s[i].rnd = (int64_t *)A(N * 8);
    }
    if (shared)
        for (int i = 1; i < NS; ++i) { s[i].pot = s[0].pot; s[i].prev = s[0].prev; s[i].up = s[0].up; s[i].head = s[0].head; s[i].rew = s[0].rew;
                                       s[i].tobj = s[0].tobj; s[i].obs = s[0].obs; s[i].reset = s[0].reset; s[i].prog = s[0].prog; s[i].rnd = s[0].rnd; }
    Big big; for (int i = 0; i < 160; ++i) big.w[i] = 0.001f * i;
    printf("{\"outputs\": \"%s\"", shared ? "shared" : "separate");
#define RUN(F, LDS) printf(", \"f%d_lds%d\": %.1f", F, LDS, time_us([&](int i) { hipLaunchKernelGGL((skel<F>), dim3((unsigned)ntiles), dim3(64), LDS, 0, s[i % NS], big, ntiles); }, 100)); fflush(stdout)
    RUN(0, 28000); RUN(1, 28000); RUN(2, 28000); RUN(3, 28000); RUN(4, 28000); RUN(8, 28000); RUN(16, 28000);
    RUN(32, 13000); RUN(63, 13000); RUN(0, 28000); RUN(0, 0); RUN(63, 0); RUN(31, 28000);
    printf("}\n");
    return 0;
}
