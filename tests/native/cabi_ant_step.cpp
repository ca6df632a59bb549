// The C ABI used the way a non-Python host would use it: plain hipMalloc'd buffers, a stream of
// the caller's own, include/igx.h and nothing else of this repository at compile time.  Runs
// igx_ant_step on synthetic state and compares with the CPU checker (test infrastructure; linked
// here only as the checker).  Prints "cabi ok" and returns 0 on success.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "igx.h"

extern "C" int orc_ant_step(const igx_loco_params *p, const igx_loco_buffers *b, int64_t n);

#define HIP_OK(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));             \
            return 2;                                                                   \
        }                                                                               \
    } while (0)

static uint32_t lcg_state = 12345u;
static float uniform() { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) * (1.0f / 16777216.0f); }

template <class T>
struct Buf {
    std::vector<T> host;
    T *dev = nullptr;
    int alloc(size_t count)
    {
        host.assign(count, T(0));
        HIP_OK(hipMalloc(&dev, count * sizeof(T)));
        return 0;
    }
    int up() { HIP_OK(hipMemcpy(dev, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice)); return 0; }
    int down(std::vector<T> &out)
    {
        out.resize(host.size());
        HIP_OK(hipMemcpy(out.data(), dev, host.size() * sizeof(T), hipMemcpyDeviceToHost));
        return 0;
    }
};

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 100003;
    const int D = 8, S = 24, NOBS = 60;
    igx_loco_params p;
    std::memset(&p, 0, sizeof p);
    p.num_dof = D; p.num_sensor_floats = S; p.dt = 0.0166f; p.dof_vel_scale = 0.2f; p.contact_force_scale = 0.1f;
    p.angular_velocity_scale = 0.1f; p.up_weight = 0.1f; p.heading_weight = 0.5f; p.actions_cost_scale = 0.005f;
    p.energy_cost_scale = 0.05f; p.joints_at_limit_cost_scale = 0.1f; p.termination_height = 0.31f; p.death_cost = -2.0f;
    p.max_episode_length_m1 = 999.0f; p.targets[0] = 1000.0f; p.inv_start_rot[3] = 1.0f; p.basis_vec0[0] = 1.0f;
    p.basis_vec1[2] = 1.0f; p.seed = 7; p.rng_offset = 3; p.clip_obs = 5.0f;
    for (int j = 0; j < D; ++j) { p.dof_limits_lower[j] = -1.0f - 0.01f * j; p.dof_limits_upper[j] = 1.0f + 0.02f * j; }

    Buf<float> root, dof, sens, act, iroot, idof, obs, pot, ppot, up, head, rew, tobj, obsc;
    Buf<int64_t> reset, prog, rnd;
    Buf<uint8_t> tout;
    if (root.alloc(n * 13) || dof.alloc(n * D * 2) || sens.alloc(n * S) || act.alloc(n * D) || iroot.alloc(n * 13) ||
        idof.alloc(n * D) || obs.alloc(n * NOBS) || pot.alloc(n) || ppot.alloc(n) || up.alloc(n * 3) || head.alloc(n * 3) ||
        rew.alloc(n) || tobj.alloc(n) || obsc.alloc(n * NOBS) || reset.alloc(n) || prog.alloc(n) || rnd.alloc(n) || tout.alloc(n))
        return 2;
    for (int64_t e = 0; e < n; ++e) {
        float *r = &root.host[e * 13];
        r[0] = uniform() - 0.5f; r[1] = uniform() - 0.5f; r[2] = 0.25f + 0.6f * uniform();
        float q[4], nn = 0.0f;
        for (int i = 0; i < 4; ++i) { q[i] = uniform() - 0.5f; nn += q[i] * q[i]; }
        nn = std::sqrt(nn) + 1e-6f;
        for (int i = 0; i < 4; ++i) r[3 + i] = q[i] / nn;
        for (int i = 7; i < 13; ++i) r[i] = 2.0f * uniform() - 1.0f;
        for (int j = 0; j < D; ++j) {
            dof.host[(e * D + j) * 2] = -1.1f + 2.3f * uniform();
            dof.host[(e * D + j) * 2 + 1] = 4.0f * uniform() - 2.0f;
            act.host[e * D + j] = 2.0f * uniform() - 1.0f;
        }
        for (int k = 0; k < S; ++k) sens.host[e * S + k] = 20.0f * uniform() - 10.0f;
        iroot.host[e * 13 + 2] = 0.55f; iroot.host[e * 13 + 6] = 1.0f;
        pot.host[e] = -1000.0f / p.dt;
        reset.host[e] = uniform() < 0.02f;
        prog.host[e] = (int64_t)(uniform() * 1000.0f);
    }
    for (auto *b : {&root, &dof, &sens, &act, &iroot, &idof, &pot, &ppot}) if (b->up()) return 2;
    if (reset.up() || prog.up() || rnd.up()) return 2;

    igx_loco_buffers d;
    std::memset(&d, 0, sizeof d);
    d.root_states = root.dev; d.dof_state = dof.dev; d.sensors = sens.dev; d.actions = act.dev;
    d.initial_root_states = iroot.dev; d.initial_dof_pos = idof.dev; d.obs = obs.dev; d.potentials = pot.dev;
    d.prev_potentials = ppot.dev; d.up_vec = up.dev; d.heading_vec = head.dev; d.rew = rew.dev; d.reset = reset.dev;
    d.progress = prog.dev; d.randomize = rnd.dev; d.true_objective = tobj.dev; d.obs_clamped = obsc.dev; d.time_outs = tout.dev;

    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    if (igx_ant_step(nullptr, &d, n, stream) != IGX_E_NULL) { std::fprintf(stderr, "NULL params accepted\n"); return 1; }
    const int rc = igx_ant_step(&p, &d, n, stream);
    if (rc != IGX_OK) { std::fprintf(stderr, "igx_ant_step: %d %s\n", rc, igx_last_error_string()); return 1; }
    HIP_OK(hipStreamSynchronize(stream));

    // the CPU checker on the host copies
    igx_loco_buffers h = d;
    std::vector<float> h_obs(n * NOBS), h_ppot(n), h_up(n * 3), h_head(n * 3), h_rew(n), h_tobj(n), h_obsc(n * NOBS);
    std::vector<uint8_t> h_tout(n);
    h.root_states = root.host.data(); h.dof_state = dof.host.data(); h.sensors = sens.host.data(); h.actions = act.host.data();
    h.initial_root_states = iroot.host.data(); h.initial_dof_pos = idof.host.data(); h.obs = h_obs.data();
    h.potentials = pot.host.data(); h.prev_potentials = h_ppot.data(); h.up_vec = h_up.data(); h.heading_vec = h_head.data();
    h.rew = h_rew.data(); h.reset = reset.host.data(); h.progress = prog.host.data(); h.randomize = rnd.host.data();
    h.true_objective = h_tobj.data(); h.obs_clamped = h_obsc.data(); h.time_outs = h_tout.data();
    if (orc_ant_step(&p, &h, n) != 0) return 1;

    // device buffers against the checker's host arrays
    auto mismatches = [&](int64_t &bad) -> int {
        std::vector<float> g_obs, g_rew, g_pot, g_obsc, g_dof;
        std::vector<int64_t> g_reset, g_prog, g_rnd;
        std::vector<uint8_t> g_tout;
        if (obs.down(g_obs) || rew.down(g_rew) || pot.down(g_pot) || obsc.down(g_obsc) || dof.down(g_dof) || reset.down(g_reset) ||
            prog.down(g_prog) || rnd.down(g_rnd) || tout.down(g_tout))
            return 2;
        bad = 0;
        for (int64_t e = 0; e < n; ++e) {
            bad += g_reset[e] != reset.host[e] || g_prog[e] != prog.host[e] || g_rnd[e] != rnd.host[e] || g_tout[e] != h_tout[e];
            bad += std::memcmp(&g_rew[e], &h_rew[e], 4) != 0 || std::memcmp(&g_pot[e], &pot.host[e], 4) != 0;
            for (int c = 0; c < NOBS; ++c) {
                const float a = g_obs[e * NOBS + c], w = h_obs[e * NOBS + c];
                if (c >= 7 && c <= 9) {  // yaw, roll, angle to target: atan2 of the device math library, 1e-5 of pi
                    float diff = std::fabs(a - w);
                    diff = std::fmin(diff, std::fabs(diff - 6.2831855f));
                    bad += !(diff <= 1e-5f * 3.1415927f);
                } else {
                    bad += std::memcmp(&a, &w, 4) != 0;
                }
                const float cl = std::fmin(std::fmax(a, -p.clip_obs), p.clip_obs);
                bad += std::memcmp(&g_obsc[e * NOBS + c], &cl, 4) != 0;
            }
        }
        for (size_t i = 0; i < g_dof.size(); ++i) bad += std::memcmp(&g_dof[i], &dof.host[i], 4) != 0;
        return 0;
    };
    int64_t bad = 0;
    if (mismatches(bad)) return 2;
    if (bad) { std::fprintf(stderr, "cabi: %lld mismatches over %lld envs\n", (long long)bad, (long long)n); return 1; }

    // The same entry point under stream capture: the call only enqueues a kernel, so it records into a
    // HIP graph.  A recorded launch repeats its argument values; the draws of the flagged resets stay
    // fresh through the device-resident counter that the kernel adds to rng_offset.
    uint64_t *counter = nullptr;
    HIP_OK(hipMalloc(&counter, sizeof(uint64_t)));
    igx_loco_params pg = p;
    pg.rng_offset_dev = counter;
    hipGraph_t graph;
    hipGraphExec_t exec;
    HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeGlobal));
    const int rcg = igx_ant_step(&pg, &d, n, stream);
    HIP_OK(hipStreamEndCapture(stream, &graph));
    if (rcg != IGX_OK) { std::fprintf(stderr, "igx_ant_step under capture: %d %s\n", rcg, igx_last_error_string()); return 1; }
    HIP_OK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    for (uint64_t replay = 0; replay < 2; ++replay) {
        const uint64_t count = 5 + replay;
        for (int64_t e = 0; e < n; e += 3) reset.host[e] = 1;  // plenty of resets: the draws must match too
        if (reset.up()) return 2;
        HIP_OK(hipMemcpyAsync(counter, &count, sizeof count, hipMemcpyHostToDevice, stream));
        HIP_OK(hipGraphLaunch(exec, stream));
        HIP_OK(hipStreamSynchronize(stream));
        igx_loco_params ph = p;
        ph.rng_offset = p.rng_offset + count;
        if (orc_ant_step(&ph, &h, n) != 0) return 1;
        if (mismatches(bad)) return 2;
        if (bad) { std::fprintf(stderr, "cabi: graph replay %llu: %lld mismatches\n", (unsigned long long)replay, (long long)bad); return 1; }
    }
    HIP_OK(hipGraphExecDestroy(exec));
    HIP_OK(hipGraphDestroy(graph));
    HIP_OK(hipFree(counter));
    std::printf("cabi ok: igx_ant_step eager and replayed from a graph, %lld envs, version %d\n", (long long)n, igx_version());
    return 0;
}
