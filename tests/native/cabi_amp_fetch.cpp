// Imitation sampling through the C ABI from a plain C++ host: synthetic motion tables uploaded with hipMemcpy,
// a whole fetch_amp_obs_demo per launch (igx_motion_fetch_amp_obs: draws + [n, steps * 105] batch), every batch
// streamed to pinned host memory by a whole-row recorder, on a stream of the caller's own.  Checks the draws and
// the batches against the CPU checker (linked here only as the checker) and the host copies against the device
// batches.  Prints "cabi amp fetch ok" and returns 0 on success.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "igx.h"

extern "C" int orc_motion_sample(const igx_motion_tables *t, const igx_motion_sampler *s, int64_t n, int64_t *motion_ids,
                                 double *motion_times);
extern "C" int orc_motion_amp_obs_steps(const igx_motion_tables *t, const int64_t *ids, const double *times, int64_t n,
                                        int32_t num_steps, double step_dt, int32_t local_root_obs, float *out);

#define HIP_OK(call)                                                                    \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess) {                                                         \
            std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_));             \
            return 2;                                                                   \
        }                                                                               \
    } while (0)
#define IGX_OK_OR_FAIL(call)                                                            \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                 \
            std::fprintf(stderr, "%s -> %d: %s\n", #call, rc_, igx_last_error_string()); \
            return 3;                                                                   \
        }                                                                               \
    } while (0)

static uint32_t lcg_state = 2024u;
static float uniform() { lcg_state = lcg_state * 1664525u + 1013904223u; return (float)(lcg_state >> 8) * (1.0f / 16777216.0f); }
static float gauss() { float s = 0.0f; for (int i = 0; i < 6; ++i) s += uniform(); return (s - 3.0f) * 1.4f; }

template <class T>
static int upload(const std::vector<T> &h, T **d)
{
    HIP_OK(hipMalloc(d, h.size() * sizeof(T)));
    HIP_OK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

// rotations well away from identity (the knife edge of quat_to_angle_axis is a test of its own elsewhere):
// unit quaternions that drift smoothly from frame to frame
static void fill_quats(std::vector<float> &q, int frames, int per_frame)
{
    for (int j = 0; j < per_frame; ++j) {
        float a[4], b[4];
        for (int c = 0; c < 4; ++c) { a[c] = gauss(); b[c] = gauss() * 0.02f; }
        for (int f = 0; f < frames; ++f) {
            float v[4], nrm = 0.0f;
            for (int c = 0; c < 4; ++c) { v[c] = a[c] + b[c] * (float)f; nrm += v[c] * v[c]; }
            nrm = std::sqrt(nrm);
            for (int c = 0; c < 4; ++c) q[((size_t)f * per_frame + j) * 4 + c] = v[c] / nrm;
        }
    }
}

int main(int argc, char **argv)
{
    const int64_t n = argc > 1 ? std::atoll(argv[1]) : 512;
    const int fetches = argc > 2 ? std::atoi(argv[2]) : 6;
    const int K = 2, M = 2, F0 = 90, F1 = 61, F = F0 + F1, NO = 105;
    const double dt_step = 0.0332;

    // ---- tables (host copies double as the checker's input)
    std::vector<double> length = {(F0 - 1) / 30.0, (F1 - 1) / 60.0}, mdt = {1.0 / 30.0, 1.0 / 60.0}, cdf = {0.35, 1.0};
    std::vector<int64_t> nframes = {F0, F1}, offset = {0, F0};
    std::vector<float> root_pos(F * 3), root_rot(F * 4), local_rot((size_t)F * 15 * 4), root_vel(F * 3), root_ang(F * 3),
        key_pos((size_t)F * 12), dof_vel((size_t)F * 28);
    for (auto *v : {&root_pos, &root_vel, &root_ang, &key_pos, &dof_vel})
        for (auto &x : *v) x = gauss() * 0.5f;
    fill_quats(root_rot, F, 1);
    fill_quats(local_rot, F, 15);
    igx_motion_tables host_t, dev_t;
    std::memset(&host_t, 0, sizeof host_t);
    host_t.num_motions = M;
    host_t.motion_length = length.data(); host_t.motion_dt = mdt.data(); host_t.motion_num_frames = nframes.data();
    host_t.frame_offset = offset.data(); host_t.root_pos = root_pos.data(); host_t.root_rot = root_rot.data();
    host_t.local_rot = local_rot.data(); host_t.root_vel = root_vel.data(); host_t.root_ang_vel = root_ang.data();
    host_t.key_pos = key_pos.data(); host_t.dof_vel = dof_vel.data();
    dev_t = host_t;
    double *d_length, *d_mdt, *d_cdf;
    int64_t *d_nframes, *d_offset;
    float *d_rp, *d_rr, *d_lr, *d_rv, *d_ra, *d_kp, *d_dv;
    if (upload(length, &d_length) || upload(mdt, &d_mdt) || upload(cdf, &d_cdf) || upload(nframes, &d_nframes) ||
        upload(offset, &d_offset) || upload(root_pos, &d_rp) || upload(root_rot, &d_rr) || upload(local_rot, &d_lr) ||
        upload(root_vel, &d_rv) || upload(root_ang, &d_ra) || upload(key_pos, &d_kp) || upload(dof_vel, &d_dv))
        return 2;
    dev_t.motion_length = d_length; dev_t.motion_dt = d_mdt; dev_t.motion_num_frames = d_nframes; dev_t.frame_offset = d_offset;
    dev_t.root_pos = d_rp; dev_t.root_rot = d_rr; dev_t.local_rot = d_lr; dev_t.root_vel = d_rv; dev_t.root_ang_vel = d_ra;
    dev_t.key_pos = d_kp; dev_t.dof_vel = d_dv;

    // ---- one batch buffer, the draws, a recorder for whole rows, a stream of our own
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    float *d_batch;
    int64_t *d_ids;
    double *d_times;
    HIP_OK(hipMalloc(&d_batch, (size_t)n * K * NO * sizeof(float)));
    HIP_OK(hipMalloc(&d_ids, (size_t)n * sizeof(int64_t)));
    HIP_OK(hipMalloc(&d_times, (size_t)n * sizeof(double)));
    igx_recorder *rec = nullptr;
    IGX_OK_OR_FAIL(igx_recorder_create(&rec, n, K * NO, nullptr, 0, 4, fetches));

    igx_motion_sampler s;
    std::memset(&s, 0, sizeof s);
    s.cdf = d_cdf; s.random_time = 1; s.ref_init_prob = 1.0f; s.seed = 99;
    igx_motion_sampler hs = s;
    hs.cdf = cdf.data();
    std::vector<std::vector<float>> device_batches(fetches, std::vector<float>((size_t)n * K * NO));
    std::vector<int64_t> ids(n), want_ids(n);
    std::vector<double> times(n), want_times(n);
    std::vector<float> want((size_t)n * K * NO);
    double worst = 0.0;
    for (int f = 0; f < fetches; ++f) {
        s.rng_offset = hs.rng_offset = (uint64_t)(f + 1);
        IGX_OK_OR_FAIL(igx_motion_fetch_amp_obs(&dev_t, &s, n, K, dt_step, f & 1, d_batch, d_ids, d_times, stream));
        IGX_OK_OR_FAIL(igx_recorder_capture(rec, d_batch, stream));
        // (reading back here serialises the loop; a real consumer would only keep enqueueing)
        HIP_OK(hipMemcpyAsync(device_batches[f].data(), d_batch, device_batches[f].size() * sizeof(float), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(ids.data(), d_ids, n * sizeof(int64_t), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(times.data(), d_times, n * sizeof(double), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        orc_motion_sample(&host_t, &hs, n, want_ids.data(), want_times.data());
        if (std::memcmp(ids.data(), want_ids.data(), n * sizeof(int64_t)) || std::memcmp(times.data(), want_times.data(), n * sizeof(double))) {
            std::fprintf(stderr, "fetch %d: the draws differ from the checker's\n", f);
            return 1;
        }
        orc_motion_amp_obs_steps(&host_t, want_ids.data(), want_times.data(), n, K, dt_step, f & 1, want.data());
        for (size_t i = 0; i < want.size(); ++i) {
            const double ref = want[i], got = device_batches[f][i];
            const double err = std::fabs(got - ref) / std::fmax(1.0, std::fabs(ref));
            if (!(err <= worst)) worst = err;
            if (!(err <= 1e-5)) {
                std::fprintf(stderr, "fetch %d entry %zu (column %zu): got %.9g want %.9g\n", f, i, i % NO, got, ref);
                return 1;
            }
        }
    }
    IGX_OK_OR_FAIL(igx_recorder_sync(rec));
    if (igx_recorder_steps(rec) != fetches) {
        std::fprintf(stderr, "recorder holds %lld steps, %d captured\n", (long long)igx_recorder_steps(rec), fetches);
        return 1;
    }
    const float *host_rows = igx_recorder_host_data(rec);
    for (int f = 0; f < fetches; ++f)
        if (std::memcmp(host_rows + (size_t)f * n * K * NO, device_batches[f].data(), (size_t)n * K * NO * sizeof(float))) {
            std::fprintf(stderr, "batch %d in pinned host memory differs from the device batch\n", f);
            return 1;
        }
    // one more capture than the recorder was created for is refused, not written past the end
    if (igx_recorder_capture(rec, d_batch, stream) != IGX_E_SIZE) {
        std::fprintf(stderr, "a capture beyond max_steps was not refused\n");
        return 1;
    }
    IGX_OK_OR_FAIL(igx_recorder_destroy(rec));
    HIP_OK(hipStreamDestroy(stream));
    std::printf("cabi amp fetch ok: %d fetches of %lld x %d observations, worst error %.2e\n", fetches, (long long)n, K, worst);
    return 0;
}
