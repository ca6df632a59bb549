// Trajectory recorder: selected columns of a [n_rows, row_width] device buffer,
// every step, to pinned host memory without stalling the producer.
//
//   capture k:  producer stream   wait(slot_free[s])                 (first capture of a reused slot s)
//                                 gather kernel -> step k % B of staging slot s = (k / B) % S
//                                 record(slot_full[s])               (last capture of the slot's batch)
//               copy stream       wait(slot_full[s])
//                                 drain kernel on 16 workgroups: slot -> host + first * slab (pinned, mapped,
//                                 B slabs, contiguous); IGX_RECORDER_DRAIN_BLOCKS=0 selects hipMemcpyAsync
//                                 record(slot_free[s])
//
// A slot batches B consecutive steps (B slabs of [num_cols][n_rows] are contiguous on the host
// as well), B chosen so that a copy moves up to 128 MiB (at most 64 steps): small copies are bound by the
// per-copy latency of the copy engine, and the producer with them; large ones run at the link's rate.  The producer only
// ever waits for a copy that is S batches old; igx_recorder_sync flushes a partly filled batch.  Replaces the
// per-step obs.clone().cpu() + torch.cat of tasks/cartpole_sample.py:119-121,
// :289-293 (reference paths relative to IsaacGymEnvs/isaacgymenvs/).
#include <new>
#include <vector>

#include "igx_device.hpp"
#include "igx_host.hpp"
#include "igx_tile.hpp"

struct igx_recorder {
    int64_t n_rows = 0, max_steps = 0, steps = 0;
    int64_t flushed = 0;       // steps whose copy to the host has been enqueued
    int64_t batch = 1;         // steps per staging slot
    hipStream_t last_producer = nullptr;
    int32_t row_width = 0, num_cols = 0, ring_slots = 0;
    int device = -1;
    int32_t *d_cols = nullptr;
    float *d_ring = nullptr;   // [ring_slots][batch][num_cols][n_rows]  (whole rows: [..][n_rows][row_width])
    float *h_data = nullptr;   // [max_steps][num_cols][n_rows], pinned   (whole rows: [max_steps][n_rows][row_width])
    float *h_data_dev = nullptr;  // the same memory as the device addresses it
    int drain_blocks = 0;      // > 0: copy with recorder_drain_kernel on that many workgroups; 0: hipMemcpyAsync
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> slot_full, slot_free;
};

namespace igx {

// out[c][row] = src[row][cols[c]]: a wave reads 64 consecutive rows of one column
// (strided, but the whole tile of rows is touched by the num_cols passes while it is
// in cache) and writes 256 contiguous bytes
__global__ __launch_bounds__(256) void recorder_gather_kernel(const float *__restrict__ src, int row_width,
                                                              const int32_t *__restrict__ cols, int num_cols,
                                                              int64_t n_rows, float *__restrict__ out)
{
    const int64_t row = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (row >= n_rows) return;
    const float *r = src + row * row_width;
    for (int c = 0; c < num_cols; ++c) st_stream(out + (int64_t)c * n_rows + row, r[cols[c]]);
}

// whole-row capture (cols == NULL at creation): the [n_rows, row_width] buffer as it is, 16 bytes per lane
__global__ __launch_bounds__(256) void recorder_rows_kernel(const float *__restrict__ src, float *__restrict__ out,
                                                            int64_t floats, int vec)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (vec) {
        const int64_t n4 = floats / 4;
        if (i < n4) st_stream(reinterpret_cast<v4f *>(out) + i, ld_stream(reinterpret_cast<const v4f *>(src) + i));
        if (i < floats - 4 * n4) out[4 * n4 + i] = src[4 * n4 + i];
    } else {
        for (int64_t k = 4 * i; k < floats && k < 4 * i + 4; ++k) out[k] = src[k];
    }
}

// staging slot -> pinned host memory by a kernel of a few workgroups (the host buffer is mapped into the
// device's address space).  The runtime's own device-to-host copy is a blit kernel that fills the whole
// GPU for the duration of the transfer, and the producer's kernels queue behind it; a narrow kernel leaves
// the compute units to the producer while it keeps the link busy.
__global__ __launch_bounds__(256) void recorder_drain_kernel(const v4f *__restrict__ src, v4f *__restrict__ host_dst,
                                                             int64_t n4, const float *__restrict__ src_tail,
                                                             float *__restrict__ dst_tail, int tail)
{
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) host_dst[i] = ld_stream(src + i);
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

static int hip_fail(hipError_t e, const char *what)
{
    fail((int)e, "igx_recorder: %s: %s", what, hipGetErrorString(e));
    return (int)e;
}

}  // namespace igx

using namespace igx;

#define REC_HIP(call, what)                              \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) {                          \
            igx_recorder_destroy(r);                     \
            return hip_fail(e_, what);                   \
        }                                                \
    } while (0)

extern "C" int igx_recorder_create(igx_recorder **out, int64_t n_rows, int32_t row_width, const int32_t *cols,
                                   int32_t num_cols, int32_t ring_slots, int64_t max_steps)
{
    if (!out) return fail(IGX_E_NULL, "igx_recorder_create: NULL argument");
    *out = nullptr;
    const bool whole_rows = cols == nullptr;  // every column, rows kept as they are
    if (whole_rows && num_cols != 0) return fail(IGX_E_NULL, "igx_recorder_create: cols is NULL but num_cols is %d", num_cols);
    if (whole_rows) num_cols = row_width;
    if (n_rows <= 0 || row_width <= 0 || num_cols <= 0 || (!whole_rows && num_cols > 1024) || ring_slots < 1 ||
        ring_slots > 64 || max_steps <= 0)
        return fail(IGX_E_SIZE, "igx_recorder_create: bad sizes");
    for (int c = 0; !whole_rows && c < num_cols; ++c)
        if (cols[c] < 0 || cols[c] >= row_width) return fail(IGX_E_PARAM, "igx_recorder_create: column %d out of range", cols[c]);
    const int64_t slab = (int64_t)num_cols * n_rows;
    if (slab > ((int64_t)1 << 40) / max_steps) return fail(IGX_E_SIZE, "igx_recorder_create: host buffer too large");
    igx_recorder *r = new (std::nothrow) igx_recorder();
    if (!r) return fail(IGX_E_SIZE, "igx_recorder_create: out of host memory");
    r->n_rows = n_rows; r->row_width = row_width; r->num_cols = num_cols; r->ring_slots = ring_slots;
    r->max_steps = max_steps;
    const int64_t slab_bytes = (int64_t)sizeof(float) * slab;
    // steps per host copy: copies of a hundred MiB run at the link's rate (8 MiB ones at 3/4, 1 MiB ones at
    // 1/2 of it); at most 64 steps, so that small captures do not sit on the device for long
    static const int64_t target = (int64_t)tuning_int("IGX_RECORDER_COPY_MIB", 128) << 20;
    r->batch = target / slab_bytes;
    r->batch = r->batch < 1 ? 1 : r->batch > 64 ? 64 : r->batch;
    if (r->batch > max_steps) r->batch = max_steps;
    REC_HIP(hipGetDevice(&r->device), "hipGetDevice");
    if (!whole_rows) {
        REC_HIP(hipMalloc(&r->d_cols, sizeof(int32_t) * num_cols), "hipMalloc(cols)");
        REC_HIP(hipMemcpy(r->d_cols, cols, sizeof(int32_t) * num_cols, hipMemcpyHostToDevice), "hipMemcpy(cols)");
    }
    REC_HIP(hipMalloc(&r->d_ring, sizeof(float) * slab * r->batch * ring_slots), "hipMalloc(ring)");
    REC_HIP(hipHostMalloc(&r->h_data, sizeof(float) * slab * max_steps, hipHostMallocMapped), "hipHostMalloc");
    r->drain_blocks = tuning_int("IGX_RECORDER_DRAIN_BLOCKS", 16);
    if (r->drain_blocks < 0 || r->drain_blocks > 4096) r->drain_blocks = 0;
    if (r->drain_blocks > 0) REC_HIP(hipHostGetDevicePointer((void **)&r->h_data_dev, r->h_data, 0), "hipHostGetDevicePointer");
    REC_HIP(hipStreamCreateWithFlags(&r->copy_stream, hipStreamNonBlocking), "hipStreamCreate");
    for (int s = 0; s < ring_slots; ++s) {
        hipEvent_t a = nullptr, b = nullptr;
        REC_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming), "hipEventCreate");
        r->slot_full.push_back(a);
        REC_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming), "hipEventCreate");
        r->slot_free.push_back(b);
    }
    *out = r;
    return IGX_OK;
}

#undef REC_HIP
#define REC_HIP(call, what)                              \
    do {                                                 \
        hipError_t e_ = (call);                          \
        if (e_ != hipSuccess) return hip_fail(e_, what); \
    } while (0)

// enqueue the copy of steps [flushed, steps) - all in one staging slot - behind the producer's work so far
static int recorder_flush(igx_recorder *r, hipStream_t ps)
{
    if (r->flushed == r->steps) return IGX_OK;
    const int64_t slab = (int64_t)r->num_cols * r->n_rows;
    const int slot = (int)((r->flushed / r->batch) % r->ring_slots);
    const float *stage = r->d_ring + slab * (slot * r->batch + r->flushed % r->batch);
    const int64_t count = r->steps - r->flushed;
    REC_HIP(hipEventRecord(r->slot_full[slot], ps), "record(slot_full)");
    REC_HIP(hipStreamWaitEvent(r->copy_stream, r->slot_full[slot], 0), "wait(slot_full)");
    const int64_t floats = slab * count, n4 = floats / 4;
    float *dst_dev = r->h_data_dev ? r->h_data_dev + slab * r->flushed : nullptr;
    if (dst_dev && aligned16(stage) && aligned16(dst_dev)) {
        hipLaunchKernelGGL(recorder_drain_kernel, dim3((unsigned)r->drain_blocks), dim3(256), 0, r->copy_stream,
                           reinterpret_cast<const v4f *>(stage), reinterpret_cast<v4f *>(dst_dev), n4, stage + 4 * n4,
                           dst_dev + 4 * n4, (int)(floats - 4 * n4));
        int rc = launch_status("igx_recorder: drain");
        if (rc != IGX_OK) return rc;
    } else {  // odd row counts put slabs at addresses that are not multiples of 16 bytes: the runtime's copy
        REC_HIP(hipMemcpyAsync(r->h_data + slab * r->flushed, stage, sizeof(float) * floats, hipMemcpyDeviceToHost,
                               r->copy_stream),
                "hipMemcpyAsync");
    }
    REC_HIP(hipEventRecord(r->slot_free[slot], r->copy_stream), "record(slot_free)");
    r->flushed = r->steps;
    return IGX_OK;
}

extern "C" int igx_recorder_capture(igx_recorder *r, const float *src, igx_stream_t producer_stream)
{
    if (!r || !src) return fail(IGX_E_NULL, "igx_recorder_capture: NULL argument");
    if (r->steps >= r->max_steps) return fail(IGX_E_SIZE, "igx_recorder_capture: all %lld steps taken", (long long)r->max_steps);
    hipStream_t ps = static_cast<hipStream_t>(producer_stream);
    const int64_t batch_index = r->steps / r->batch, pos = r->steps % r->batch;
    const int slot = (int)(batch_index % r->ring_slots);
    const int64_t slab = (int64_t)r->num_cols * r->n_rows;
    float *stage = r->d_ring + slab * (slot * r->batch + pos);
    if (pos == 0 && batch_index >= r->ring_slots) REC_HIP(hipStreamWaitEvent(ps, r->slot_free[slot], 0), "wait(slot_free)");
    if (r->d_cols) {
        hipLaunchKernelGGL(recorder_gather_kernel, dim3((unsigned)((r->n_rows + 255) / 256)), dim3(256), 0, ps, src,
                           (int)r->row_width, r->d_cols, (int)r->num_cols, r->n_rows, stage);
    } else {
        const int vec = aligned16(src) && aligned16(stage);
        hipLaunchKernelGGL(recorder_rows_kernel, dim3((unsigned)((slab / 4 + 256) / 256)), dim3(256), 0, ps, src, stage, slab,
                           vec);
    }
    int rc = launch_status("igx_recorder_capture");
    if (rc != IGX_OK) return rc;
    r->steps += 1;
    r->last_producer = ps;
    if (pos == r->batch - 1) return recorder_flush(r, ps);
    return IGX_OK;
}

extern "C" int igx_recorder_sync(igx_recorder *r)
{
    if (!r) return fail(IGX_E_NULL, "igx_recorder_sync: NULL recorder");
    int rc = recorder_flush(r, r->last_producer);  // a partly filled batch
    if (rc != IGX_OK) return rc;
    REC_HIP(hipStreamSynchronize(r->copy_stream), "hipStreamSynchronize");
    return IGX_OK;
}

extern "C" int64_t igx_recorder_steps(const igx_recorder *r) { return r ? r->steps : -1; }

extern "C" const float *igx_recorder_host_data(const igx_recorder *r) { return r ? r->h_data : nullptr; }

extern "C" int igx_recorder_destroy(igx_recorder *r)
{
    if (!r) return IGX_OK;
    if (r->copy_stream) (void)hipStreamSynchronize(r->copy_stream);
    for (hipEvent_t e : r->slot_full) (void)hipEventDestroy(e);
    for (hipEvent_t e : r->slot_free) (void)hipEventDestroy(e);
    if (r->copy_stream) (void)hipStreamDestroy(r->copy_stream);
    if (r->d_cols) (void)hipFree(r->d_cols);
    if (r->d_ring) (void)hipFree(r->d_ring);
    if (r->h_data) (void)hipHostFree(r->h_data);
    delete r;
    return IGX_OK;
}
