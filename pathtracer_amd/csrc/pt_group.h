// pt_group.h -- several GPUs of one node behind ONE handle, for a single-process host (included by ptgpu.hip).
//
// The reference's host is one process whose main loop hands every frame to a pool of workers and waits for them
// (reference src/window.rs:109-116 WorkerPool::new, :166-196 process(tile) x tiles + wait()).  A PtGroup keeps that
// shape with GPUs in place of the worker threads: one PtHandle and one host thread per device, each device rendering
// the global sample indices s with s % n_devices == k for ALL pixels (the counter RNG is keyed on the global index,
// so the union over the devices is exactly the sample set one GPU would render), scene and BVH replicated.  The data
// path has no collective.  The one exchange happens when the host asks for the image: the devices' u32 accumulators
// are summed onto the first device with RCCL (ncclReduce, ncclUint32, ncclSum over xGMI) into a buffer of its own, so
// every device keeps its partial sums and rendering can go on afterwards; integer sums make the result bit-identical
// to one GPU's whatever the reduction order.
//
// Calls SMALLER than the group - the window loop renders ONE sample per frame (reference src/window.rs:166) - have no
// sample indices to split.  They are split by image tile instead (ptgpu_render_tiles: the reference's 32x32 work tiles,
// interleaved over the devices): every device renders all of the call's samples for its own tiles, so a frame of one
// sample keeps every GPU busy, and a device can show its tiles without anybody else's sums.  ptgpu_group_frame builds
// the window loop on that: each device's last pass writes the display bytes of its tiles, ONE exchange of bytes
// (ncclReduce, ncclUint8/ncclSum: the tiles are disjoint, so the sum is the union; a quarter of the u32 sums' size) brings
// them onto the first device, a copy brings them to the host, and the call returns the frame before - the host draws
// one frame behind and no device ever waits for it.  Sample shards and tile shards may follow each other on the same
// accumulators: all that the sum over the devices needs is that every (pixel, sample index) was rendered once.
//
// Stream ordering.  Each member handle owns at most PT_MAX_PIPES (4) pass streams and one copy stream on ITS device, and
// the group adds none: every exchange is enqueued on the members' copy streams (always the same stream per
// communicator, so RCCL's operations on it are ordered like their calls) - the reduction of the sums after the members
// have been synchronised, a frame's exchange of display bytes behind the event of the frame's last pass.  Inside a handle every
// cross-stream wait names an event that was recorded EARLIER in host time (run_pass records `resolved` at the end of
// pass k-1 before pass k is enqueued; ptgpu_render's final join waits for events already recorded).  By induction
// over host enqueue order the oldest unfinished command of a device is either a kernel or a wait whose event is
// already complete, so no stream can wait for a signal queued behind it, however streams map to hardware queues.
#pragma once

#include <rccl/rccl.h>

#include <condition_variable>
#include <functional>
#include <thread>

struct PtGroup {
    struct Worker {
        std::thread thread;
        std::mutex m;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = false, quit = false;
        int rc = PT_OK;
    };
    std::vector<PtHandle *> members;
    std::vector<int> devices;
    std::vector<Worker *> workers;
    std::vector<ncclComm_t> comms;
    bool comms_ready = false;
    std::string error;
    // on devices[0]: the reduced accumulator and what is shown
    uint32_t *d_sum = nullptr;
    uint8_t *d_pixels8 = nullptr, *h_pixels8 = nullptr;
    uint32_t width = 0, height = 0;
    bool have_scene = false;
    uint64_t cursor = 0;          // global sample index of the next sample (per pixel)
    uint64_t total_samples = 0;   // what the last reduction summed
    bool dirty = true;            // something was rendered (or cleared) since the last reduction of the sums
    bool only_tiles = true;       // since the last clear every sample was rendered by tile shards: each device's own tiles hold
                                  // complete sums, which is what lets a device compute the display bytes of its tiles alone
    bool shared_device = false;   // test configuration: several members on ONE device (PTGPU_GROUP_SHARED_DEVICE=1), exchanges by
                                  // a kernel on that device instead of RCCL, which refuses two ranks on one GPU
    // pipelined presentation (ptgpu_group_frame): frame k lives in slot k % FRAME_SLOTS here and in every member
    struct Frame { uint8_t *d_img8 = nullptr, *h_img8 = nullptr; uint64_t num_samples = 0; };
    Frame frames[FRAME_SLOTS];
    uint64_t frames_begun = 0;
    uint64_t exchange_bytes = 0;
    double exchange_ms = 0.0;
    hipEvent_t t0 = nullptr, t1 = nullptr;   // around the reduction on devices[0]'s stream (created at the first one)
};

namespace {

int group_fail(PtGroup *g, int code, const std::string &msg) {
    if (g) g->error = msg;
    return code;
}

void group_worker_main(PtGroup::Worker *w) {
    for (;;) {
        std::function<int()> job;
        {
            std::unique_lock<std::mutex> lock(w->m);
            w->cv.wait(lock, [&] { return w->has_job || w->quit; });
            if (w->quit) return;
            job = std::move(w->job);
            w->has_job = false;
        }
        int rc;
        try { rc = job(); } catch (const std::bad_alloc &) { rc = PT_ERR_OUT_OF_MEMORY; } catch (...) { rc = PT_ERR_INTERNAL; }
        {
            std::lock_guard<std::mutex> lock(w->m);
            w->rc = rc;
            w->done = true;
        }
        w->cv.notify_all();
    }
}

// process(job) on every worker, then wait(): the reference's fan-out and barrier (src/window.rs:166-196), one job per
// device instead of one per tile.  Returns the first failure and copies that member's error text.
int group_run(PtGroup *g, const std::function<int(uint32_t)> &fn) {
    const uint32_t n = (uint32_t)g->workers.size();
    // every job is built before the first is handed out: an allocation that fails here leaves no worker holding a
    // reference to `fn` while this function unwinds
    std::vector<std::function<int()>> jobs;
    jobs.reserve(n);
    for (uint32_t k = 0; k < n; ++k) jobs.emplace_back([&fn, k] { return fn(k); });
    for (uint32_t k = 0; k < n; ++k) {
        PtGroup::Worker *w = g->workers[k];
        {
            std::lock_guard<std::mutex> lock(w->m);
            w->job = std::move(jobs[k]);   // (a move: nothing below can throw)
            w->has_job = true;
            w->done = false;
        }
        w->cv.notify_all();
    }
    int first = PT_OK;
    for (uint32_t k = 0; k < n; ++k) {
        PtGroup::Worker *w = g->workers[k];
        std::unique_lock<std::mutex> lock(w->m);
        w->cv.wait(lock, [&] { return w->done; });
        if (w->rc != PT_OK && first == PT_OK) {
            first = w->rc;
            g->error = "device " + std::to_string(g->devices[k]) + ": " + ptgpu_last_error(g->members[k]);
        }
    }
    return first;
}

void group_free_buffers(PtGroup *g) {
    if (g->devices.empty()) return;
    (void)hipSetDevice(g->devices[0]);
    if (g->d_sum) (void)hipFree(g->d_sum);
    if (g->d_pixels8) (void)hipFree(g->d_pixels8);
    if (g->h_pixels8) (void)hipHostFree(g->h_pixels8);
    g->d_sum = nullptr; g->d_pixels8 = nullptr; g->h_pixels8 = nullptr;
    for (PtGroup::Frame &f : g->frames) {
        if (f.d_img8) (void)hipFree(f.d_img8);
        if (f.h_img8) (void)hipHostFree(f.h_img8);
        f.d_img8 = nullptr; f.h_img8 = nullptr;
    }
    g->frames_begun = 0;
}

void group_free_events(PtGroup *g) {
    if (g->devices.empty()) return;
    (void)hipSetDevice(g->devices[0]);
    if (g->t0) (void)hipEventDestroy(g->t0);
    if (g->t1) (void)hipEventDestroy(g->t1);
    g->t0 = g->t1 = nullptr;
}

// samples with global index below c that belong to device k of n
inline uint64_t shard_count_below(uint64_t c, uint32_t k, uint32_t n) { return (c + n - 1 - k) / n; }

#define NCCLCHK(g, call)                                                                                     \
    do {                                                                                                     \
        ncclResult_t r_ = (call);                                                                            \
        if (r_ != ncclSuccess) return group_fail(g, PT_ERR_HIP, std::string(#call) + ": " + ncclGetErrorString(r_)); \
    } while (0)
#define GHIPCHK(g, call)                                                                                     \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess)                                                                                \
            return group_fail(g, e_ == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Test configuration only: the members share one device, where RCCL cannot connect them.  dst = sum of the members'
// buffers, by kernels on that device (the members are idle or ordered behind `wait_for`).
__global__ void k_group_add_u32(uint32_t *dst, const uint32_t *src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}
__global__ void k_group_add_u8(uint8_t *dst, const uint8_t *src, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)(dst[i] + src[i]);
}

int group_comms(PtGroup *g) {
    if (g->comms_ready || g->shared_device) return PT_OK;
    // one communicator per device, all owned by this process (xGMI links between the devices are RCCL's business)
    g->comms.assign(g->members.size(), nullptr);
    NCCLCHK(g, ncclCommInitAll(g->comms.data(), (int)g->members.size(), g->devices.data()));
    g->comms_ready = true;
    return PT_OK;
}

// One exchange onto devices[0]: dst = sum over the members of src(k), `count` elements of 4 bytes (u32 = true) or 1,
// enqueued on every member's copy stream (member k's behind wait_for(k) when that is not null).
template <class Src, class Wait>
int group_sum_onto_first(PtGroup *g, void *dst, size_t count, bool u32, Src src, Wait wait_for) {
    const uint32_t n = (uint32_t)g->members.size();
    int rc = group_comms(g);
    if (rc) return rc;
    for (uint32_t k = 0; k < n; ++k) {
        hipEvent_t e = wait_for(k);
        if (e) { GHIPCHK(g, hipSetDevice(g->devices[k])); GHIPCHK(g, hipStreamWaitEvent(g->members[k]->copy_stream, e, 0)); }
    }
    if (g->shared_device) {
        hipStream_t st = g->members[0]->copy_stream;
        GHIPCHK(g, hipSetDevice(g->devices[0]));
        for (uint32_t k = 1; k < n; ++k) {   // everything on the first member's stream, behind the others' events
            hipEvent_t e = wait_for(k);
            if (e) GHIPCHK(g, hipStreamWaitEvent(st, e, 0));
        }
        GHIPCHK(g, hipMemsetAsync(dst, 0, count * (u32 ? 4 : 1), st));
        for (uint32_t k = 0; k < n; ++k) {
            if (u32) k_group_add_u32<<<1024, 256, 0, st>>>((uint32_t *)dst, (const uint32_t *)src(k), count);
            else k_group_add_u8<<<1024, 256, 0, st>>>((uint8_t *)dst, (const uint8_t *)src(k), count);
        }
        GHIPCHK(g, hipGetLastError());
        return PT_OK;
    }
    NCCLCHK(g, ncclGroupStart());
    for (uint32_t k = 0; k < n; ++k) {
        ncclResult_t r = ncclReduce(src(k), dst, count, u32 ? ncclUint32 : ncclUint8, ncclSum, 0, g->comms[k], g->members[k]->copy_stream);
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return group_fail(g, PT_ERR_HIP, std::string("ncclReduce: ") + ncclGetErrorString(r));
        }
    }
    NCCLCHK(g, ncclGroupEnd());
    return PT_OK;
}

// samples per pixel the members' sums add up to
uint64_t group_samples(const PtGroup *g) {
    uint64_t total = 0, tiles = ~0ull;
    for (const PtHandle *h : g->members) { total += h->local_samples; tiles = std::min(tiles, h->tile_samples); }
    return total + (g->members.empty() ? 0 : tiles);
}

// Everything rendered so far, summed onto devices[0] (g->d_sum); total_samples = what it holds per pixel.  Skipped when
// nothing was rendered since the last one (a host that presents again without rendering).
int group_reduce(PtGroup *g) {
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = group_run(g, [g](uint32_t k) { return ptgpu_sync(g->members[k]); });
    if (rc) return rc;
    if (!g->dirty) return PT_OK;
    const uint32_t n = (uint32_t)g->members.size();
    const size_t words = (size_t)g->width * g->height * 3;
    if ((rc = group_comms(g))) return rc;
    hipStream_t st0 = g->members[0]->copy_stream;
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    if (!g->t0) GHIPCHK(g, hipEventCreate(&g->t0));
    if (!g->t1) GHIPCHK(g, hipEventCreate(&g->t1));
    GHIPCHK(g, hipEventRecord(g->t0, st0));
    rc = group_sum_onto_first(g, g->d_sum, words, true, [g](uint32_t k) { return (const void *)g->members[k]->d_pixels32; },
                              [](uint32_t) { return (hipEvent_t) nullptr; });
    if (rc) return rc;
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    GHIPCHK(g, hipEventRecord(g->t1, st0));
    for (uint32_t k = 0; k < n; ++k) GHIPCHK(g, hipStreamSynchronize(g->members[k]->copy_stream));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, g->t0, g->t1);
    g->exchange_ms = ms;
    g->exchange_bytes = words * 4;
    g->total_samples = group_samples(g);
    g->dirty = false;
    return PT_OK;
}

// After an overflow in a frame, or when a frame cannot be pipelined: every member synchronised (voided passes rendered
// again), the frame pipeline of the group and of every member back at its start.
int group_frames_reset(PtGroup *g, bool sync) {
    int rc = PT_OK;
    if (sync) rc = group_run(g, [g](uint32_t k) { return ptgpu_sync(g->members[k]); });
    for (PtHandle *h : g->members) frames_reset(h);
    g->frames_begun = 0;
    return rc;
}

}  // namespace

extern "C" {

int ptgpu_group_create(const PtConfig *cfg, const int32_t *devices, uint32_t n_devices, PtGroup **out) {
    PT_API_BEGIN
    if (!cfg || !out) return PT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int visible = 0;
    if (hipGetDeviceCount(&visible) != hipSuccess || visible <= 0) return PT_ERR_NO_DEVICE;
    if (n_devices == 0) {   // all of the node's GPUs
        if (devices) return PT_ERR_INVALID_ARGUMENT;
        n_devices = (uint32_t)visible;
    }
    if (n_devices > (uint32_t)PT_MAX_DEVICES) return PT_ERR_INVALID_ARGUMENT;
    PtGroup *g = new (std::nothrow) PtGroup();
    if (!g) return PT_ERR_OUT_OF_MEMORY;
    // Test configuration (no second GPU at hand): with PTGPU_GROUP_SHARED_DEVICE=1 in the environment the members may
    // share ONE device; the exchanges are then kernels on that device, everything else is the code a node of GPUs runs.
    const char *shared = std::getenv("PTGPU_GROUP_SHARED_DEVICE");
    const bool allow_shared = shared && shared[0] == '1' && devices != nullptr;
    for (uint32_t k = 0; k < n_devices; ++k) {
        const int dev = devices ? devices[k] : (int)k;
        bool dup = false;
        for (int d : g->devices) dup |= d == dev;
        if (dup && allow_shared && dev == g->devices[0]) { g->shared_device = true; dup = false; }
        if (g->shared_device && dev != g->devices[0]) dup = true;   // (all on one device, or each on its own)
        if (dev < 0 || dup) { delete g; return PT_ERR_INVALID_ARGUMENT; }   // one handle per device
        if (dev >= visible) { delete g; return PT_ERR_NO_DEVICE; }
        g->devices.push_back(dev);
    }
    for (uint32_t k = 0; k < n_devices; ++k) {
        PtConfig c = *cfg;
        c.device = g->devices[k];
        c.shard_index = k;
        c.shard_count = n_devices;
        PtHandle *h = nullptr;
        int rc = ptgpu_create(&c, &h);
        if (rc) { ptgpu_group_destroy(g); return rc; }
        g->members.push_back(h);
    }
    for (uint32_t k = 0; k < n_devices; ++k) {
        PtGroup::Worker *w = new PtGroup::Worker();
        g->workers.push_back(w);
        w->thread = std::thread(group_worker_main, w);
    }
    *out = g;
    return PT_OK;
    PT_API_END(nullptr)
}

void ptgpu_group_destroy(PtGroup *g) {
    if (!g) return;
    for (PtGroup::Worker *w : g->workers) {
        { std::lock_guard<std::mutex> lock(w->m); w->quit = true; }
        w->cv.notify_all();
        if (w->thread.joinable()) w->thread.join();
        delete w;
    }
    for (PtHandle *h : g->members) if (h) { (void)hipSetDevice(h->device); (void)ptgpu_sync(h); }
    if (g->comms_ready) for (ncclComm_t c : g->comms) if (c) (void)ncclCommDestroy(c);
    group_free_buffers(g);
    group_free_events(g);
    for (PtHandle *h : g->members) ptgpu_destroy(h);
    delete g;
}

int ptgpu_group_upload_scene(PtGroup *g, const PtSceneDesc *scene) {
    PT_API_BEGIN
    if (!g || !scene) return PT_ERR_INVALID_ARGUMENT;
    g->have_scene = false;
    int rc = group_run(g, [g, scene](uint32_t k) { return ptgpu_upload_scene(g->members[k], scene); });
    if (rc) return rc;
    const size_t words = (size_t)scene->width * scene->height * 3;
    if (scene->width != g->width || scene->height != g->height || !g->d_sum) {
        group_free_buffers(g);
        GHIPCHK(g, hipSetDevice(g->devices[0]));
        GHIPCHK(g, hipMalloc((void **)&g->d_sum, words * 4));
        GHIPCHK(g, hipMalloc((void **)&g->d_pixels8, words));
        GHIPCHK(g, hipHostMalloc((void **)&g->h_pixels8, words, hipHostMallocDefault));
        for (PtGroup::Frame &f : g->frames) {
            GHIPCHK(g, hipMalloc((void **)&f.d_img8, words));
            GHIPCHK(g, hipHostMalloc((void **)&f.h_img8, words, hipHostMallocDefault));
        }
        g->width = scene->width; g->height = scene->height;
    }
    g->cursor = 0;
    g->total_samples = 0;
    g->dirty = true;
    g->only_tiles = true;
    g->frames_begun = 0;
    g->have_scene = true;
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_clear(PtGroup *g) {
    PT_API_BEGIN
    if (!g) return PT_ERR_INVALID_ARGUMENT;
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = group_run(g, [g](uint32_t k) { return ptgpu_clear(g->members[k]); });
    if (rc) return rc;
    g->cursor = 0;
    g->total_samples = 0;
    g->dirty = true;
    g->only_tiles = true;
    g->frames_begun = 0;
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_render(PtGroup *g, uint32_t n_samples) {
    PT_API_BEGIN
    if (!g) return PT_ERR_INVALID_ARGUMENT;
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    if (n_samples == 0) return PT_OK;
    const uint32_t n = (uint32_t)g->members.size();
    const uint64_t c0 = g->cursor, c1 = g->cursor + n_samples;
    if (c1 > 0xFFFFFFFFull) return group_fail(g, PT_ERR_INVALID_ARGUMENT, "sample indices are 32-bit");
    int rc;
    if (n_samples < n || n == 1) {
        // fewer samples than devices (the window loop's one sample per frame): split by image tile, every device renders
        // all of them for its own tiles (a group of one: the two splits are the same thing)
        rc = group_run(g, [g, c0, n_samples](uint32_t k) { return ptgpu_render_tiles(g->members[k], c0, n_samples); });
    } else {
        // device k takes the global sample indices of [c0, c1) that are congruent to k, for every pixel
        rc = group_run(g, [g, n, c0, c1](uint32_t k) {
            const uint64_t mine = ptgpu_group_shard_samples(c0, c1 - c0, k, n);
            if (!mine) return (int)PT_OK;
            const uint64_t first = c0 + ((uint64_t)k + n - c0 % n) % n;   // the first index >= c0 congruent to k
            PtHandle *h = g->members[k];
            if (hipSetDevice(h->device) != hipSuccess) return (int)PT_ERR_HIP;
            return render_range(h, (uint32_t)mine, false, (uint32_t)first, nullptr, 0u, true, n);
        });
        g->only_tiles = false;
    }
    g->dirty = true;
    if (rc) {
        // some devices may have enqueued their share and others not: the accumulators no longer add up to a contiguous
        // range of sample indices, so the group refuses further work until the scene is uploaded again
        g->have_scene = false;
        g->error += " (ptgpu_group_render failed part-way: upload the scene again)";
        return rc;
    }
    g->cursor = c1;
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_frame(PtGroup *g, uint32_t n_samples, const uint8_t **pixels, uint64_t *num_samples) {
    PT_API_BEGIN
    if (!g || !pixels) return PT_ERR_INVALID_ARGUMENT;
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    if (g->cursor + n_samples > 0xFFFFFFFFull) return group_fail(g, PT_ERR_INVALID_ARGUMENT, "sample indices are 32-bit");
    const uint32_t n = (uint32_t)g->members.size();
    const size_t bytes = (size_t)g->width * g->height * 3;
    int rc;
    // The synchronous way: what is left when the frames cannot be pipelined - sums that are not all from tile shards (no
    // device can then show its tiles alone), or a frame that lost a pass to an overflow.
    auto present_now = [&]() -> int {
        int r = ptgpu_group_present_pixels8(g, pixels);
        if (r == PT_OK && num_samples) *num_samples = g->total_samples;
        return r;
    };
    if (!g->only_tiles) {
        if ((rc = group_frames_reset(g, false))) return rc;
        if (n_samples) { if ((rc = group_run(g, [g, n_samples](uint32_t k) { return ptgpu_render_tiles(g->members[k], g->cursor, n_samples); }))) return rc; }
        g->cursor += n_samples;
        g->dirty = true;
        return present_now();
    }
    if (n_samples > 0) {
        if ((rc = group_comms(g))) return rc;
        const uint64_t seq = g->frames_begun + 1;
        const uint32_t shown = (uint32_t)(g->cursor + n_samples);   // (only tile shards so far: the cursor is the sample count)
        const uint32_t first = (uint32_t)g->cursor;
        // every device: the frame's samples for its own tiles, the last pass writing the display bytes of those tiles
        rc = group_run(g, [g, n_samples, first, shown](uint32_t k) {
            PtHandle *h = g->members[k];
            if (hipSetDevice(h->device) != hipSuccess) return (int)PT_ERR_HIP;
            return frame_begin(h, n_samples, true, first, shown);
        });
        g->dirty = true;
        if (rc) { g->have_scene = false; g->error += " (ptgpu_group_frame failed part-way: upload the scene again)"; return rc; }
        g->cursor += n_samples;
        g->frames_begun = seq;
        PtGroup::Frame &gf = g->frames[seq % FRAME_SLOTS];
        gf.num_samples = shown;
        // the one exchange of the frame: the devices' tiles (disjoint, zero elsewhere) summed onto the first device, each
        // member's part behind its own last pass; then the image and every member's overflow flags on their way to the host
        rc = group_sum_onto_first(g, gf.d_img8, bytes, false, [g, seq](uint32_t k) { return (const void *)frame_slot(g->members[k], seq).d_pixels8; },
                                  [g](uint32_t k) { PtHandle *h = g->members[k]; return (h->passes > 0 && last_pipe(h).resolved_pending) ? last_pipe(h).resolved : (hipEvent_t) nullptr; });
        if (rc) return rc;
        for (uint32_t k = 0; k < n; ++k) {
            PtHandle *h = g->members[k];
            GHIPCHK(g, hipSetDevice(g->devices[k]));
            if (k == 0) GHIPCHK(g, hipMemcpyAsync(gf.h_img8, gf.d_img8, bytes, hipMemcpyDeviceToHost, h->copy_stream));
            if ((rc = frame_copy(h, frame_slot(h, seq), true))) return group_fail(g, rc, ptgpu_last_error(h));
        }
    }
    if (g->frames_begun == 0) return present_now();   // (a flush before any frame)
    // hand out the newest frame of which every member's part has arrived - at the oldest the frame two before the newest,
    // which is waited for -, provided none of its passes overflowed
    const uint64_t target = frame_choice(g->frames_begun, n_samples > 0, [g, n](uint64_t seq) {
        for (uint32_t k = 0; k < n; ++k)
            if (hipEventQuery(frame_slot(g->members[k], seq).copied) != hipSuccess) return false;
        return true;
    });
    (void)hipGetLastError();
    bool voided = false;
    for (uint32_t k = 0; k < n; ++k) {
        PtHandle *h = g->members[k];
        bool v = false;
        GHIPCHK(g, hipSetDevice(g->devices[k]));
        if ((rc = frame_wait(h, frame_slot(h, target), &v))) return group_fail(g, rc, ptgpu_last_error(h));
        voided |= v;
    }
    if (voided) {
        if ((rc = group_frames_reset(g, true))) return rc;
        return present_now();
    }
    *pixels = g->frames[target % FRAME_SLOTS].h_img8;
    if (num_samples) *num_samples = g->frames[target % FRAME_SLOTS].num_samples;
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_sync(PtGroup *g) {
    PT_API_BEGIN
    if (!g) return PT_ERR_INVALID_ARGUMENT;
    return group_run(g, [g](uint32_t k) { return ptgpu_sync(g->members[k]); });
    PT_API_END(nullptr)
}

int ptgpu_group_read_pixels32(PtGroup *g, uint32_t *out, uint64_t *num_samples) {
    PT_API_BEGIN
    if (!g || !out) return PT_ERR_INVALID_ARGUMENT;
    int rc = group_reduce(g);
    if (rc) return rc;
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    GHIPCHK(g, hipMemcpy(out, g->d_sum, (size_t)g->width * g->height * 12, hipMemcpyDeviceToHost));
    if (num_samples) *num_samples = g->total_samples;
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_read_pixels8(PtGroup *g, uint8_t *out) {
    PT_API_BEGIN
    if (!g || !out) return PT_ERR_INVALID_ARGUMENT;
    int rc = group_reduce(g);
    if (rc) return rc;
    PtHandle *h0 = g->members[0];
    const uint32_t words = g->width * g->height * 3;
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    k_pixels8<<<std::min<uint32_t>(4096, (words + BLOCK - 1) / BLOCK), BLOCK, 0, h0->stream>>>(g->d_sum, g->d_pixels8, words, h0->sc.iso_factor,
                                                                                              (uint32_t)g->total_samples);
    GHIPCHK(g, hipGetLastError());
    GHIPCHK(g, hipMemcpyAsync(out, g->d_pixels8, words, hipMemcpyDeviceToHost, h0->stream));
    GHIPCHK(g, hipStreamSynchronize(h0->stream));
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_present_pixels8(PtGroup *g, const uint8_t **pixels) {
    if (!g || !pixels) return PT_ERR_INVALID_ARGUMENT;
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_group_read_pixels8(g, g->h_pixels8);
    if (rc) return rc;
    *pixels = g->h_pixels8;
    return PT_OK;
}

int ptgpu_group_get_stats(PtGroup *g, PtStats *out) {
    PT_API_BEGIN
    if (!g || !out) return PT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof *out);
    for (PtHandle *h : g->members) {
        PtStats s;
        int rc = ptgpu_get_stats(h, &s);
        if (rc) return group_fail(g, rc, ptgpu_last_error(h));
        out->samples += s.samples; out->rays += s.rays; out->shadow_rays += s.shadow_rays; out->records += s.records;
        out->terminal_records += s.terminal_records; out->bvh_steps += s.bvh_steps;
        out->device_bytes += s.device_bytes; out->queue_bytes += s.queue_bytes;
        for (int k = 0; k < PT_KERNEL_COUNT; ++k) { out->kernel_launches[k] += s.kernel_launches[k]; out->kernel_ms[k] += s.kernel_ms[k]; }
        for (int l = 0; l < 25; ++l) out->level_records[l] += s.level_records[l];
        // (the same on every device)
        out->bvh_nodes = s.bvh_nodes; out->bvh_leaf_prims = s.bvh_leaf_prims; out->bvh_depth = s.bvh_depth;
        out->bvh_binary_depth = s.bvh_binary_depth; out->bvh_forced_median_splits = s.bvh_forced_median_splits;
        out->bvh_level_fold = s.bvh_level_fold;
        out->queue_growth = std::max(out->queue_growth, s.queue_growth);
        out->samples_per_pass = s.samples_per_pass; out->passes_in_flight = s.passes_in_flight;
    }
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_exchange_stats(PtGroup *g, uint64_t *bytes, double *ms) {
    if (!g) return PT_ERR_INVALID_ARGUMENT;
    if (bytes) *bytes = g->exchange_bytes;
    if (ms) *ms = g->exchange_ms;
    return PT_OK;
}

uint64_t ptgpu_group_shard_samples(uint64_t first, uint64_t count, uint32_t k, uint32_t n) {
    if (n == 0 || k >= n) return 0;
    return shard_count_below(first + count, k, n) - shard_count_below(first, k, n);
}

uint32_t ptgpu_group_size(const PtGroup *g) { return g ? (uint32_t)g->members.size() : 0u; }

PtHandle *ptgpu_group_member(PtGroup *g, uint32_t k) { return (g && k < g->members.size()) ? g->members[k] : nullptr; }

const char *ptgpu_group_last_error(PtGroup *g) { return g ? g->error.c_str() : "null group"; }

}  // extern "C"
