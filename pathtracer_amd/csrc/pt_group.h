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
// Stream ordering.  Each member handle owns at most PT_MAX_PIPES (4) streams on ITS device, as many as a process gets
// hardware queues per device, and the group adds none: the reduction is enqueued on each member's own stream after
// that member has been synchronised, so at that moment nothing else is queued on the device.  Inside a handle every
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
    for (uint32_t k = 0; k < n; ++k) {
        PtGroup::Worker *w = g->workers[k];
        {
            std::lock_guard<std::mutex> lock(w->m);
            w->job = [&fn, k] { return fn(k); };
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

// Everything rendered so far, summed onto devices[0] (g->d_sum); total_samples = what it holds per pixel.
int group_reduce(PtGroup *g) {
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = group_run(g, [g](uint32_t k) { return ptgpu_sync(g->members[k]); });
    if (rc) return rc;
    const uint32_t n = (uint32_t)g->members.size();
    const size_t words = (size_t)g->width * g->height * 3;
    if (!g->comms_ready) {
        // one communicator per device, all owned by this process (xGMI links between the devices are RCCL's business)
        g->comms.assign(n, nullptr);
        NCCLCHK(g, ncclCommInitAll(g->comms.data(), (int)n, g->devices.data()));
        g->comms_ready = true;
    }
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    if (!g->t0) GHIPCHK(g, hipEventCreate(&g->t0));
    if (!g->t1) GHIPCHK(g, hipEventCreate(&g->t1));
    GHIPCHK(g, hipEventRecord(g->t0, g->members[0]->stream));
    NCCLCHK(g, ncclGroupStart());
    for (uint32_t k = 0; k < n; ++k) {
        ncclResult_t r = ncclReduce(g->members[k]->d_pixels32, g->d_sum, words, ncclUint32, ncclSum, 0, g->comms[k], g->members[k]->stream);
        if (r != ncclSuccess) {
            (void)ncclGroupEnd();
            return group_fail(g, PT_ERR_HIP, std::string("ncclReduce: ") + ncclGetErrorString(r));
        }
    }
    NCCLCHK(g, ncclGroupEnd());
    GHIPCHK(g, hipSetDevice(g->devices[0]));
    GHIPCHK(g, hipEventRecord(g->t1, g->members[0]->stream));
    for (uint32_t k = 0; k < n; ++k) GHIPCHK(g, hipStreamSynchronize(g->members[k]->stream));
    float ms = 0.0f;
    (void)hipEventElapsedTime(&ms, g->t0, g->t1);
    g->exchange_ms = ms;
    g->exchange_bytes = words * 4;
    uint64_t total = 0;
    for (PtHandle *h : g->members) total += h->local_samples;
    g->total_samples = total;
    return PT_OK;
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
    for (uint32_t k = 0; k < n_devices; ++k) {
        const int dev = devices ? devices[k] : (int)k;
        bool dup = false;
        for (int d : g->devices) dup |= d == dev;
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
        g->width = scene->width; g->height = scene->height;
    }
    g->cursor = 0;
    g->total_samples = 0;
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
    return PT_OK;
    PT_API_END(nullptr)
}

int ptgpu_group_render(PtGroup *g, uint32_t n_samples) {
    PT_API_BEGIN
    if (!g) return PT_ERR_INVALID_ARGUMENT;
    if (!g->have_scene) return group_fail(g, PT_ERR_NO_SCENE, "no scene uploaded");
    const uint32_t n = (uint32_t)g->members.size();
    const uint64_t c0 = g->cursor, c1 = g->cursor + n_samples;
    // device k takes the global sample indices of [c0, c1) that are congruent to k: its local samples number
    // shard_count_below(c0) ... shard_count_below(c1) - 1, which is where its own cursor stands
    int rc = group_run(g, [g, n, c0, c1](uint32_t k) {
        const uint64_t mine = ptgpu_group_shard_samples(c0, c1 - c0, k, n);
        return mine ? ptgpu_render(g->members[k], (uint32_t)mine) : (int)PT_OK;
    });
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
