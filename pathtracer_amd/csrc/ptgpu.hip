// ptgpu.hip -- host side of libptgpu: the C ABI of include/ptgpu.h, scene flattening, queue memory and
// the per-pass launch sequence.  Replaces the reference's WorkerPool fan-out and Backbuffer
// (reference src/window.rs:104-116,166-196; src/tracer.rs:200-283).  No CPU rendering path exists here.
#include "../../include/ptgpu.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "pt_kernels.h"
#include "pt_bvh_build.h"

using namespace pt;

namespace {

struct EventSpan { int kernel; hipEvent_t start, stop; };

}  // namespace

// Queue memory of ptgpu_render_noalloc: the levels, counters and overflow flags of one pipeline pass, in an array that
// is part of the loaded code object instead of coming from hipMalloc.  Every GPU the library runs on holds its own copy
// at its own address (looked up once per device in ptgpu_create, with that device current); in host code the array's
// name is only a handle for that lookup, never an address to pass to a kernel.
static __device__ __align__(256) unsigned char scratch[PTGPU_NOALLOC_SCRATCH_BYTES];
constexpr int PT_MAX_DEVICES = 64;
constexpr uint32_t NOALLOC_FLAGS = 64;   // overflow flags (= passes) per round of ptgpu_render_noalloc

// What one pipeline pass owns: its queues, counters, stream and (BVH scenes) traversal spill area.  Passes on
// different pipes overlap on the GPU; only the accumulation into the pixel sums is ordered between them.
constexpr int PT_MAX_PIPES = 4;
constexpr uint32_t FLAG_RING = PASS_FLAG_RING;
constexpr int FRAME_SLOTS = 3;   // frames of ptgpu_frame in flight at most: the newest two and the one handed out
struct Pipe {
    hipStream_t stream = nullptr;      // pipe 0 uses the handle's stream
    hipEvent_t resolved = nullptr;     // recorded after the pass has added its samples to the accumulator
    bool resolved_pending = false;
    std::vector<Level> levels;
    std::vector<std::pair<void *, size_t>> level_allocs;  // (block, bytes) of each queue level
    uint32_t *d_counts = nullptr;      // [(n_levels + 2) * NSHARD * CSTRIDE]: per level, then the shadow-ray row, then the traced row
    uint32_t *d_shard_caps = nullptr;  // [n_levels] shard capacity of each level (this pipe's part of the handle's array)
    uint32_t batch = 0;                // samples per pixel its levels are sized for
    uint2 *d_bvh_spill = nullptr;
    DevScene sc{};                     // the handle's scene with this pipe's spill area
};

struct PtHandle {
    PtConfig cfg{};
    int device = 0;
    hipStream_t stream = nullptr;   // what the caller sees: everything a call enqueues is ordered on it when the call returns
    std::string error;

    bool have_scene = false;
    std::vector<std::pair<void *, size_t>> level_pool;  // queue level blocks between free_levels and the next alloc_levels
    DevScene sc{};
    // device scene buffers
    float4 *d_sph = nullptr, *d_pln = nullptr, *d_mat = nullptr, *d_lights = nullptr;
    float *d_sky = nullptr, *d_gamma = nullptr;
    float4 *d_bvh_nodes = nullptr, *d_shade_rows = nullptr;
    uint32_t *d_bvh_always = nullptr, *d_prim_rows = nullptr;
    uint64_t bvh_nodes = 0, bvh_leaf_prims = 0;
    uint64_t bvh_depth = 0, bvh_binary_depth = 0, bvh_forced_median_splits = 0, bvh_level_fold = 0;
    bool has_glass = false;

    // accumulators
    uint32_t *d_pixels32 = nullptr;
    uint8_t *d_pixels8 = nullptr;
    uint8_t *h_pixels8 = nullptr;  // page-locked host copy handed out by ptgpu_present_pixels8 (sized for the current scene)
    uint32_t *d_ids = nullptr;
    uint64_t num_samples = 0;      // what pixels8 divides by (global count after a multi-GPU combine)
    uint64_t local_samples = 0;    // samples per pixel this handle accumulated since clear (ptgpu_render: every pixel of the image)
    uint64_t tile_samples = 0;     // samples per pixel OF ITS TILES this handle accumulated since clear (ptgpu_render_tiles)
    uint64_t sample_cursor = 0;    // local index of the next sample to render: never reused, also not when a pass was dropped

    // Tile shards (ptgpu_render_tiles): the image pixels of the 32x32 tiles this handle owns among shard_count handles
    // (ptgpu_tile_shard), tile after tile, row-major inside a tile.  Null when shard_count == 1 (every tile: the identity).
    uint32_t *d_tile_pixels = nullptr;
    uint32_t n_tile_pixels = 0;

    // Pipelined presentation (ptgpu_frame): the display bytes of up to FRAME_SLOTS frames on the device and in page-locked
    // host memory, each followed (at flags_offset) by a snapshot of the overflow flags.  The last pass of a frame writes
    // the device copy (k_resolve), the copy stream brings it to the host in one transfer, and a LATER call hands it out -
    // so the host never waits for the frame it has just asked for, and the GPU always has the next frames queued.
    struct Frame {
        uint8_t *d_pixels8 = nullptr, *h_pixels8 = nullptr;   // [flags_offset + FLAG_RING * 4]
        bool has_flags = false;          // a pass wrote the snapshot (false: the frame needed none - a tile shard without tiles)
        hipEvent_t copied = nullptr;     // recorded on copy_stream behind the transfer
        uint64_t num_samples = 0;        // samples per pixel the image shows
        uint64_t seq = 0;                // 1, 2, ... since the scene was uploaded or cleared
        uint64_t last_order = 0;         // `order` of the frame's last pass: the flag snapshot is final for every pass up to it
        bool in_flight = false;          // enqueued and not yet handed out
    };
    Frame frames[FRAME_SLOTS];
    size_t flags_offset = 0;             // of the current image: its display bytes rounded up to 256
    hipStream_t copy_stream = nullptr;
    uint64_t frames_begun = 0;

    // queues: one set per pass in flight
    int n_levels = 0;              // max_depth + 1
    uint32_t batch = 0;            // samples per pixel per pass
    int n_pipes = 0;
    uint64_t passes = 0;           // passes launched since the scene was uploaded
    int prev_pipe = 0;             // the pipe of the pass launched last (the next one takes the next pipe that is large enough)
    Pipe pipes[PT_MAX_PIPES];
    unsigned char *scratch_ptr = nullptr;  // this device's copy of `scratch`
    Pipe static_pipe;                      // ptgpu_render_noalloc's pass: queues carved out of scratch_ptr, the handle's stream
    bool no_spans = false;                 // inside ptgpu_render_noalloc: no per-kernel events (creating one allocates)
    uint32_t *d_shard_caps = nullptr;  // [n_pipes * n_levels]
    Counters *d_ctr = nullptr;
    // Levels of a scene with glass are sized for `growth`-fold branching, not for the worst case 2^depth.  Every pass
    // has an overflow flag; a pass that filled a level adds nothing to the accumulator and is repeated, after the
    // levels have been re-allocated with twice the growth, when the host next synchronises.
    uint32_t growth = 1;
    bool fixed_capacity = false;       // PtConfig.branch_capacity given: no re-allocation, overflow is an error
    uint32_t *d_flags = nullptr;       // [FLAG_RING] one flag per pass in flight
    uint64_t flag_cursor = 0;
    // A pass as launched: `spp` samples per pixel with global sample indices first, first + stride, ...; tiles: only the
    // pixels of this handle's tiles.  `frame`: the ptgpu_frame it belongs to (0: none).
    struct Pending { uint32_t slot, spp, first, stride; bool tiles; uint64_t order; };
    std::vector<Pending> pending;
    uint64_t launched = 0;             // passes launched so far: `order` of the next one
    bool fused = true;
    bool ahead = false;           // trace-ahead pipeline: a kernel shades its records and traces their children (fuse_trace_shade 3)
    bool lds_scene = false;       // primitive/material tables fit the shade kernel's LDS copy
    bool regroup = true;          // fused kernel regroups each block's records by material arm (fuse_trace_shade 2 / auto)
    size_t lds_bytes = 0;
    uint64_t device_bytes = 0;
    uint64_t level_bytes = 0;      // the part of device_bytes held by queue levels

    // profiling
    std::vector<EventSpan> spans;
    std::vector<hipEvent_t> free_events;
    uint64_t kernel_launches[PT_KERNEL_COUNT] = {};
    double kernel_ms[PT_KERNEL_COUNT] = {};
};

namespace {

int fail(PtHandle *h, int code, const std::string &msg) {
    if (h) h->error.assign(msg);   // (copied into the text's own storage, which keeps its reserved room: see api_caught)
    return code;
}

// No C++ exception may cross the C boundary (a std::bad_alloc from a scene-sized std::vector would reach a C, Rust or
// ctypes caller as std::terminate): every entry point that returns a status runs between PT_API_BEGIN and PT_API_END.
// Reports a fixed string; the handle's error text has room for it from creation on (ERROR_TEXT_RESERVE), so this
// neither allocates nor throws.
constexpr size_t ERROR_TEXT_RESERVE = 1024;
int api_caught(PtHandle *h, int code, const char *what) noexcept {
    try { if (h) h->error = what; } catch (...) { }
    return code;
}

// A failed HIP call.  Inside ptgpu_render_noalloc (no_spans) only the call's own text is reported - composing a message
// with the runtime's error string would allocate.
int fail_hip(PtHandle *h, hipError_t e, const char *call) noexcept {
    const int code = e == hipErrorOutOfMemory ? PT_ERR_OUT_OF_MEMORY : PT_ERR_HIP;
    if (!h || h->no_spans) return api_caught(h, code, call);
    try { h->error.assign(std::string(call) + ": " + hipGetErrorString(e)); } catch (...) { return api_caught(h, code, call); }
    return code;
}

#define HIPCHK(h, call)                                                                                      \
    do {                                                                                                     \
        hipError_t e_ = (call);                                                                              \
        if (e_ != hipSuccess) return fail_hip(h, e_, #call);                                                 \
    } while (0)
#define PT_API_BEGIN try {
#define PT_API_END(h)                                                                                        \
    } catch (const std::bad_alloc &) { return api_caught(h, PT_ERR_OUT_OF_MEMORY, "host memory allocation failed"); \
    } catch (const std::length_error &) { return api_caught(h, PT_ERR_OUT_OF_MEMORY, "host container size limit exceeded"); \
    } catch (const std::exception &e) { return api_caught(h, PT_ERR_INTERNAL, e.what());                     \
    } catch (...) { return api_caught(h, PT_ERR_INTERNAL, "unknown C++ exception"); }

template <class T>
int dev_alloc(PtHandle *h, T **p, size_t n) {
    *p = nullptr;
    if (n == 0) return PT_OK;
    HIPCHK(h, hipMalloc((void **)p, n * sizeof(T)));
    h->device_bytes += n * sizeof(T);
    return PT_OK;
}

template <class T>
void dev_free(T *&p) {
    if (p) (void)hipFree(p);
    p = nullptr;
}

void drain_level_pool(PtHandle *h) {
    for (auto &a : h->level_pool) (void)hipFree(a.first);
    h->level_pool.clear();
}

void free_levels(PtHandle *h) {
    for (Pipe &p : h->pipes) {
        // Blocks go to the handle's pool, not back to the driver: the next alloc_levels (a re-upload of a scene of the
        // same size, or levels that grew) takes blocks of the sizes it needs from there and frees the rest.  Freeing and
        // allocating tens of GiB takes the HIP runtime seconds at times (measured: 1-5 s for 65 GiB).
        for (auto &a : p.level_allocs) h->level_pool.push_back(a);
        p.level_allocs.clear();
        p.levels.clear();
        if (p.d_counts) { h->device_bytes -= (uint64_t)(h->n_levels + 2) * NSHARD * CSTRIDE * 4; }
        dev_free(p.d_counts);
        if (p.d_bvh_spill) { h->device_bytes -= (uint64_t)(BVH_STACK - BVH_LDS_ENTRIES) * BVH_TRAV_BLOCKS * BVH_STACK_STRIDE * sizeof(uint2); }
        dev_free(p.d_bvh_spill);
        p.resolved_pending = false;
    }
    h->device_bytes -= h->level_bytes;
    h->level_bytes = 0;
    if (h->d_shard_caps) h->device_bytes -= (uint64_t)h->n_levels * 4 * (uint64_t)std::max(1, h->n_pipes);
    dev_free(h->d_shard_caps);
}

void free_scene(PtHandle *h) {
    dev_free(h->d_sph); dev_free(h->d_pln); dev_free(h->d_mat); dev_free(h->d_lights); dev_free(h->d_sky); dev_free(h->d_gamma);
    dev_free(h->d_bvh_nodes); dev_free(h->d_bvh_always); dev_free(h->d_prim_rows); dev_free(h->d_shade_rows);
    dev_free(h->d_pixels32); dev_free(h->d_pixels8); dev_free(h->d_ids); dev_free(h->d_tile_pixels);
    h->n_tile_pixels = 0;
    if (h->h_pixels8) { (void)hipHostFree(h->h_pixels8); h->h_pixels8 = nullptr; }
    for (PtHandle::Frame &f : h->frames) {
        dev_free(f.d_pixels8);
        if (f.h_pixels8) { (void)hipHostFree(f.h_pixels8); f.h_pixels8 = nullptr; }
        f.in_flight = false;
    }
    free_levels(h);
    dev_free(h->d_ctr); dev_free(h->d_flags);
    h->pending.clear();
    h->have_scene = false;
    h->device_bytes = 0;
}

inline V3 toV3(PtVec3 v) { return v3(v.x, v.y, v.z); }

float4 mat0(const PtMaterial &m) { return make_float4(u2f((uint32_t)m.kind), m.color.x, m.color.y, m.color.z); }
float4 mat1(const PtMaterial &m) { return make_float4(m.roughness, m.metalness, m.ior, 0.0f); }
// Per-material constants of the Physically arm, computed once with the operations the reference performs per hit:
// f0 = mix_vec3(&Vec3::new(R, R, R), &reflectivity, metalness) (src/tracer.rs:543, src/common.rs:68-70) and
// PI * brdf_lambert(reflectivity) = PI * (reflectivity / PI) (src/tracer.rs:334-337,567).
float4 mat2(const PtMaterial &m) {
    const float R = 0.04f;
    V3 color = toV3(m.color);
    V3 f0 = (1.0f - m.metalness) * v3(R, R, R) + m.metalness * color;
    return make_float4(f0.x, f0.y, f0.z, 0.0f);
}
float4 mat3(const PtMaterial &m) {
    V3 pil = PI_F * (toV3(m.color) / PI_F);
    return make_float4(pil.x, pil.y, pil.z, 0.0f);
}

// Per-plane constants of intersect_plane (reference src/scene.rs:244-249), computed once with the same operations.
// An axis-aligned rectangle (see plane_test_axis) is stored with its edges in the order the short test expects.
void pack_plane(const PtPlane &p, float4 *out) {
    V3 u = toV3(p.u), v = toV3(p.v);
    V3 un = normalize(u), vn = normalize(v);
    float ul = length(u), vl = length(v);
    bool swap_edges;
    uint32_t code = plane_axis_code(toV3(p.normal), toV3(p.origin), un, ul, vn, vl, swap_edges);
    if (swap_edges) { std::swap(un, vn); std::swap(ul, vl); }
    float code_bits;
    memcpy(&code_bits, &code, 4);
    out[0] = make_float4(p.normal.x, p.normal.y, p.normal.z, p.origin.x);
    out[1] = make_float4(p.origin.y, p.origin.z, un.x, un.y);
    out[2] = make_float4(un.z, ul, vn.x, vn.y);
    out[3] = make_float4(vn.z, vl, code_bits, 0.0f);
}

// Scene -> the flat tables the kernels read (ids: spheres, emissive spheres, planes, emissive planes; reference
// src/scene.rs:266,280).
void flatten_scene(const PtSceneDesc *s, std::vector<float4> &sph, std::vector<float4> &pln, std::vector<float4> &mat,
                   std::vector<float4> &lights, bool &glass) {
    // ---- flatten (ids: spheres, emissive spheres, planes, emissive planes; reference src/scene.rs:266,280)
    const uint64_t n_sph = s->n_spheres + s->n_emissive_spheres, n_pl = s->n_planes + s->n_emissive_planes;
    sph.assign(n_sph, float4{}); pln.assign(4 * n_pl, float4{}); mat.assign(4 * (n_sph + n_pl), float4{});
    lights.assign(s->n_emissive_spheres + 4 * s->n_emissive_planes, float4{});
    glass = false;
    size_t id = 0;
    auto put_sphere = [&](const PtSphere &sp) {
        sph[id] = make_float4(sp.origin.x, sp.origin.y, sp.origin.z, sp.radius);
        mat[4 * id] = mat0(sp.material); mat[4 * id + 1] = mat1(sp.material); mat[4 * id + 2] = mat2(sp.material); mat[4 * id + 3] = mat3(sp.material);
        glass |= sp.material.kind == PT_MAT_TRANSLUCENT;
        ++id;
    };
    for (uint64_t i = 0; i < s->n_spheres; ++i) put_sphere(s->spheres[i]);
    for (uint64_t i = 0; i < s->n_emissive_spheres; ++i) {
        const PtMaterial &m = s->emissive_spheres[i].material;
        lights[i] = make_float4(m.color.x, m.color.y, m.color.z, 0.0f);
        put_sphere(s->emissive_spheres[i]);
    }
    auto put_plane = [&](const PtPlane &p) {
        pack_plane(p, &pln[4 * (id - n_sph)]);
        mat[4 * id] = mat0(p.material); mat[4 * id + 1] = mat1(p.material); mat[4 * id + 2] = mat2(p.material); mat[4 * id + 3] = mat3(p.material);
        glass |= p.material.kind == PT_MAT_TRANSLUCENT;
        ++id;
    };
    for (uint64_t i = 0; i < s->n_planes; ++i) put_plane(s->planes[i]);
    for (uint64_t i = 0; i < s->n_emissive_planes; ++i) {
        const PtPlane &p = s->emissive_planes[i];
        float4 *L = &lights[s->n_emissive_spheres + 4 * i];
        L[0] = make_float4(p.origin.x, p.origin.y, p.origin.z, p.normal.x);
        L[1] = make_float4(p.normal.y, p.normal.z, p.u.x, p.u.y);
        L[2] = make_float4(p.u.z, p.v.x, p.v.y, p.v.z);
        L[3] = make_float4(p.material.color.x, p.material.color.y, p.material.color.z, 0.0f);
        put_plane(p);
    }
}

// What the traversal needs to know of the builder's padding.
void apply_builder_limits(DevScene &sc, const BvhBuilder &b) { sc.bvh_reach = b.tame_reach(); }

// Shading rows are ordered as the trace kernel names hits: by unit of the tree (a leaf primitive's; the rows of node units
// stay empty), then the `always` list.  The inverse, for the scan over all primitives (bvh_finish): primitive -> row
// (MISS for a plane with a NaN normal, which is in neither and which no ray hits).
std::vector<uint32_t> shading_order(const BvhBuilder &b) {
    std::vector<uint32_t> order(b.unit_prim);
    order.insert(order.end(), b.always.begin(), b.always.end());
    return order;
}
std::vector<uint32_t> rows_of_primitives(const std::vector<uint32_t> &order, size_t n_prims) {
    std::vector<uint32_t> rows(std::max<size_t>(1, n_prims), MISS);
    for (size_t k = 0; k < order.size(); ++k)
        if (order[k] != BvhBuilder::MISS_ID) rows[order[k]] = (uint32_t)k;
    return rows;
}

hipEvent_t get_event(PtHandle *h) {
    if (!h->free_events.empty()) { hipEvent_t e = h->free_events.back(); h->free_events.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Span {
    PtHandle *h; int k; hipStream_t st; hipEvent_t a = nullptr, b = nullptr;
    Span(PtHandle *h_, int k_, hipStream_t st_) : h(h_), k(k_), st(st_) {
        if (h->cfg.profile_events && !h->no_spans) { a = get_event(h); b = get_event(h); (void)hipEventRecord(a, st); }
    }
    ~Span() {
        if (a) { (void)hipEventRecord(b, st); h->spans.push_back({k, a, b}); }
    }
};

int drain_spans(PtHandle *h) {
    for (auto &s : h->spans) {
        float ms = 0.0f;
        HIPCHK(h, hipEventSynchronize(s.stop));
        HIPCHK(h, hipEventElapsedTime(&ms, s.start, s.stop));
        h->kernel_ms[s.kernel] += ms;
        h->kernel_launches[s.kernel] += 1;
        h->free_events.push_back(s.start);
        h->free_events.push_back(s.stop);
    }
    h->spans.clear();
    return PT_OK;
}

// Start of a pass: every counter row zero (levels, the shadow-ray row, the traced row), the pass's overflow flag down,
// and level 0 holding records 0 .. n_records - 1 (its shards filled one after the other).  One launch: a pass of one
// sample per pixel is a handful of short kernels, and a separate fill for the zeroes would be one more of them.
__global__ void k_set_counts0(uint32_t *counts, uint32_t n_rows, uint32_t n_records, uint32_t shard_cap, uint32_t *overflow) {
    if (threadIdx.x == 0) *overflow = 0u;
    for (uint32_t i = threadIdx.x; i < n_rows * NSHARD * CSTRIDE; i += blockDim.x) {
        uint32_t c = 0;
        if (i < NSHARD * CSTRIDE && (i % CSTRIDE) == 0u) {   // row 0, the counter of shard q (the words beside it: see k_trace_dyn)
            const uint32_t q = i / CSTRIDE;
            const uint64_t start = (uint64_t)q * shard_cap;
            if (start < n_records) c = (uint32_t)std::min<uint64_t>(shard_cap, n_records - start);
        }
        counts[i] = c;
    }
}

uint32_t round_up(uint64_t x, uint32_t m) { return (uint32_t)(((x + m - 1) / m) * m); }

// Pipes (passes in flight) and their sizes, auto.
// Small scenes: two pipes of equal size - the forward kernels are bound by arithmetic and latency, the backward sweep by
// HBM bandwidth, and one pass's sweep runs under the other's forward kernels (+4-7 % on the sample scene, +5 % on the
// room; a third pass adds nothing).
// BVH scenes: ONE large pipe - a launch of the trace kernel ends with a few rays that walk hundreds of nodes on an
// otherwise idle GPU, and a longer launch dilutes that tail better than a second pass fills it (1M-rectangle field at
// 1080p: one pass of 64 spp 7 550 Mrays/s, one of 32 spp 7 140, two of 16 spp 6 890, three of 8 spp 6 410) - plus two SMALL
// ones that only passes of a few samples use: the window loop's frames of one sample, of which three in flight take
// 1.10 ms each at 1080p where one in flight takes 2.1 (a launch of 2M rays is nearly all tail).  (Three pipes and the
// copy stream are four streams, as many as a process gets hardware queues; a fourth pipe would share one, and costs.)
constexpr uint32_t SMALL_PIPE_SPP = 4;
int big_pipes_wanted(const PtHandle *h) {
    const uint32_t c = h->cfg.concurrent_passes;
    if (c == 0) return h->sc.bvh_nodes ? 1 : 2;
    return (int)std::min<uint32_t>(c, PT_MAX_PIPES);
}
int small_pipes_wanted(const PtHandle *h) { return (h->cfg.concurrent_passes == 0 && h->sc.bvh_nodes) ? 2 : 0; }

int alloc_levels(PtHandle *h) {
    const uint32_t P = h->sc.width * h->sc.height;
    const int D = h->cfg.max_depth;
    h->n_levels = D + 1;
    const uint32_t branch_cap = h->growth;
    // Samples per pixel per pass, auto: what a memory budget for the queues of all pipes allows, at most 64 (128 for BVH
    // scenes) and at most 64M (128M) records at level 0.  Measured at 1080p with the pipes as above (profiles/debug/
    // frame_loop.py; queue memory, Mrays/s): sample scene 2 x 32 spp 44.5 GiB 31 100, 2 x 16 22.3 GiB 30 810, 2 x 11 15.3 GiB
    // 30 440, 2 x 8 11.1 GiB 29 950, 2 x 4 5.6 GiB 28 850; room at depth 16 2 x 8 39 GiB 23 930, 2 x 3 14.6 GiB 23 630; the
    // 1M-rectangle field (BVH) 1 x 64 44.5 GiB 7 550, 1 x 32 22.3 GiB 7 140, 1 x 23 16 GiB 6 770.  Scenes without a BVH lose
    // 1-3 % from 44 to 15 GiB - every kernel boundary drains the chip once, and two passes in flight cover most of that -
    // so their budget is 16 GiB; the BVH trace kernel's tail costs 10 % there, so BVH scenes keep 48 GiB.  (Allocating the
    // queues also takes the runtime time: 0.2 s for 44 GiB at first upload.)
    uint32_t b = h->cfg.samples_per_batch;
    const uint64_t target = (h->sc.bvh_nodes ? 128u : 64u) << 20;
    if (b == 0) b = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(h->sc.bvh_nodes ? 128 : 64, target / std::max<uint32_t>(P, 1)));
    auto factor = [&](int l) -> uint64_t {
        if (!h->has_glass) return 1;  // only Translucent branches (reference src/tracer.rs:511-536)
        return std::min<uint64_t>(l < 31 ? (1ull << l) : (1ull << 31), branch_cap);
    };
    const uint64_t rec_bytes = Level::RECORD_BYTES;
    const int n_big = big_pipes_wanted(h);
    const uint64_t budget = (h->sc.bvh_nodes ? 48ull : 16ull) << 30;  // (the estimate below counts every level at full size)
    auto estimate = [&](uint32_t spp) { uint64_t t = 0; for (int l = 0; l <= D; ++l) t += (uint64_t)P * spp * factor(l) * rec_bytes * 5 / 4; return t; };
    if (h->cfg.samples_per_batch == 0) {
        while (b > 1 && estimate(b) * n_big > budget) {
            // (halving first: a power of two when the budget is far away; then down in single steps)
            b = estimate(b / 2) * n_big > budget ? b / 2 : b - 1;
        }
    }
    h->batch = b;
    const int n_small = b > SMALL_PIPE_SPP ? small_pipes_wanted(h) : 0;
    h->n_pipes = std::min(PT_MAX_PIPES, n_big + n_small);
    std::vector<uint32_t> caps((size_t)h->n_pipes * h->n_levels);
    int rc;
    if ((rc = dev_alloc(h, &h->d_shard_caps, caps.size()))) return rc;
    h->passes = 0;
    h->prev_pipe = h->n_pipes - 1;
    for (int q = 0; q < h->n_pipes; ++q) {
        Pipe &pipe = h->pipes[q];
        pipe.batch = q < n_big ? b : SMALL_PIPE_SPP;
        const uint64_t S = (uint64_t)P * pipe.batch;
        pipe.levels.resize(h->n_levels);
        if ((rc = dev_alloc(h, &pipe.d_counts, (size_t)(h->n_levels + 2) * NSHARD * CSTRIDE))) return rc;
        HIPCHK(h, hipMemsetAsync(pipe.d_counts, 0, (size_t)(h->n_levels + 2) * NSHARD * CSTRIDE * 4, h->stream));
        pipe.sc = h->sc;
        pipe.d_shard_caps = h->d_shard_caps + (size_t)q * h->n_levels;
        if (h->sc.bvh_nodes) {
            if ((rc = dev_alloc(h, &pipe.d_bvh_spill, (size_t)(BVH_STACK - BVH_LDS_ENTRIES) * BVH_TRAV_BLOCKS * BVH_STACK_STRIDE))) return rc;
            pipe.sc.bvh_spill = pipe.d_bvh_spill;
        }
        for (int l = 0; l <= D; ++l) {
            uint64_t recs = S * factor(l);
            uint64_t per = (recs + NSHARD - 1) / NSHARD;
            uint32_t cap = (l == 0) ? round_up(per, 64) : round_up(per + per / 4 + 128, 64);
            if ((uint64_t)cap * NSHARD >= (1ull << 31))
                return fail(h, PT_ERR_INVALID_ARGUMENT, "queue level too large for 31-bit record indices; lower samples_per_batch");
            caps[(size_t)q * h->n_levels + l] = cap;
            size_t n = (size_t)cap * NSHARD;
            Level &L = pipe.levels[l];
            L.shard_cap = cap;
            L.count = pipe.d_counts + (size_t)l * NSHARD * CSTRIDE;
            // one allocation per level, carved into the SoA arrays (16-byte arrays first)
            size_t bytes = n * rec_bytes;
            char *base = nullptr;
            for (size_t k = 0; k < h->level_pool.size(); ++k)
                if (h->level_pool[k].second == bytes) {
                    base = (char *)h->level_pool[k].first;
                    h->level_pool.erase(h->level_pool.begin() + (long)k);
                    break;
                }
            if (!base && hipMalloc((void **)&base, bytes) != hipSuccess) {
                (void)hipGetLastError();
                drain_level_pool(h);   // the pooled blocks of other sizes may be what is in the way
                HIPCHK(h, hipMalloc((void **)&base, bytes));
            }
            h->device_bytes += bytes;
            h->level_bytes += bytes;
            pipe.level_allocs.push_back({base, bytes});
            L.base = base;
        }
    }
    HIPCHK(h, hipMemcpyAsync(h->d_shard_caps, caps.data(), caps.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drain_level_pool(h);   // blocks of sizes no level has any more
    return PT_OK;
}

// Blocks of the tile kernels (grid-stride over wave-tiles).  Measured on the sample scene: at 1080p with 8 spp per pass
// 2048 blocks (all resident, ~6 tiles per wave) is 7 % slower than 8192 (1-2 tiles per wave, the dispatcher balances
// uneven tiles); with 32 spp per pass and one pass in flight 16384 is another 1 % ahead (with two passes of 11 spp in
// flight, the default, it is 2.5 % behind 8192, and anything from 1536 to 8192 is the same), while a 1-spp pass of a
// 1024x1024 image loses 6 % with that many mostly empty blocks.  A pass of ONE sample per pixel (the window loop's frame: 2M records at 1080p, 32k
// wave-tiles) runs best with 1280-4096 blocks - every wave then takes several tiles and pays its start-up (scene copy to
// LDS, shard scan, cold instruction cache) once: frame loop at 1080p 0.279 ms per frame with 8192 blocks, 0.262 with 2048,
// 0.295 with 512.  PTGPU_GRID overrides it (tuning only).
static int grid_tiles(uint64_t level0_records) {
    static int g = [] { const char *e = getenv("PTGPU_GRID"); int v = e ? atoi(e) : 0; return v > 0 ? v : 0; }();
    return g ? g : (level0_records >= (48u << 20) ? 16384 : level0_records > (4u << 20) ? 8192 : 2048);
}
#define GRID_TILES grid_tiles(S)

// What one pipeline pass renders: `spp` samples per pixel with the global sample indices first, first + stride, ... -
// for every pixel of the image, or (tiles) for the pixels of this handle's tiles.  pixels8 != null: the pass also leaves
// the display bytes of its pixels there, computed from the sums it has just written and `shown` samples per pixel.
struct PassSpec {
    uint32_t spp, first, stride;
    bool tiles;
    uint8_t *pixels8;          // (a frame's buffer: the snapshot of the overflow flags goes behind the bytes, see PtHandle::Frame)
    uint32_t shown;
};
// The pass ptgpu_render cuts out of the handle's own sample shard: local sample indices local_first ...
PassSpec own_shard_pass(const PtHandle *h, uint32_t spp, uint32_t local_first) {
    return PassSpec{spp, h->cfg.shard_index + local_first * h->cfg.shard_count, h->cfg.shard_count, false, nullptr, 0u};
}

// Enqueues one pipeline pass on `pipe`.  `flag` is the pass's overflow word (device memory): cleared at the start of the
// pass, set by an append that finds a level full.
int run_pass(PtHandle *h, Pipe &pipe, Pipe *other, const PassSpec &ps, uint32_t *flag) {
    const uint32_t spp = ps.spp;
    const bool mapped = ps.tiles && h->d_tile_pixels != nullptr;
    const uint32_t P = mapped ? h->n_tile_pixels : h->sc.width * h->sc.height;
    const DevScene &sc = pipe.sc;
    std::vector<Level> &levels = pipe.levels;
    const uint32_t S = P * spp;
    const int D = h->cfg.max_depth;
    PassParams pp;
    pp.n_pixels = P;
    pp.sample_stride = ps.stride;
    pp.sample_first = ps.first;
    pp.pixel_map = mapped ? h->d_tile_pixels : nullptr;
    hipStream_t st = pipe.stream;
    const uint32_t flat_grid = (uint32_t)std::min<uint64_t>(8192, (S + BLOCK - 1) / BLOCK);
    const size_t lds = h->lds_bytes;
    // kernels that walk the BVH index a per-thread spill area sized for BVH_TRAV_BLOCKS blocks
    const bool bvh = sc.bvh_nodes != nullptr;
    const uint32_t RESCAN_GRID = 512;   // (of the kernels that repeat the few records a traversing kernel left to the linear scan)
    const uint32_t TG = sc.bvh_nodes ? std::min<uint32_t>(BVH_TRAV_BLOCKS, GRID_TILES) : GRID_TILES;

    for (Level &L : levels) L.overflow = flag;
    k_set_counts0<<<1, 1024, 0, st>>>(pipe.d_counts, (uint32_t)(D + 3), S, levels[0].shard_cap, flag);   // levels 0..D, shadow row, traced row
    if (!h->fused && !h->ahead && !bvh) {
        Span s(h, PT_KERNEL_RAYGEN, st);
        k_raygen<<<flat_grid, BLOCK, 0, st>>>(sc, levels[0], pp, S);
    }
    for (int l = 0; l < D; ++l) {
        Level &L = levels[l];
        Level &Lp = levels[l > 0 ? l - 1 : 0];
        Level &Ln = levels[l + 1];
        const uint32_t last = (l == D - 1) ? 1u : 0u;
        if (h->ahead) {
            Span s(h, PT_KERNEL_TRACE_SHADE, st);
            uint32_t *traced_row = pipe.d_counts + (size_t)(h->n_levels + 1) * NSHARD * CSTRIDE;
            if (h->lds_scene) {
                if (l == 0) k_ahead<true, true><<<TG, BLOCK, lds, st>>>(sc, L, Ln, pp, 0u, last, traced_row, h->d_ctr);
                else k_ahead<false, true><<<TG, BLOCK, lds, st>>>(sc, L, Ln, pp, (uint32_t)l, last, traced_row, h->d_ctr);
            } else {
                if (l == 0) k_ahead<true, false><<<TG, BLOCK, lds, st>>>(sc, L, Ln, pp, 0u, last, traced_row, h->d_ctr);
                else k_ahead<false, false><<<TG, BLOCK, lds, st>>>(sc, L, Ln, pp, (uint32_t)l, last, traced_row, h->d_ctr);
            }
        } else if (h->fused) {
            Span s(h, PT_KERNEL_TRACE_SHADE, st);
            if (h->regroup) {
                if (bvh) {
                    if (l == 0) k_shade_sorted<true, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade_sorted<false, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                } else {
                    if (l == 0) k_shade_sorted<true, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade_sorted<false, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                }
            } else {
                if (bvh) {  // (lds_scene is off with a BVH)
                    if (l == 0) k_shade<true, true, false, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade<true, false, false, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                } else if (h->lds_scene && sc.n_sph > DEFER_SPHERES) {   // many spheres: their full tests lane by lane (nearest_brute)
                    if (l == 0) k_shade<true, true, true, false, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade<true, false, true, false, true><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                } else if (h->lds_scene) {
                    if (l == 0) k_shade<true, true, true, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade<true, false, true, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                } else {
                    if (l == 0) k_shade<true, true, false, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);
                    else k_shade<true, false, false, false><<<TG, BLOCK, lds, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
                }
            }
        } else {
            {
                Span s(h, PT_KERNEL_TRACE, st);
                // (k_rescan: returns at once unless the trace kernel met rays only the linear scan can answer)
                if (bvh && l == 0) { k_trace_dyn<true><<<TG, BLOCK, lds, st>>>(sc, L, pp, h->d_ctr); k_rescan<true><<<RESCAN_GRID, BLOCK, 0, st>>>(sc, L, pp); }
                else if (bvh) { k_trace_dyn<false><<<TG, BLOCK, lds, st>>>(sc, L, pp, h->d_ctr); k_rescan<false><<<RESCAN_GRID, BLOCK, 0, st>>>(sc, L, pp); }
                else k_trace<<<TG, BLOCK, lds, st>>>(sc, L);
            }
            Span s(h, PT_KERNEL_SHADE, st);
            if (bvh && l == 0) k_shade<false, true, false, true><<<GRID_TILES, BLOCK, 0, st>>>(sc, L, Lp, Ln, pp, 0u, last, h->d_ctr);   // hits named by shading row
            else if (bvh) k_shade<false, false, false, true><<<GRID_TILES, BLOCK, 0, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
            else if (h->lds_scene) k_shade<false, false, true, false><<<GRID_TILES, BLOCK, 0, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
            else k_shade<false, false, false, false><<<GRID_TILES, BLOCK, 0, st>>>(sc, L, Lp, Ln, pp, (uint32_t)l, last, h->d_ctr);
        }
    }
    {
        Span s(h, PT_KERNEL_TERMINAL, st);
        uint32_t *shadow_row = pipe.d_counts + (size_t)h->n_levels * NSHARD * CSTRIDE;
        if (bvh) {
            k_terminal<true><<<TG, BLOCK, lds, st>>>(sc, levels[D], levels[D - 1], shadow_row);
            k_terminal<false, true><<<RESCAN_GRID, BLOCK, 0, st>>>(sc, levels[D], levels[D - 1], shadow_row);
        } else k_terminal<false><<<TG, BLOCK, lds, st>>>(sc, levels[D], levels[D - 1], shadow_row);
    }
    // Backward sweep over the even levels, deepest first; each pulls from the odd level above it (when that is a
    // traced level) and pushes into the odd level below.  With max_depth 1 level 1 holds terminal samples, which push.
    for (int l = (D - 1) & ~1; l >= 2; l -= 2) {
        Span s(h, PT_KERNEL_GATHER, st);
        k_gather<<<GRID_TILES, BLOCK, 0, st>>>(levels[l], levels[l + 1], levels[l - 1], l + 1 <= D - 1 ? 1u : 0u);
    }
    // The accumulator is read-modify-written without atomics: passes add their samples one after the other.
    if (other && other->resolved_pending) HIPCHK(h, hipStreamWaitEvent(st, other->resolved, 0));
    {
        Span s(h, PT_KERNEL_RESOLVE, st);
        k_resolve<<<(uint32_t)std::min<uint64_t>(8192, (P + BLOCK - 1) / BLOCK), BLOCK, 0, st>>>(sc, levels[0], levels[1], pp, spp, D >= 2 ? 1u : 0u, h->d_pixels32,
                                                                                                 ps.pixels8, ps.shown, h->d_flags,
                                                                                                 ps.pixels8 ? reinterpret_cast<uint32_t *>(ps.pixels8 + h->flags_offset) : nullptr);
    }
    // (the event stands for "this pass has added its samples": what the next pass's resolve step and a frame's copy wait
    // for; the bookkeeping kernel behind it only feeds the statistics, which are read after a full synchronisation)
    HIPCHK(h, hipEventRecord(pipe.resolved, st));
    pipe.resolved_pending = true;
    k_account<<<1, 64, 0, st>>>(pipe.d_counts, pipe.d_shard_caps, h->n_levels, h->ahead ? 1 : 0, flag, h->d_ctr);
    HIPCHK(h, hipGetLastError());
    return PT_OK;
}

// The next pass of the handle's own queues: round-robin over the pipes, its overflow flag a slot of the ring that is
// remembered (with the samples it stands for) until the host has looked at it.
int launch_pass(PtHandle *h, const PassSpec &ps) {
    // round-robin over the pipes whose levels hold a pass of this size (a BVH scene's small pipes take frames only)
    int q = h->prev_pipe;
    for (int t = 1; t <= h->n_pipes; ++t) {
        const int c = (h->prev_pipe + t) % h->n_pipes;
        if (h->pipes[c].batch >= ps.spp) { q = c; break; }
    }
    Pipe &pipe = h->pipes[q];
    Pipe *other = (h->passes > 0 && q != h->prev_pipe) ? &h->pipes[h->prev_pipe] : nullptr;  // the pass before this one
    const uint32_t slot = (uint32_t)(h->flag_cursor++ % FLAG_RING);
    h->pending.push_back({slot, ps.spp, ps.first, ps.stride, ps.tiles, h->launched++});
    int rc = run_pass(h, pipe, other, ps, h->d_flags + slot);
    if (rc) return rc;
    ++h->passes;
    h->prev_pipe = q;
    return PT_OK;
}
Pipe &last_pipe(PtHandle *h) { return h->pipes[h->prev_pipe]; }   // (of the pass launched last)

int sync_pipes(PtHandle *h) {
    for (Pipe &p : h->pipes) {
        if (p.stream) HIPCHK(h, hipStreamSynchronize(p.stream));
        p.resolved_pending = false;
    }
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));   // (frames on their way to the host read what later calls may change)
    return PT_OK;
}

// Looks at the overflow flags of the passes launched since the last look (all streams idle).  Voided passes are
// repeated: with the levels re-allocated for twice the branching each time round, which ends at the latest when a
// level can hold 2^level records per sample.  With a fixed PtConfig.branch_capacity an overflow is an error instead.
struct Range { uint32_t first, spp, stride; bool tiles; };
void drop_voided(PtHandle *h, const std::vector<Range> &todo) {
    for (const Range &r : todo) {
        uint64_t &mine = r.tiles ? h->tile_samples : h->local_samples;
        mine -= std::min<uint64_t>(mine, r.spp);
        h->num_samples -= std::min<uint64_t>(h->num_samples, r.spp);
    }
}

int resolve_pending(PtHandle *h) {
    std::vector<Range> todo;  // sample ranges (per pixel) still to be rendered again, served from the back
    for (;;) {
        // ---- look at the flags of everything launched since the last look
        if (!h->pending.empty()) {
            std::vector<uint32_t> flags(FLAG_RING);
            HIPCHK(h, hipMemcpy(flags.data(), h->d_flags, FLAG_RING * 4, hipMemcpyDeviceToHost));
            bool any = false;
            for (const auto &p : h->pending)
                if (flags[p.slot]) { todo.push_back({p.first, p.spp, p.stride, p.tiles}); any = true; }
            h->pending.clear();
            if (any) {
                if (h->fixed_capacity || h->growth >= (1u << 30)) {
                    // the voided passes added nothing and are not repeated: take them out of the counts pixels8 and a
                    // multi-GPU combine divide by (their sample indices stay used up)
                    drop_voided(h, todo);
                    return fail(h, PT_ERR_QUEUE_OVERFLOW,
                                h->fixed_capacity ? "a ray queue level overflowed (glass branching beyond branch_capacity); the samples of that pass are missing"
                                                  : "a ray queue level overflowed at full capacity; the samples of that pass are missing");
                }
                h->growth *= 2;
                free_levels(h);
                int rc = alloc_levels(h);
                if (rc) {
                    // No usable queues any more (out of memory at the doubled size, or a level beyond 31-bit indices): the
                    // handle goes back to "no scene" so that later calls fail with PT_ERR_NO_SCENE instead of launching
                    // kernels on freed queues.  The error text says what happened.
                    std::string why = h->error;
                    free_scene(h);
                    return fail(h, rc, "queues could not be re-allocated after an overflow (" + why + "); the scene has to be uploaded again");
                }
            }
        }
        if (todo.empty()) return PT_OK;
        // ---- render again what was voided, as many passes as there are flag slots
        while (!todo.empty() && h->pending.size() + PT_MAX_PIPES < FLAG_RING) {
            Range &r = todo.back();
            const uint32_t spp = std::min(h->batch, r.spp);  // (the pass size may have shrunk with the new level sizes)
            int rc = launch_pass(h, PassSpec{spp, r.first, r.stride, r.tiles, nullptr, 0u});
            if (rc) { drop_voided(h, todo); return rc; }
            r.first += spp * r.stride;
            r.spp -= spp;
            if (r.spp == 0) todo.pop_back();
        }
        int rc = sync_pipes(h);
        if (rc) return rc;
    }
}

// Enqueues n samples per pixel as passes of at most h->batch.  tiles == false: the next n samples of the handle's own
// sample shard, for every pixel (ptgpu_render).  tiles == true: the global sample indices first .. first + n - 1 for the
// pixels of the handle's tiles.  pixels8 != null: the last pass leaves the display bytes of its pixels there, for `shown`
// samples per pixel (0: the handle's own count after this call).  join: order the handle's stream behind every pipe
// at the end (what ptgpu_stream promises for ptgpu_render; a frame loop must not, or pipe 0's next pass would wait for
// the other pipes' passes to finish instead of running beside them).
// strided (with tiles == false): not the handle's own shard but the global sample indices first, first + stride, ... - how
// a group hands a member its share of a range that does not start where the member's own cursor stands.
int render_range(PtHandle *h, uint32_t n, bool tiles, uint32_t first, uint8_t *pixels8, uint32_t shown, bool join = true, uint32_t strided = 0u) {
    if (tiles && h->d_tile_pixels && h->n_tile_pixels == 0) {   // more shards than tiles: nothing to render, the counts advance
        h->tile_samples += n;
        h->num_samples += n;
        return PT_OK;
    }
    uint32_t done = 0;
    while (done < n) {
        const uint32_t spp = std::min(h->batch, n - done);
        PassSpec ps = tiles ? PassSpec{spp, first + done, 1u, true, nullptr, 0u}
                    : strided ? PassSpec{spp, first + done * strided, strided, false, nullptr, 0u}
                              : own_shard_pass(h, spp, (uint32_t)h->sample_cursor);
        if (pixels8 && done + spp == n) {
            ps.pixels8 = pixels8;
            ps.shown = shown ? shown : (uint32_t)(h->num_samples + spp);
        }
        int rc = launch_pass(h, ps);
        if (rc) return rc;
        done += spp;
        if (tiles) h->tile_samples += spp;
        else { h->local_samples += spp; if (!strided) h->sample_cursor += spp; }
        h->num_samples += spp;
        if (h->pending.size() + PT_MAX_PIPES >= FLAG_RING) {  // every flag slot is in use: look at them now
            if ((rc = sync_pipes(h))) return rc;
            if ((rc = resolve_pending(h))) return rc;
        }
        if (h->cfg.profile_events && h->spans.size() > 4096) {
            if ((rc = sync_pipes(h))) return rc;
            if ((rc = drain_spans(h))) return rc;
        }
    }
    // Calls that touch the accumulator (clear, import, upload) finish on the host before they return, so the other pipes
    // need no edge FROM the handle's stream; successive calls overlap like the passes of one call.
    if (join)
        for (int q = 1; q < h->n_pipes; ++q)
            if (h->pipes[q].resolved_pending) HIPCHK(h, hipStreamWaitEvent(h->stream, h->pipes[q].resolved, 0));
    return PT_OK;
}

// ---- pipelined presentation ------------------------------------------------------------------------------------------
// Frame k = ++frames_begun lives in slot k % FRAME_SLOTS.  Begin: its samples are enqueued, the last pass writing the
// display bytes and the flag snapshot into the frame's device buffer.  (The slot was last used by frame k - FRAME_SLOTS,
// whose transfer an earlier call has waited for - frame_take never lets more than FRAME_SLOTS - 1 frames go unawaited -
// and which the caller may hold only until this call.)
PtHandle::Frame &frame_slot(PtHandle *h, uint64_t seq) { return h->frames[seq % FRAME_SLOTS]; }
int frame_begin(PtHandle *h, uint32_t n, bool tiles, uint32_t first, uint32_t shown) {
    PtHandle::Frame &f = frame_slot(h, h->frames_begun + 1);
    const size_t bytes = (size_t)h->sc.width * h->sc.height * 3;
    h->flags_offset = (bytes + 255) & ~(size_t)255;
    const size_t total = h->flags_offset + FLAG_RING * 4;
    if (!f.d_pixels8) {
        int rc = dev_alloc(h, &f.d_pixels8, total);
        if (rc) return rc;
        // (hipMemset of device memory may return before it has run, on the null stream, which the handle's non-blocking
        // streams do not wait for: without the synchronisation it can land on top of the frame's display bytes)
        HIPCHK(h, hipMemset(f.d_pixels8, 0, total));
        HIPCHK(h, hipStreamSynchronize(nullptr));
    }
    if (!f.h_pixels8) HIPCHK(h, hipHostMalloc((void **)&f.h_pixels8, total, hipHostMallocDefault));
    const uint64_t before = h->launched;
    int rc = render_range(h, n, tiles, first, f.d_pixels8, shown, false);
    if (rc) return rc;
    ++h->frames_begun;
    f.seq = h->frames_begun;
    f.num_samples = shown ? shown : h->num_samples;
    f.last_order = h->launched;   // (the snapshot is final for every pass with order < last_order)
    f.has_flags = h->launched != before;
    f.in_flight = true;
    return PT_OK;
}

// The frame's way to the host, on the copy stream behind the frame's last pass: display bytes and flag snapshot in one
// transfer (`image_only`: the caller moves the bytes itself - a group, which gathers its members' tiles first).
int frame_copy(PtHandle *h, PtHandle::Frame &f, bool image_only = false) {
    if (h->passes > 0 && last_pipe(h).resolved_pending) HIPCHK(h, hipStreamWaitEvent(h->copy_stream, last_pipe(h).resolved, 0));
    if (image_only) {
        if (f.has_flags)
            HIPCHK(h, hipMemcpyAsync(f.h_pixels8 + h->flags_offset, f.d_pixels8 + h->flags_offset, FLAG_RING * 4, hipMemcpyDeviceToHost, h->copy_stream));
    } else {
        HIPCHK(h, hipMemcpyAsync(f.h_pixels8, f.d_pixels8, h->flags_offset + FLAG_RING * 4, hipMemcpyDeviceToHost, h->copy_stream));
    }
    HIPCHK(h, hipEventRecord(f.copied, h->copy_stream));
    return PT_OK;
}

// Waits for frame `f` and retires the passes its flag snapshot covers.  *voided: one of them overflowed a queue level -
// its samples are missing from the sums and from every image computed since; the caller recovers (frame_take).
int frame_wait(PtHandle *h, PtHandle::Frame &f, bool *voided) {
    HIPCHK(h, hipEventSynchronize(f.copied));
    f.in_flight = false;
    *voided = false;
    if (!f.has_flags) return PT_OK;
    const uint32_t *flags = reinterpret_cast<const uint32_t *>(f.h_pixels8 + h->flags_offset);
    size_t keep = 0;
    for (size_t k = 0; k < h->pending.size(); ++k) {
        const PtHandle::Pending &p = h->pending[k];
        if (p.order < f.last_order) {
            if (flags[p.slot]) { *voided = true; return PT_OK; }   // (pending stays whole: resolve_pending goes through it)
            continue;
        }
        h->pending[keep++] = p;
    }
    h->pending.resize(keep);
    return PT_OK;
}

void frames_reset(PtHandle *h) {
    for (PtHandle::Frame &f : h->frames) f.in_flight = false;
    h->frames_begun = 0;
}
// Which frame a call hands out: the newest whose transfer has arrived, but never one older than the frame two before
// the newest - that one is waited for, which is also what keeps the host from running further ahead of the GPU.  A
// call that began no frame (a flush) waits for the newest.  0: no frame has been begun since upload / clear.
template <class Arrived>
uint64_t frame_choice(uint64_t frames_begun, bool began_one, Arrived arrived) {
    if (frames_begun == 0) return 0;
    if (!began_one) return frames_begun;
    const uint64_t oldest = frames_begun > (uint64_t)(FRAME_SLOTS - 1) ? frames_begun - (FRAME_SLOTS - 1) : 1;
    for (uint64_t k = frames_begun; k > oldest; --k)
        if (arrived(k)) return k;
    return oldest;
}
// *out == nullptr: there is no frame to hand out (nothing was rendered through frames yet), or a pass overflowed a queue
// level and its samples are missing from the frames in flight - everything has then been synchronised and the voided
// passes rendered again with larger levels, the frame pipeline starts afresh, and the caller presents the sums as they
// stand.
int frame_take(PtHandle *h, bool began_one, PtHandle::Frame **out) {
    *out = nullptr;
    const uint64_t target = frame_choice(h->frames_begun, began_one, [h](uint64_t k) { return hipEventQuery(frame_slot(h, k).copied) == hipSuccess; });
    if (target == 0) return PT_OK;
    (void)hipGetLastError();   // (hipErrorNotReady of the queries above is no error)
    PtHandle::Frame &f = frame_slot(h, target);
    bool voided = false;
    int rc = frame_wait(h, f, &voided);
    if (rc) return rc;
    if (voided) {
        if ((rc = sync_pipes(h))) return rc;
        if ((rc = resolve_pending(h))) return rc;
        frames_reset(h);
        return PT_OK;
    }
    *out = &f;
    return PT_OK;
}

// ---- ptgpu_render without any allocation -------------------------------------------------------------------------
// Where the pieces of one pass live inside the scratch array for a given pass size.
struct ScratchPlan {
    uint32_t caps[25];
    size_t level_off[25];
    size_t flags_off, caps_off, counts_off, end;
};
bool plan_scratch(uint64_t pixels, int n_levels, bool glass, uint32_t growth, uint32_t spp, ScratchPlan &pl) {
    auto align = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t off = 0;
    pl.flags_off = off; off = align(off + NOALLOC_FLAGS * 4);
    pl.caps_off = off; off = align(off + 25 * 4);
    pl.counts_off = off; off = align(off + (size_t)(n_levels + 2) * NSHARD * CSTRIDE * 4);
    const uint64_t S = pixels * spp;
    for (int l = 0; l < n_levels; ++l) {
        const uint64_t f = glass ? std::min<uint64_t>(l < 31 ? (1ull << l) : (1ull << 31), growth) : 1;
        const uint64_t recs = S * f, per = (recs + NSHARD - 1) / NSHARD;
        const uint64_t cap = l == 0 ? ((per + 63) / 64) * 64 : ((per + per / 4 + 128 + 63) / 64) * 64;   // as alloc_levels sizes them
        if (cap * NSHARD >= (1ull << 31)) return false;
        pl.caps[l] = (uint32_t)cap;
        pl.level_off[l] = off;
        off = align(off + (size_t)cap * NSHARD * Level::RECORD_BYTES);
        if (off > (size_t)PTGPU_NOALLOC_SCRATCH_BYTES) return false;
    }
    pl.end = off;
    return true;
}
struct CapsArg { uint32_t v[25]; };
__global__ void k_set_caps(uint32_t *caps, CapsArg a, int n) {
    if ((int)threadIdx.x < n) caps[threadIdx.x] = a.v[threadIdx.x];
}

}  // namespace

extern "C" {

void ptgpu_default_config(PtConfig *cfg) {
    if (!cfg) return;
    std::memset(cfg, 0, sizeof *cfg);
    cfg->device = 0;
    cfg->seed = 314159265u;  // the reference's xorshift seed (src/common.rs:28), reused as the counter-RNG key
    cfg->max_depth = 4;      // src/tracer.rs:633
    cfg->sub_pixel_override = -1;
    cfg->shard_index = 0;
    cfg->shard_count = 1;
    cfg->samples_per_batch = 0;
    cfg->branch_capacity = 0;
    cfg->accel = PT_ACCEL_AUTO;
    cfg->fuse_trace_shade = -1;
    cfg->profile_events = 0;
    cfg->concurrent_passes = 0;
    cfg->camera_model = 0;
    cfg->focus_distance = 10.0f;  // src/tracer.rs:155
    cfg->lens_aperture = 4.0f;    // src/tracer.rs:177
}

int ptgpu_create(const PtConfig *cfg, PtHandle **out) {
    PT_API_BEGIN
    if (!cfg || !out) return PT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (cfg->max_depth < 1 || cfg->max_depth > 24 || cfg->shard_count == 0 || cfg->shard_index >= cfg->shard_count)
        return PT_ERR_INVALID_ARGUMENT;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0 || cfg->device < 0 || cfg->device >= n_dev) return PT_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return PT_ERR_NO_DEVICE;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return PT_ERR_NO_DEVICE;  // the kernels are built for gfx950 only
    if (hipSetDevice(cfg->device) != hipSuccess) return PT_ERR_HIP;
    PtHandle *h = new (std::nothrow) PtHandle();
    if (!h) return PT_ERR_OUT_OF_MEMORY;
    h->cfg = *cfg;
    h->device = cfg->device;
    h->error.reserve(ERROR_TEXT_RESERVE);
    if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) { delete h; return PT_ERR_HIP; }
    h->pipes[0].stream = h->stream;
    bool ok = true;
    for (int q = 1; q < PT_MAX_PIPES; ++q) ok = ok && hipStreamCreateWithFlags(&h->pipes[q].stream, hipStreamNonBlocking) == hipSuccess;
    for (Pipe &p : h->pipes) ok = ok && hipEventCreateWithFlags(&p.resolved, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&h->static_pipe.resolved, hipEventDisableTiming) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (PtHandle::Frame &f : h->frames) {
        ok = ok && hipEventCreateWithFlags(&f.copied, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) { ptgpu_destroy(h); return PT_ERR_HIP; }
    // this device's copy of the static scratch array (device cfg->device is current)
    if (cfg->device < PT_MAX_DEVICES) {
        static std::mutex table_lock;
        static unsigned char *table[PT_MAX_DEVICES] = {};
        std::lock_guard<std::mutex> guard(table_lock);
        if (!table[cfg->device]) {
            unsigned char *ptr = nullptr;
            if (hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) {
                (void)hipGetLastError();
                ptgpu_destroy(h);
                return PT_ERR_HIP;
            }
            table[cfg->device] = ptr;
        }
        h->scratch_ptr = table[cfg->device];
    }
    h->static_pipe.stream = h->stream;
    h->static_pipe.levels.resize(25);   // max_depth <= 24: sized here so that ptgpu_render_noalloc never allocates on the host either
    *out = h;
    return PT_OK;
    PT_API_END(nullptr)
}

void ptgpu_destroy(PtHandle *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    for (Pipe &p : h->pipes) if (p.stream) (void)hipStreamSynchronize(p.stream);
    if (h->copy_stream) (void)hipStreamSynchronize(h->copy_stream);
    free_scene(h);
    drain_level_pool(h);
    for (PtHandle::Frame &f : h->frames) {
        if (f.copied) (void)hipEventDestroy(f.copied);
    }
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (Pipe &p : h->pipes) if (p.resolved) (void)hipEventDestroy(p.resolved);
    if (h->static_pipe.resolved) (void)hipEventDestroy(h->static_pipe.resolved);
    for (int q = 1; q < PT_MAX_PIPES; ++q) if (h->pipes[q].stream) (void)hipStreamDestroy(h->pipes[q].stream);
    for (auto &s : h->spans) { (void)hipEventDestroy(s.start); (void)hipEventDestroy(s.stop); }
    for (auto e : h->free_events) (void)hipEventDestroy(e);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

int ptgpu_upload_scene(PtHandle *h, const PtSceneDesc *s) {
    PT_API_BEGIN
    if (!h || !s) return PT_ERR_INVALID_ARGUMENT;
    if (s->width == 0 || s->height == 0 || (uint64_t)s->width * s->height > (1ull << 28))
        return fail(h, PT_ERR_INVALID_ARGUMENT, "image size must be 1..2^28 pixels");
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_emissive_spheres && !s->emissive_spheres) ||
        (s->n_emissive_planes && !s->emissive_planes))
        return fail(h, PT_ERR_INVALID_ARGUMENT, "null primitive array with non-zero count");
    const uint64_t n_sph = s->n_spheres + s->n_emissive_spheres, n_pl = s->n_planes + s->n_emissive_planes;
    if (n_sph + n_pl >= (1ull << 28)) return fail(h, PT_ERR_INVALID_ARGUMENT, "too many primitives");
    HIPCHK(h, hipSetDevice(h->device));
    // every pipe, not only the handle's stream: a render call that failed midway never joined the others into it
    for (Pipe &p : h->pipes) if (p.stream) HIPCHK(h, hipStreamSynchronize(p.stream));
    if (h->copy_stream) HIPCHK(h, hipStreamSynchronize(h->copy_stream));
    free_scene(h);

    std::vector<float4> sph, pln, mat, lights;
    bool glass = false;
    flatten_scene(s, sph, pln, mat, lights, glass);
    h->has_glass = glass;

    int rc;
    if ((rc = dev_alloc(h, &h->d_sph, std::max<size_t>(1, sph.size())))) return rc;
    if ((rc = dev_alloc(h, &h->d_pln, std::max<size_t>(1, pln.size())))) return rc;
    if ((rc = dev_alloc(h, &h->d_mat, std::max<size_t>(1, mat.size())))) return rc;
    if ((rc = dev_alloc(h, &h->d_lights, std::max<size_t>(1, lights.size())))) return rc;
    if (!sph.empty()) HIPCHK(h, hipMemcpy(h->d_sph, sph.data(), sph.size() * 16, hipMemcpyHostToDevice));
    if (!pln.empty()) HIPCHK(h, hipMemcpy(h->d_pln, pln.data(), pln.size() * 16, hipMemcpyHostToDevice));
    if (!mat.empty()) HIPCHK(h, hipMemcpy(h->d_mat, mat.data(), mat.size() * 16, hipMemcpyHostToDevice));
    if (!lights.empty()) HIPCHK(h, hipMemcpy(h->d_lights, lights.data(), lights.size() * 16, hipMemcpyHostToDevice));

    DevScene &sc = h->sc;
    std::memset(&sc, 0, sizeof sc);
    sc.n_sph = (uint32_t)n_sph; sc.n_pl = (uint32_t)n_pl;
    sc.n_es = (uint32_t)s->n_emissive_spheres; sc.n_ep = (uint32_t)s->n_emissive_planes;
    sc.n_pl_plain = (uint32_t)s->n_planes;
    sc.sph = h->d_sph; sc.pln = h->d_pln; sc.mat = h->d_mat; sc.lights = h->d_lights;
    // CameraSampler step vectors, reference src/tracer.rs:104-111
    sc.cam_origin = toV3(s->camera.plane_origin);
    sc.cam_su = toV3(s->camera.plane_u) / (float)s->width;
    sc.cam_sv = toV3(s->camera.plane_v) / (float)s->height;
    sc.cam_pos = toV3(s->camera.position);
    sc.tone = s->camera.tone_mapping;
    sc.exposure = s->camera.exposure;
    sc.iso_factor = s->camera.iso / 100.0f;  // src/tracer.rs:621
    sc.sky_kind = s->sky.kind;
    sc.sky_radiance = toV3(s->sky.radiance);
    sc.sky_pixels = nullptr;
    if (s->sky.kind == PT_SKY_HDRI && s->sky.pixels && s->sky.width && s->sky.height) {
        size_t n = (size_t)s->sky.width * s->sky.height * 3;
        if ((rc = dev_alloc(h, &h->d_sky, n))) return rc;
        HIPCHK(h, hipMemcpy(h->d_sky, s->sky.pixels, n * 4, hipMemcpyHostToDevice));
        sc.sky_pixels = h->d_sky;
        sc.sky_w = (uint32_t)s->sky.width; sc.sky_h = (uint32_t)s->sky.height;
    }
    {   // step positions of trunc(255 * x^(1/2.2)) on [0,1], by bisection over the (monotone) f32 bit patterns
        std::vector<float> table(256, 0.0f);
        for (uint32_t k = 1; k < 256; ++k) {
            uint32_t lo = 0, hi = 0x3f800000u;  // quantise(lo) < k <= quantise(hi): 0 -> 0, 1.0 -> 255
            while (hi - lo > 1) {
                uint32_t mid = lo + (hi - lo) / 2;
                if (gamma_quantise_direct(u2f(mid)) >= k) hi = mid; else lo = mid;
            }
            table[k] = u2f(hi);
        }
        if ((rc = dev_alloc(h, &h->d_gamma, 256))) return rc;
        HIPCHK(h, hipMemcpy(h->d_gamma, table.data(), 256 * 4, hipMemcpyHostToDevice));
        sc.gamma_table = h->d_gamma;
    }
    sc.width = s->width; sc.height = s->height;
    sc.jitter = h->cfg.sub_pixel_override < 0 ? (s->sub_pixel_sampling != 0) : (h->cfg.sub_pixel_override != 0);
    sc.seed = h->cfg.seed;
    sc.max_depth = h->cfg.max_depth;
    sc.thin_lens = h->cfg.camera_model == 1;
    sc.focus_distance = h->cfg.focus_distance;
    sc.circle_of_confusion = 22.0f / h->cfg.lens_aperture;  // src/tracer.rs:181

    // ---- acceleration structure
    // Automatic choice: the fused linear-scan kernels stay ahead of the BVH pipeline (separate trace kernel, dependent
    // node fetches) up to about 100 primitives - measured on the closed room with N spheres at 1080p, depth 8
    // (profiles/debug/accel_crossover.py): 27 primitives 23.0 vs 12.8 Grays/s, 87: 12.6 vs 11.1, 117: 10.4 vs 10.6,
    // 177: 7.6 vs 10.3, 237: 5.6 vs 9.9.
    bool want_bvh = h->cfg.accel == PT_ACCEL_BVH || (h->cfg.accel == PT_ACCEL_AUTO && n_sph + n_pl > 100);
    h->bvh_nodes = h->bvh_leaf_prims = 0;
    h->bvh_depth = h->bvh_binary_depth = h->bvh_forced_median_splits = h->bvh_level_fold = 0;
    h->lds_bytes = 0;
    if (want_bvh && n_sph + n_pl > 0) {
        BvhBuilder builder;
        builder.build(sph, pln, toV3(s->camera.position), toV3(s->camera.plane_origin));
        if (builder.unit_prim.size() > BVH_MAX_UNITS) {
            return fail(h, PT_ERR_INVALID_ARGUMENT, "scene too large for the acceleration structure (more than 2^26 nodes and primitives)");
        }
        if ((rc = dev_alloc(h, &h->d_bvh_nodes, builder.units.size()))) return rc;
        HIPCHK(h, hipMemcpy(h->d_bvh_nodes, builder.units.data(), builder.units.size() * sizeof(float4), hipMemcpyHostToDevice));
        if ((rc = dev_alloc(h, &h->d_bvh_always, std::max<size_t>(1, builder.always.size())))) return rc;
        if (!builder.always.empty())
            HIPCHK(h, hipMemcpy(h->d_bvh_always, builder.always.data(), builder.always.size() * 4, hipMemcpyHostToDevice));
        sc.bvh_nodes = h->d_bvh_nodes;  // (the spill area belongs to the pipes)
        sc.bvh_always = h->d_bvh_always; sc.n_always = (uint32_t)builder.always.size();
        apply_builder_limits(sc, builder);
        {
            const std::vector<uint32_t> order = shading_order(builder);
            const std::vector<uint32_t> prim_rows = rows_of_primitives(order, n_sph + n_pl);
            if ((rc = dev_alloc(h, &h->d_prim_rows, prim_rows.size()))) return rc;
            HIPCHK(h, hipMemcpy(h->d_prim_rows, prim_rows.data(), prim_rows.size() * 4, hipMemcpyHostToDevice));
            sc.prim_rows = h->d_prim_rows;
            std::vector<float4> rows(4 * std::max<size_t>(1, order.size()), make_float4(0.f, 0.f, 0.f, 0.f));
            for (size_t k = 0; k < order.size(); ++k) {
                const uint32_t id = order[k];
                if (id == BvhBuilder::MISS_ID) continue;
                float4 m1 = mat[4 * (size_t)id + 1];
                m1.w = u2f(id);
                float4 geo = id < n_sph ? sph[id] : pln[4 * (size_t)(id - n_sph)];   // sphere centre / plane normal in .xyz
                geo.w = 0.0f;
                rows[4 * k] = mat[4 * (size_t)id];
                rows[4 * k + 1] = m1;
                rows[4 * k + 2] = mat[4 * (size_t)id + 2];
                rows[4 * k + 3] = geo;
            }
            if ((rc = dev_alloc(h, &h->d_shade_rows, rows.size()))) return rc;
            HIPCHK(h, hipMemcpy(h->d_shade_rows, rows.data(), rows.size() * sizeof(float4), hipMemcpyHostToDevice));
            sc.shade_rows = h->d_shade_rows;
            sc.n_bvh_units = (uint32_t)builder.unit_prim.size();
        }
        h->bvh_nodes = builder.n_nodes; h->bvh_leaf_prims = builder.n_leaf_prims;
        h->bvh_depth = (uint64_t)builder.depth4; h->bvh_binary_depth = (uint64_t)builder.max_depth_seen;
        h->bvh_forced_median_splits = builder.forced_median_splits; h->bvh_level_fold = builder.level_fold ? 1 : 0;
        h->lds_bytes = (size_t)BVH_LDS_ENTRIES * BVH_STACK_STRIDE * sizeof(uint2);
    }
    if (sc.bvh_nodes == nullptr && n_sph + n_pl > 0) {   // linear scan: the box of everything, for the primary rays' early out
        BvhBuilder bounds;
        bounds.boxes(sph, pln, toV3(s->camera.position), toV3(s->camera.plane_origin));
        float lo[3], hi[3];
        if (bounds.scene_bounds(lo, hi)) {
            sc.has_bounds = 1;
            sc.bounds_lo = v3(lo[0], lo[1], lo[2]);
            sc.bounds_hi = v3(hi[0], hi[1], hi[2]);
        }
    }
    // auto: BVH scenes run intersection as its own kernel (dynamic ray fetch), small scenes fuse it with shading
    h->fused = h->cfg.fuse_trace_shade < 0 ? (sc.bvh_nodes == nullptr) : (h->cfg.fuse_trace_shade != 0);
    h->lds_scene = sc.bvh_nodes == nullptr && lds_scene_fits(sc.n_sph, sc.n_pl);
    h->regroup = h->cfg.fuse_trace_shade == 2;  // measured no faster than the plain fused kernel: opt-in only
    h->ahead = h->cfg.fuse_trace_shade == 3 && sc.bvh_nodes == nullptr;

    // ---- accumulators and queues
    const size_t P = (size_t)s->width * s->height;
    if ((rc = dev_alloc(h, &h->d_pixels32, P * 3))) return rc;
    if ((rc = dev_alloc(h, &h->d_pixels8, P * 3))) return rc;
    if ((rc = dev_alloc(h, &h->d_ids, P))) return rc;
    if (h->cfg.shard_count > 1) {   // the pixels of this handle's tiles, for ptgpu_render_tiles
        std::vector<uint32_t> mine;
        const uint32_t tiles_x = (s->width + PT_TILE - 1) / PT_TILE, tiles_y = (s->height + PT_TILE - 1) / PT_TILE;
        for (uint32_t ty = 0; ty < tiles_y; ++ty)
            for (uint32_t tx = 0; tx < tiles_x; ++tx) {
                if (ptgpu_tile_shard(tx, ty, tiles_x, h->cfg.shard_count) != h->cfg.shard_index) continue;
                // clipped to the image like the tiles of the CPU pool (the reference never clips, src/window.rs:167-178)
                for (uint32_t y = ty * PT_TILE; y < std::min(s->height, (ty + 1) * PT_TILE); ++y)
                    for (uint32_t x = tx * PT_TILE; x < std::min(s->width, (tx + 1) * PT_TILE); ++x) mine.push_back(y * s->width + x);
            }
        h->n_tile_pixels = (uint32_t)mine.size();
        if ((rc = dev_alloc(h, &h->d_tile_pixels, std::max<size_t>(1, mine.size())))) return rc;
        if (!mine.empty()) HIPCHK(h, hipMemcpy(h->d_tile_pixels, mine.data(), mine.size() * 4, hipMemcpyHostToDevice));
    }
    if ((rc = dev_alloc(h, &h->d_ctr, 1))) return rc;
    if ((rc = dev_alloc(h, &h->d_flags, (size_t)FLAG_RING))) return rc;
    HIPCHK(h, hipMemset(h->d_flags, 0, FLAG_RING * 4));
    HIPCHK(h, hipStreamSynchronize(nullptr));   // (see frame_begin: the passes' streams do not wait for the null stream)
    h->fixed_capacity = h->cfg.branch_capacity != 0;
    // Levels start at one record per sample and level (plus the 25 % slack of every shard): in the reference's scene 0.57 -
    // 0.71 records per sample reach levels 1-3, glass included.  Scenes that branch more overflow once and grow.
    h->growth = h->fixed_capacity ? h->cfg.branch_capacity : 1u;
    h->flag_cursor = 0;
    h->pending.clear();
    if ((rc = alloc_levels(h))) return rc;
    h->have_scene = true;
    return ptgpu_clear(h);
    PT_API_END(h)
}

int ptgpu_clear(PtHandle *h) {
    PT_API_BEGIN
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    int rc0 = sync_pipes(h);
    if (rc0) return rc0;
    h->pending.clear();  // whatever those passes were worth is being thrown away
    const size_t P = (size_t)h->sc.width * h->sc.height;
    HIPCHK(h, hipMemsetAsync(h->d_pixels32, 0, P * 3 * 4, h->stream));
    HIPCHK(h, hipMemsetAsync(h->d_pixels8, 0, P * 3, h->stream));
    for (PtHandle::Frame &f : h->frames) {
        if (f.d_pixels8) HIPCHK(h, hipMemsetAsync(f.d_pixels8, 0, P * 3, h->stream));   // (a tile shard only ever writes its own tiles)
    }
    frames_reset(h);
    HIPCHK(h, hipMemsetAsync(h->d_ctr, 0, sizeof(Counters), h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    int rc = drain_spans(h);
    if (rc) return rc;
    h->num_samples = 0;
    h->local_samples = 0;
    h->tile_samples = 0;
    h->sample_cursor = 0;
    std::memset(h->kernel_launches, 0, sizeof h->kernel_launches);
    std::memset(h->kernel_ms, 0, sizeof h->kernel_ms);
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_render(PtHandle *h, uint32_t n_samples) {
    PT_API_BEGIN
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    return render_range(h, n_samples, false, 0u, nullptr, 0u);
    PT_API_END(h)
}

int ptgpu_render_tiles(PtHandle *h, uint64_t first_sample, uint32_t n_samples) {
    PT_API_BEGIN
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    if (first_sample + n_samples > 0xFFFFFFFFull) return fail(h, PT_ERR_INVALID_ARGUMENT, "sample indices are 32-bit");
    HIPCHK(h, hipSetDevice(h->device));
    return render_range(h, n_samples, true, (uint32_t)first_sample, nullptr, 0u);
    PT_API_END(h)
}

uint32_t ptgpu_tile_shard(uint32_t tile_x, uint32_t tile_y, uint32_t tiles_x, uint32_t n) {
    if (n <= 1) return 0;
    // tile_id mod n, with the rows shifted against each other when a row holds a whole number of rounds (else every
    // shard would own whole columns of tiles)
    const uint64_t id = (uint64_t)tile_y * tiles_x + tile_x + (tiles_x % n == 0 ? tile_y : 0u);
    return (uint32_t)(id % n);
}

int ptgpu_frame(PtHandle *h, uint32_t n_samples, const uint8_t **pixels, uint64_t *num_samples) {
    PT_API_BEGIN
    if (!h || !pixels) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    int rc;
    if (n_samples > 0) {
        if ((rc = frame_begin(h, n_samples, false, 0u, 0u))) return rc;
        if ((rc = frame_copy(h, frame_slot(h, h->frames_begun)))) return rc;
    }
    PtHandle::Frame *f = nullptr;
    if ((rc = frame_take(h, n_samples > 0, &f))) return rc;
    if (!f) {   // (no frame to hand out: the display bytes of the sums as they stand, the synchronous way)
        if ((rc = ptgpu_present_pixels8(h, pixels))) return rc;
        if (num_samples) *num_samples = h->num_samples;
        return PT_OK;
    }
    *pixels = f->h_pixels8;
    if (num_samples) *num_samples = f->num_samples;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_render_noalloc(PtHandle *h, uint32_t n_samples) {
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    // Everything below reports through fixed strings (api_caught): composing a message would allocate.  Nothing in the
    // body allocates or throws by design; the handler is there so that nothing can cross the C boundary all the same.
    try {
    if (!h->have_scene) return api_caught(h, PT_ERR_NO_SCENE, "no scene uploaded");
    if (!h->scratch_ptr) return api_caught(h, PT_ERR_INVALID_ARGUMENT, "ptgpu_render_noalloc: no static scratch on this device ordinal");
    if (h->sc.bvh_nodes)
        return api_caught(h, PT_ERR_INVALID_ARGUMENT,
                          "ptgpu_render_noalloc: scenes traced through a BVH (more than 100 primitives, or PT_ACCEL_BVH) need a "
                          "traversal spill area larger than the static scratch; use ptgpu_render");
    if (n_samples == 0) return PT_OK;
#define NA_HIP(call) do { if ((call) != hipSuccess) { (void)hipGetLastError(); h->no_spans = false; return api_caught(h, PT_ERR_HIP, "ptgpu_render_noalloc: " #call " failed"); } } while (0)
    NA_HIP(hipSetDevice(h->device));
    // passes of earlier ptgpu_render calls add into the same accumulator from other streams: let them finish (their
    // overflow flags stay pending for the next ptgpu_sync)
    for (Pipe &p : h->pipes) if (p.stream) NA_HIP(hipStreamSynchronize(p.stream));
    const uint64_t P = (uint64_t)h->sc.width * h->sc.height;
    const int n_levels = h->cfg.max_depth + 1;
    uint32_t growth = h->fixed_capacity ? h->cfg.branch_capacity : 1u;
    ScratchPlan pl;
    auto largest_pass = [&](uint32_t want) -> uint32_t {   // most samples per pass (<= want) whose queues fit the scratch; 0: not even one
        uint32_t spp = std::min<uint32_t>(want, 64);
        while (spp > 0 && !plan_scratch(P, n_levels, h->has_glass, growth, spp, pl)) spp /= 2;
        return spp;
    };
    uint32_t spp = largest_pass(n_samples);
    if (spp == 0)
        return api_caught(h, PT_ERR_INVALID_ARGUMENT,
                          "ptgpu_render_noalloc: the ray queues of one sample per pixel of this image at this max_depth exceed the "
                          "static scratch (PTGPU_NOALLOC_SCRATCH_BYTES); use ptgpu_render or a smaller image");
    Pipe &pipe = h->static_pipe;
    pipe.sc = h->sc;
    h->no_spans = true;
    struct Range { uint32_t first, count; };
    constexpr int STACK = 512;
    Range stack[STACK];
    int sp = 0;
    stack[sp++] = {(uint32_t)h->sample_cursor, n_samples};
    h->sample_cursor += n_samples;
    bool planned = false;
    uint64_t lost = 0;   // samples of passes that overflowed and could not be repeated
    Range launched[NOALLOC_FLAGS];
    uint32_t flags[NOALLOC_FLAGS];
    uint32_t n_launched = 0;
    // Waits for the passes launched in this round and counts the samples of those that were accumulated, so that the
    // accumulator and what it is divided by stay in step also when the call ends early; false: the device cannot be
    // asked any more (then neither count means anything).
    auto settle = [&]() -> bool {
        if (hipStreamSynchronize(h->stream) != hipSuccess) return false;
        if (n_launched && hipMemcpy(flags, h->scratch_ptr + pl.flags_off, n_launched * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
        for (uint32_t k = 0; k < n_launched; ++k)
            if (!flags[k]) { h->local_samples += launched[k].count; h->num_samples += launched[k].count; }
        return true;
    };
    while (sp > 0) {
        if (!planned) {   // (re)carve the scratch for the current growth and pass size
            unsigned char *base = h->scratch_ptr;
            pipe.d_counts = (uint32_t *)(base + pl.counts_off);
            pipe.d_shard_caps = (uint32_t *)(base + pl.caps_off);
            CapsArg caps{};
            for (int l = 0; l < n_levels; ++l) {
                Level &L = pipe.levels[l];
                L.base = (char *)(base + pl.level_off[l]);
                L.count = pipe.d_counts + (size_t)l * NSHARD * CSTRIDE;
                L.shard_cap = pl.caps[l];
                caps.v[l] = pl.caps[l];
            }
            // the array keeps whatever the last call on this GPU left in it: counters start at zero
            NA_HIP(hipMemsetAsync(base, 0, pl.level_off[0], h->stream));
            k_set_caps<<<1, 32, 0, h->stream>>>(pipe.d_shard_caps, caps, n_levels);
            planned = true;
        }
        Range r = stack[--sp];
        // one round: up to NOALLOC_FLAGS passes of this range, each with its own overflow flag
        n_launched = 0;
        while (r.count > 0 && n_launched < NOALLOC_FLAGS) {
            const uint32_t k = std::min(spp, r.count);
            uint32_t *flag = (uint32_t *)(h->scratch_ptr + pl.flags_off) + n_launched;
            int rc = run_pass(h, pipe, nullptr, own_shard_pass(h, k, r.first), flag);
            if (rc) {   // (the passes already enqueued in this round still run and add their samples: count them)
                (void)hipGetLastError();
                (void)settle();
                h->no_spans = false;
                return rc;
            }
            launched[n_launched++] = {r.first, k};
            r.first += k;
            r.count -= k;
        }
        if (r.count > 0) stack[sp++] = r;
        if (!settle()) {
            (void)hipGetLastError();
            h->no_spans = false;
            return api_caught(h, PT_ERR_HIP, "ptgpu_render_noalloc: waiting for the passes of this call failed");
        }
        bool any = false;
        for (uint32_t k = 0; k < n_launched; ++k) {
            if (!flags[k]) continue;
            any = true;
            if (sp < STACK) stack[sp++] = launched[k]; else lost += launched[k].count;
        }
        n_launched = 0;
        if (any) {   // glass branched beyond the levels: the voided passes are on the stack again; make room for twice the branching
            uint32_t next_spp = 0;
            if (!h->fixed_capacity && growth < (1u << 30)) {
                growth *= 2;
                next_spp = largest_pass(spp);
            }
            if (next_spp == 0) {   // cannot grow inside the scratch: everything still to do is dropped
                for (int k = 0; k < sp; ++k) lost += stack[k].count;
                sp = 0;
                break;
            }
            spp = next_spp;
            planned = false;
        }
    }
    h->no_spans = false;
#undef NA_HIP
    if (lost)
        return api_caught(h, PT_ERR_QUEUE_OVERFLOW,
                          "ptgpu_render_noalloc: glass branching overflowed the queues that fit the static scratch; some samples of this "
                          "call were not accumulated (use ptgpu_render, a smaller image or a smaller max_depth)");
    return PT_OK;
    } catch (...) {
        h->no_spans = false;
        return api_caught(h, PT_ERR_INTERNAL, "ptgpu_render_noalloc: unexpected C++ exception");
    }
}

int ptgpu_sync(PtHandle *h) {
    PT_API_BEGIN
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    HIPCHK(h, hipSetDevice(h->device));
    int rc0 = sync_pipes(h);
    if (rc0) return rc0;
    int rc = drain_spans(h);
    if (rc) return rc;
    if (h->have_scene) return resolve_pending(h);
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_read_pixels32(PtHandle *h, uint32_t *out, uint64_t *num_samples) {
    PT_API_BEGIN
    if (!h || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_sync(h);
    if (rc) return rc;
    const size_t P = (size_t)h->sc.width * h->sc.height;
    HIPCHK(h, hipMemcpy(out, h->d_pixels32, P * 3 * 4, hipMemcpyDeviceToHost));
    if (num_samples) *num_samples = h->num_samples;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_read_pixels8(PtHandle *h, uint8_t *out) {
    PT_API_BEGIN
    if (!h || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_sync(h);
    if (rc) return rc;
    const uint32_t n = h->sc.width * h->sc.height * 3;
    k_pixels8<<<std::min<uint32_t>(4096, (n + BLOCK - 1) / BLOCK), BLOCK, 0, h->stream>>>(h->d_pixels32, h->d_pixels8, n, h->sc.iso_factor,
                                                                                         (uint32_t)h->num_samples);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_pixels8, n, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_present_pixels8(PtHandle *h, const uint8_t **pixels) {
    PT_API_BEGIN
    if (!h || !pixels) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    const uint32_t n = h->sc.width * h->sc.height * 3;
    if (!h->h_pixels8) HIPCHK(h, hipHostMalloc((void **)&h->h_pixels8, n, hipHostMallocDefault));
    int rc = ptgpu_read_pixels8(h, h->h_pixels8);
    if (rc) return rc;
    *pixels = h->h_pixels8;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_render_primary_ids(PtHandle *h, uint32_t sample_index, uint32_t *out) {
    PT_API_BEGIN
    if (!h || !out) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    const uint32_t n = h->sc.width * h->sc.height;
    const uint32_t id_blocks = std::min<uint32_t>(BVH_TRAV_BLOCKS, (n + BLOCK - 1) / BLOCK);
    if (h->sc.bvh_nodes) k_primary_ids<true><<<id_blocks, BLOCK, h->lds_bytes, h->stream>>>(h->pipes[0].sc, sample_index, h->d_ids);
    else k_primary_ids<false><<<id_blocks, BLOCK, h->lds_bytes, h->stream>>>(h->pipes[0].sc, sample_index, h->d_ids);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(out, h->d_ids, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_device_pixels32(PtHandle *h, void **device_ptr, uint64_t *n_words) {
    PT_API_BEGIN
    if (!h || !device_ptr) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_sync(h);
    if (rc) return rc;
    *device_ptr = h->d_pixels32;
    if (n_words) *n_words = (uint64_t)h->sc.width * h->sc.height * 3;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_set_num_samples(PtHandle *h, uint64_t num_samples) {
    PT_API_BEGIN
    if (!h) return PT_ERR_INVALID_ARGUMENT;
    h->num_samples = num_samples;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_export_pixels32(PtHandle *h, void *dst_device) {
    PT_API_BEGIN
    if (!h || !dst_device) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_sync(h);
    if (rc) return rc;
    const size_t bytes = (size_t)h->sc.width * h->sc.height * 3 * 4;
    HIPCHK(h, hipMemcpyAsync(dst_device, h->d_pixels32, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_import_pixels32(PtHandle *h, const void *src_device, uint64_t num_samples) {
    PT_API_BEGIN
    if (!h || !src_device) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    int rc = ptgpu_sync(h);  // passes still in flight (and repeats of voided ones) add into the accumulator this call replaces
    if (rc) return rc;
    const size_t bytes = (size_t)h->sc.width * h->sc.height * 3 * 4;
    HIPCHK(h, hipMemcpyAsync(h->d_pixels32, src_device, bytes, hipMemcpyDeviceToDevice, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->num_samples = num_samples;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_stream(PtHandle *h, void **stream) {
    PT_API_BEGIN
    if (!h || !stream) return PT_ERR_INVALID_ARGUMENT;
    *stream = (void *)h->stream;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_get_stats(PtHandle *h, PtStats *out) {
    PT_API_BEGIN
    if (!h || !out) return PT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof *out);
    if (h->have_scene) {
        int rc = ptgpu_sync(h);
        if (rc && rc != PT_ERR_QUEUE_OVERFLOW) return rc;
        Counters c;
        HIPCHK(h, hipMemcpy(&c, h->d_ctr, sizeof c, hipMemcpyDeviceToHost));
        out->rays = c.rays + c.shadow;
        out->shadow_rays = c.shadow;
        for (int k = 0; k < 8; ++k) out->debug_cycles[k] = c.stamps[k];
        out->records = c.records;
        out->bvh_steps = c.bvh_steps;
        out->queue_growth = h->has_glass ? h->growth : 1u;
        out->terminal_records = c.terminal;
        for (int l = 0; l < 25; ++l) out->level_records[l] = c.level_records[l];
        const uint64_t P = (uint64_t)h->sc.width * h->sc.height;
        out->samples = h->local_samples * P + h->tile_samples * (h->d_tile_pixels ? (uint64_t)h->n_tile_pixels : P);
    }
    out->bvh_nodes = h->bvh_nodes;
    out->bvh_leaf_prims = h->bvh_leaf_prims;
    out->bvh_depth = h->bvh_depth;
    out->bvh_binary_depth = h->bvh_binary_depth;
    out->bvh_forced_median_splits = h->bvh_forced_median_splits;
    out->bvh_level_fold = h->bvh_level_fold;
    out->queue_bytes = h->level_bytes;
    out->samples_per_pass = h->batch;
    out->passes_in_flight = (uint64_t)h->n_pipes;
    for (int k = 0; k < PT_KERNEL_COUNT; ++k) { out->kernel_launches[k] = h->kernel_launches[k]; out->kernel_ms[k] = h->kernel_ms[k]; }
    out->device_bytes = h->device_bytes;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_bvh_build_stats(const PtSceneDesc *s, PtStats *out) {
    PT_API_BEGIN
    if (!s || !out) return PT_ERR_INVALID_ARGUMENT;
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_emissive_spheres && !s->emissive_spheres) ||
        (s->n_emissive_planes && !s->emissive_planes))
        return PT_ERR_INVALID_ARGUMENT;
    if (s->n_spheres + s->n_emissive_spheres + s->n_planes + s->n_emissive_planes >= (1ull << 28)) return PT_ERR_INVALID_ARGUMENT;
    std::memset(out, 0, sizeof *out);
    std::vector<float4> sph, pln, mat, lights;
    bool glass = false;
    flatten_scene(s, sph, pln, mat, lights, glass);
    BvhBuilder builder;
    builder.build(sph, pln, toV3(s->camera.position), toV3(s->camera.plane_origin));
    out->bvh_nodes = builder.n_nodes;
    out->bvh_leaf_prims = builder.n_leaf_prims;
    out->bvh_depth = (uint64_t)builder.depth4;
    out->bvh_binary_depth = (uint64_t)builder.max_depth_seen;
    out->bvh_forced_median_splits = builder.forced_median_splits;
    out->bvh_level_fold = builder.level_fold ? 1 : 0;
    // every primitive inside the (quantised) box of every node above it, each reachable exactly once
    return builder.check() == 0 ? PT_OK : PT_ERR_INTERNAL;
    PT_API_END(nullptr)
}

// The traversal functions of the kernels, compiled for the host, over caller-supplied rays: builds the tree for `s` and
// walks it on the CPU, one ray after the other, with every unit reference and stack slot checked.
int ptgpu_bvh_host_trace(const PtSceneDesc *s, uint64_t n_rays, const float *rays, uint32_t *prims, float *ts, uint64_t *faults) {
    return ptgpu_bvh_host_trace_flags(s, n_rays, rays, prims, ts, faults, 0u);
}

int ptgpu_bvh_host_trace_flags(const PtSceneDesc *s, uint64_t n_rays, const float *rays, uint32_t *prims, float *ts, uint64_t *faults, uint32_t flags) {
    PT_API_BEGIN
    if (!s || (n_rays && (!rays || !prims || !ts)) || !faults) return PT_ERR_INVALID_ARGUMENT;
    if ((s->n_spheres && !s->spheres) || (s->n_planes && !s->planes) || (s->n_emissive_spheres && !s->emissive_spheres) ||
        (s->n_emissive_planes && !s->emissive_planes))
        return PT_ERR_INVALID_ARGUMENT;
    if (s->n_spheres + s->n_emissive_spheres + s->n_planes + s->n_emissive_planes >= (1ull << 28)) return PT_ERR_INVALID_ARGUMENT;
    std::vector<float4> sph, pln, mat, lights;
    bool glass = false;
    flatten_scene(s, sph, pln, mat, lights, glass);
    BvhBuilder builder;
    builder.build(sph, pln, toV3(s->camera.position), toV3(s->camera.plane_origin));
    DevScene sc{};
    sc.n_sph = (uint32_t)sph.size(); sc.n_pl = (uint32_t)(pln.size() / 4);
    sc.sph = sph.data(); sc.pln = pln.data();
    sc.bvh_nodes = builder.units.data();
    sc.n_bvh_units = (uint32_t)builder.unit_prim.size();
    sc.bvh_always = builder.always.data(); sc.n_always = (uint32_t)builder.always.size();
    apply_builder_limits(sc, builder);
    const std::vector<uint32_t> prim_rows = rows_of_primitives(shading_order(builder), sph.size() + pln.size() / 4);
    sc.prim_rows = prim_rows.data();
    // one lane's stack, laid out and strided exactly as a lane of the kernels sees it
    std::vector<unsigned long long> lds((size_t)BVH_LDS_ENTRIES * BVH_STACK_STRIDE);
    unsigned long long *spill = (unsigned long long *)std::calloc((size_t)(BVH_STACK - BVH_LDS_ENTRIES) * BVH_TRAV_BLOCKS * BVH_STACK_STRIDE, 8);
    if (!spill) return PT_ERR_INTERNAL;
    TravStack stk;
    stk.lds = (lds_u64 *)(uintptr_t)lds.data();
    stk.spill = spill;
    BvhHostCheck &chk = bvh_host_check();
    chk = BvhHostCheck();
    chk.unit_prim = builder.unit_prim.data();
    chk.ungated = (flags & PT_HOST_TRACE_UNGATED) != 0u;
    uint64_t steps = 0;
    for (uint64_t r = 0; r < n_rays; ++r) {
        Trav st;
        bvh_begin(sc, v3(rays[6 * r], rays[6 * r + 1], rays[6 * r + 2]), v3(rays[6 * r + 3], rays[6 * r + 4], rays[6 * r + 5]), st);
        // (leaf primitives wait for every second step, as in the trace kernel; a walk that does not end is a fault too)
        uint64_t it = 0;
        const uint64_t limit = 64ull * builder.unit_prim.size() + 64;
        for (; st.node != BVH_EMPTY && it < limit; ++it) steps += bvh_step(sc, st, stk, (it & 1u) == 1u);
        if (st.node != BVH_EMPTY) ++chk.bad_units;
        bvh_finish(sc, st);
        if (st.best != MISS && (st.best_slot >= prim_rows.size() + builder.unit_prim.size() || prim_rows[st.best] != st.best_slot)) ++chk.bad_kinds;   // (the hit's shading row)
        prims[r] = st.best;
        ts[r] = st.best_t;
    }
    std::free(spill);
    faults[0] = chk.bad_units; faults[1] = chk.bad_kinds; faults[2] = chk.bad_slots; faults[3] = steps;
    chk.unit_prim = nullptr;
    chk.ungated = false;
    return PT_OK;
    PT_API_END(nullptr)
}

const char *ptgpu_last_error(PtHandle *h) { return h ? h->error.c_str() : "null handle"; }

// Camera::new + construct_projection_plane, reference src/tracer.rs:69-96
void ptgpu_camera_new(PtVec3 position, PtVec3 target, PtVec3 up, float width, float height, float z_near, int32_t tone_mapping,
                      float exposure, float iso, PtCamera *out) {
    if (!out) return;
    V3 pos = toV3(position);
    V3 z = normalize(pos - toV3(target));
    V3 x = normalize(cross(toV3(up), z));
    V3 y = normalize(cross(z, x));
    float half_width = width / 2.0f, half_height = height / 2.0f;
    V3 to_plane_center = (-z_near) * z;
    V3 origin = ((pos + to_plane_center) - x * half_width) - y * half_height;
    V3 u = x * width, v = y * height;
    out->plane_origin = {origin.x, origin.y, origin.z};
    out->plane_u = {u.x, u.y, u.z};
    out->plane_v = {v.x, v.y, v.z};
    out->position = position;
    out->width = width; out->height = height; out->z_near = z_near;
    out->tone_mapping = tone_mapping; out->exposure = exposure; out->iso = iso;
}

// Plane::new, reference src/scene.rs:106
void ptgpu_plane_normal(PtVec3 u, PtVec3 v, PtVec3 *out) {
    if (!out) return;
    V3 n = normalize(cross(toV3(u), toV3(v)));
    *out = {n.x, n.y, n.z};
}

void ptgpu_plane_normals(PtPlane *planes, uint64_t n) {
    if (!planes) return;
    for (uint64_t i = 0; i < n; ++i) ptgpu_plane_normal(planes[i].u, planes[i].v, &planes[i].normal);
}

int ptgpu_selftest_gamma(PtHandle *h, uint64_t *mismatches) {
    PT_API_BEGIN
    if (!h || !mismatches) return PT_ERR_INVALID_ARGUMENT;
    if (!h->have_scene) return fail(h, PT_ERR_NO_SCENE, "no scene uploaded");
    HIPCHK(h, hipSetDevice(h->device));
    unsigned long long *d = nullptr, host = 0;
    HIPCHK(h, hipMalloc((void **)&d, 8));
    hipError_t e = hipMemsetAsync(d, 0, 8, h->stream);
    if (e == hipSuccess) {
        k_check_gamma<<<4096, 256, 0, h->stream>>>(h->sc.gamma_table, 0u, 0x3f800000u, d);          // +0 .. 1.0
        k_check_gamma<<<64, 256, 0, h->stream>>>(h->sc.gamma_table, 0x3f800001u, 0x3f810000u, d);    // just above 1
        k_check_gamma<<<64, 256, 0, h->stream>>>(h->sc.gamma_table, 0x80000000u, 0x80010000u, d);    // -0 and small negatives
        k_check_gamma<<<64, 256, 0, h->stream>>>(h->sc.gamma_table, 0x7f7f0000u, 0x7fc10000u, d);    // huge, inf, NaNs
        e = hipMemcpyAsync(&host, d, 8, hipMemcpyDeviceToHost, h->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    (void)hipFree(d);
    if (e != hipSuccess) return fail(h, PT_ERR_HIP, hipGetErrorString(e));
    *mismatches = host;
    return PT_OK;
    PT_API_END(h)
}

int ptgpu_eval_math(PtHandle *h, int32_t fn, const float *a, const float *b, float *out, uint64_t n) {
    PT_API_BEGIN
    if (!h || !a || !out || fn < 0 || fn > 11) return PT_ERR_INVALID_ARGUMENT;
    if (n == 0) return PT_OK;
    HIPCHK(h, hipSetDevice(h->device));
    float *da = nullptr, *db = nullptr, *dout = nullptr;
    hipError_t e = hipMalloc((void **)&da, n * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&dout, n * 4);
    if (e == hipSuccess && b) e = hipMalloc((void **)&db, n * 4);
    if (e == hipSuccess) e = hipMemcpy(da, a, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && b) e = hipMemcpy(db, b, n * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        k_eval_math<<<(unsigned)std::min<uint64_t>(4096, (n + 255) / 256), 256, 0, h->stream>>>(fn, da, db, dout, n);
        e = hipStreamSynchronize(h->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(out, dout, n * 4, hipMemcpyDeviceToHost);
    (void)hipFree(da); (void)hipFree(dout); (void)hipFree(db);   // (hipFree(nullptr) is a no-op)
    if (e != hipSuccess) return fail(h, PT_ERR_HIP, hipGetErrorString(e));
    return PT_OK;
    PT_API_END(h)
}

}  // extern "C"

#include "pt_group.h"
