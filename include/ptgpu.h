/*
 * ptgpu.h -- C ABI of libptgpu, the MI355X (gfx950) replacement for the CPU tile pool of
 * hpatjens/Pathtracer.
 *
 * What it replaces (paths relative to the reference checkout):
 *   - src/window.rs:109-116,166-196   WorkerPool::new(8, |tile| tracer::render(tile, &backbuffer, scene))
 *                                     + process(tile) x tiles + wait()      -> ptgpu_render()
 *   - src/tracer.rs:618-647           tracer::render(WorkTile, &Backbuffer, Arc<RwLock<Scene>>)
 *   - src/tracer.rs:200-283           Backbuffer {num_samples, pixels32, pixels8}, clear()
 *                                                                            -> ptgpu_clear / ptgpu_read_*
 *   - src/scene.rs:71-167             Scene / Sphere / Plane / Material / Sky data model (inputs only)
 *                                                                            -> PtSceneDesc
 *
 * Plain C: pointers, sizes and PODs only.  All host memory stays owned by the caller; the library
 * copies what it needs during the call and keeps no host pointer.  Every function returns 0 on
 * success or a negative PtStatus; nothing aborts or throws across this boundary.  A handle is
 * bound to one HIP device and must be driven from one host thread at a time.  There is NO CPU
 * fallback: without a usable gfx950 device ptgpu_create fails with PT_ERR_NO_DEVICE.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PT_OK = 0,
    PT_ERR_INVALID_ARGUMENT = -1,
    PT_ERR_NO_DEVICE = -2,        /* no HIP device / not gfx950 */
    PT_ERR_HIP = -3,              /* a HIP runtime call failed; see ptgpu_last_error */
    PT_ERR_NO_SCENE = -4,         /* render/read before ptgpu_upload_scene */
    PT_ERR_QUEUE_OVERFLOW = -5,   /* a ray queue level exceeded its capacity (glass branching); raise branch_capacity */
    PT_ERR_OUT_OF_MEMORY = -6,    /* device or host memory */
    PT_ERR_INTERNAL = -7          /* an unexpected failure inside the library (caught before it could cross this boundary) */
} PtStatus;

typedef struct { float x, y, z; } PtVec3;

/* src/scene.rs:84-92 `enum Material`.  `color` is Emissive's radiance or Physically's reflectivity. */
enum { PT_MAT_NONE = 0, PT_MAT_EMISSIVE = 1, PT_MAT_MIRROR = 2, PT_MAT_TRANSLUCENT = 3, PT_MAT_PHYSICALLY = 4 };
typedef struct {
    int32_t kind;
    PtVec3 color;
    float roughness;
    float metalness;
    float ior;
} PtMaterial;

/* src/scene.rs:128-133 `Sphere` */
typedef struct { PtVec3 origin; float radius; PtMaterial material; } PtSphere;
/* src/scene.rs:95-102 `Plane`; `normal` is what Plane::new computed (scene.rs:106) */
typedef struct { PtVec3 origin, u, v; PtMaterial material; PtVec3 normal; } PtPlane;

/* src/tracer.rs:43-48 `enum ToneMapping` */
enum { PT_TONE_CLAMP = 0, PT_TONE_REINHARD = 1, PT_TONE_EXPOSURE = 2 };
/* src/tracer.rs:50-66 `Camera` with its private `CameraPlane` flattened in */
typedef struct {
    PtVec3 plane_origin, plane_u, plane_v;
    PtVec3 position;
    float width, height, z_near;
    int32_t tone_mapping;
    float exposure;          /* ToneMapping::Exposure(value) */
    float iso;
} PtCamera;

/* src/scene.rs:71-75 `enum Sky`.  HDRI with pixels == NULL is the "texture not loaded yet" state
 * (tracer.rs:579: white). */
enum { PT_SKY_CONSTANT = 0, PT_SKY_HDRI = 1 };
typedef struct {
    int32_t kind;
    PtVec3 radiance;
    const float *pixels;     /* width*height*3 f32, row-major (src/scene.rs:7-13) */
    uint64_t width, height;
} PtSky;

/* src/scene.rs:151-160 `Scene` + src/tracer.rs:8-13 `ImageSettings`.  Primitive ids used by every id
 * buffer follow find_scene_hit's iteration order (src/scene.rs:266,280): spheres, emissive_spheres,
 * planes, emissive_planes; 0xFFFFFFFF = miss. */
typedef struct {
    uint32_t width, height;
    int32_t sub_pixel_sampling;
    PtCamera camera;
    PtSky sky;
    const PtSphere *spheres;          uint64_t n_spheres;
    const PtPlane *planes;            uint64_t n_planes;
    const PtSphere *emissive_spheres; uint64_t n_emissive_spheres;
    const PtPlane *emissive_planes;   uint64_t n_emissive_planes;
} PtSceneDesc;

enum { PT_ACCEL_AUTO = 0, PT_ACCEL_BRUTE_FORCE = 1, PT_ACCEL_BVH = 2 };

/* Knobs that are constants in the reference (tracer.rs:633 depth 4; common.rs:28 seed) or that only
 * exist because the work is on GPUs.  Zero-initialise and then call ptgpu_default_config. */
typedef struct {
    int32_t device;               /* HIP device ordinal */
    uint32_t seed;                /* counter-RNG key */
    int32_t max_depth;            /* recursion depth passed to trace_radiance (reference: 4); 1..24 */
    int32_t sub_pixel_override;   /* -1: use scene.sub_pixel_sampling; 0/1: force (the reference parser cannot express false) */
    uint32_t shard_index;         /* this handle renders global sample indices s with s % shard_count == shard_index */
    uint32_t shard_count;
    uint32_t samples_per_batch;   /* samples per pixel traced per pipeline pass; 0 = auto */
    uint32_t branch_capacity;     /* queue level l holds min(2^l, branch_capacity) x batch records and an overflow is
                                     PT_ERR_QUEUE_OVERFLOW; 0 = start at 1 and grow: a pass that overflows adds nothing and is
                                     rendered again, with the levels re-allocated twice as large, at the next synchronisation */
    int32_t accel;                /* PT_ACCEL_* */
    int32_t fuse_trace_shade;     /* 0: trace and shade kernels; 1: one fused kernel per level; 2: fused, and each block regroups
                                     its rays by material arm before shading; 3: trace-ahead (a kernel shades its records and traces
                                     their children; leaf children never become records); -1: auto (BVH scenes 0, others 1) */
    int32_t profile_events;       /* 1: bracket kernels with HIP events and report per-kernel time in PtStats */
    uint32_t concurrent_passes;   /* 1..4 pipeline passes in flight (own queues and HIP stream each; one fills the GPU while
                                     another drains its last long rays or sweeps backward); 0 = auto (2; BVH scenes: 1 pass of twice the size) */
    int32_t camera_model;         /* 0: CameraSampler::pinhole_ray (what the reference renders with, tracer.rs:629);
                                     1: CameraSampler::thin_lens_ray (tracer.rs:144-197, unused there) */
    float focus_distance;         /* thin lens: FOCUS_DISTANCE, reference constant 10.0 (tracer.rs:155) */
    float lens_aperture;          /* thin lens: LENS_APERTURE, reference constant 4.0 (tracer.rs:177); circle of confusion = 22/aperture */
} PtConfig;

enum { PT_KERNEL_RAYGEN = 0, PT_KERNEL_TRACE = 1, PT_KERNEL_SHADE = 2, PT_KERNEL_TRACE_SHADE = 3,
       PT_KERNEL_TERMINAL = 4, PT_KERNEL_GATHER = 5, PT_KERNEL_RESOLVE = 6, PT_KERNEL_COUNT = 8 };

typedef struct {
    uint64_t samples;                         /* pixel-samples accumulated by this handle since clear */
    uint64_t rays;                            /* nearest-hit queries (primary + bounce + shadow) since clear */
    uint64_t shadow_rays;                     /* the part of `rays` traced by terminal light samples */
    uint64_t records;                         /* queue records processed over all levels since clear */
    uint64_t terminal_records;                /* the part of `records` on the last level (terminal light samples that need a shadow ray) */
    uint64_t bvh_nodes, bvh_leaf_prims;       /* size of the current BVH (0 for brute force) */
    uint64_t bvh_steps;                       /* 64-byte units of the tree fetched by the BVH trace kernel since clear: 1 per node, 1 per leaf primitive */
    uint64_t kernel_launches[PT_KERNEL_COUNT];/* since clear (profile_events only) */
    double kernel_ms[PT_KERNEL_COUNT];        /* since clear (profile_events only) */
    uint64_t device_bytes;                    /* device memory held by the handle */
    uint64_t queue_growth;                    /* queue level l holds min(2^l, queue_growth) records per sample (1 without glass) */
    uint64_t debug_cycles[8];                 /* per-phase shader cycles of the fused kernel; diagnostic builds only, else 0 */
    uint64_t bvh_depth;                       /* levels of inner nodes in the 4-wide tree (a traversal stack holds at most 3 per level) */
    uint64_t bvh_binary_depth;                /* depth of the binary SAH tree it was folded from (capped at 38) */
    uint64_t bvh_forced_median_splits;        /* splits the builder took at the object median to stay under that cap */
    uint64_t bvh_level_fold;                  /* 1: the 4-wide tree was folded level by level because folding by box area was too deep */
    uint64_t queue_bytes;                     /* the part of device_bytes held by the ray queue levels */
    uint64_t level_records[25];               /* queue records per path-tree level since clear (level 0 = primary rays, level max_depth = terminal light samples) */
    uint64_t samples_per_pass;                /* samples per pixel one pipeline pass takes at most (PtConfig.samples_per_batch, or what auto chose) */
    uint64_t passes_in_flight;                /* pipeline passes that can be in flight at once (BVH scenes: one large one, plus two that only take passes of a few samples) */
} PtStats;

typedef struct PtHandle PtHandle;

void ptgpu_default_config(PtConfig *cfg);

/* Replaces WorkerPool::new + Backbuffer::new (src/window.rs:104-116). */
int ptgpu_create(const PtConfig *cfg, PtHandle **out);
void ptgpu_destroy(PtHandle *h);

/* Called wherever the reference replaces `*scene` and clears the backbuffer (src/window.rs:130-163):
 * flattens the scene into device buffers, (re)builds the BVH, resizes and zeroes the accumulators. */
int ptgpu_upload_scene(PtHandle *h, const PtSceneDesc *scene);

/* Backbuffer::clear (src/tracer.rs:242-256): zero pixels32/pixels8, sample count back to 0. */
int ptgpu_clear(PtHandle *h);

/* Replaces n_samples frames of the tile fan-out + barrier (src/window.rs:166-196): adds n_samples
 * samples per pixel to the accumulators.  Work is enqueued on the handle's stream; ptgpu_sync or any
 * ptgpu_read_* waits for it. */
int ptgpu_render(PtHandle *h, uint32_t n_samples);
int ptgpu_sync(PtHandle *h);

/* One iteration of the window loop - the tile fan-out + wait() of a frame AND the hand-over of Backbuffer.pixels8 to
 * glDrawPixels (src/window.rs:166-213) - with the two pipelined: the call enqueues n_samples more samples per pixel
 * (the reference's frame is 1: the literal of src/window.rs:166), whose last pass also writes the display bytes
 * (assign_pixel8_unsafe, src/tracer.rs:271-282, for the sums and the sample count as they stand after this frame) and
 * which then travel to page-locked host memory on a stream of their own, and returns in *pixels the NEWEST FRAME THAT
 * HAS ARRIVED there - this call's own when the GPU is ahead of the host, else an earlier one, at the oldest the frame
 * two before this call's, which is waited for; *num_samples (may be NULL) = the samples per pixel that image shows.
 * The host never waits for the frame it has just asked for and the GPU always has frames queued: a loop of
 * ptgpu_render(h, 1) + ptgpu_present_pixels8 leaves the GPU idle while the host waits, copies and draws.
 * Every image handed out is exactly Backbuffer.pixels8 after *num_samples samples - the bytes ptgpu_read_pixels8 gives
 * after that many ptgpu_render samples.  The first call after ptgpu_upload_scene / ptgpu_clear waits for its own frame;
 * n_samples == 0 renders nothing and waits for the newest frame (a flush: the last image of a loop).  *pixels stays
 * valid until the next call.  A frame is one pipeline pass when n_samples is at most the pass size
 * (PtConfig.samples_per_batch; 0 = auto). */
int ptgpu_frame(PtHandle *h, uint32_t n_samples, const uint8_t **pixels, uint64_t *num_samples);

/* Tile shards - the other way to split the path over several handles (the first: PtConfig.shard_index / shard_count
 * select sample indices, ptgpu_render).  The image is cut into the reference's 32x32 work tiles (src/window.rs:167-178;
 * clipped to the image), tile (tile_x, tile_y) belongs to handle ptgpu_tile_shard(tile_x, tile_y, tiles_x, shard_count),
 * and this call adds the global sample indices first_sample .. first_sample + n_samples - 1 for the pixels of THIS
 * handle's tiles only (its other pixels keep their sums).  The counter RNG is keyed on (pixel, global sample index), so
 * the sums of all shard_count handles, added up, are bit for bit what one handle renders for the same samples - and no
 * handle needs another's data to show its own tiles: a frame of ONE sample keeps every GPU busy.  Calls of
 * ptgpu_render and ptgpu_render_tiles may be mixed as long as every sample index is rendered once for every pixel. */
enum { PT_TILE = 32 };
int ptgpu_render_tiles(PtHandle *h, uint64_t first_sample, uint32_t n_samples);
/* The split rule itself (pure arithmetic, no device): tile_id modulo n, rows shifted against each other when a row of
 * tiles holds a whole number of rounds. */
uint32_t ptgpu_tile_shard(uint32_t tile_x, uint32_t tile_y, uint32_t tiles_x, uint32_t n);

/* ptgpu_render for callers that cannot allocate.  No counterpart in the reference: it exists for hosts that must not
 * call hipMalloc while they render - code under HIP graph capture, tight serving loops, harnesses that forbid it.
 * Same inputs (the uploaded scene) and outputs (the handle's accumulators, the same sample indices ptgpu_render would
 * use next) and the same sums bit for bit, but the ray queues of its passes live in a fixed array of
 * PTGPU_NOALLOC_SCRATCH_BYTES that is part of the loaded library (each GPU the library has run on holds one until the
 * process ends) instead of in memory the handle allocated, and the call makes no hipMalloc/hipFree and no host
 * allocation.  It returns when its samples are accumulated.
 * Limits: the queues of ONE sample per pixel must fit the array - about 180 000 pixels at max_depth 4 (e.g. 480x375),
 * 45 000 at max_depth 16, for levels of up to 1.25 records per sample (half of that when glass branches more) - else PT_ERR_INVALID_ARGUMENT; scenes traced through a BVH (by default those with
 * more than 100 primitives; PtConfig.accel = PT_ACCEL_BRUTE_FORCE keeps any scene on the linear scan) are refused likewise.  Glass
 * that branches beyond what fits gives PT_ERR_QUEUE_OVERFLOW (the samples that did fit are kept and counted).
 * All handles on one GPU share that GPU's array: calls that overlap on one GPU race; calls one after the other are
 * safe (the call clears the counters it uses). */
#define PTGPU_NOALLOC_SCRATCH_BYTES (64u << 20)
int ptgpu_render_noalloc(PtHandle *h, uint32_t n_samples);

/* Backbuffer.pixels32 (u32 x3 sums, row-major, row 0 = bottom) and the number of samples summed. */
int ptgpu_read_pixels32(PtHandle *h, uint32_t *out, uint64_t *num_samples);
/* Backbuffer.pixels8 exactly as assign_pixel8_unsafe leaves it after the last frame
 * (src/tracer.rs:271-282): min(255, iso/100 * sum / num_samples) truncated; what glDrawPixels shows
 * (src/window.rs:205-210). */
int ptgpu_read_pixels8(PtHandle *h, uint8_t *out);
/* The same bytes in page-locked memory the library owns, for a host that only hands them on - the window loop gives
 * Backbuffer.pixels8 to glDrawPixels every frame (src/window.rs:205-210): no copy into caller memory, and the
 * device-to-host copy goes by DMA (1024x1024: 0.08 instead of 0.20 ms per frame).  *pixels stays valid until the
 * next ptgpu_present_pixels8, ptgpu_upload_scene or ptgpu_destroy on this handle. */
int ptgpu_present_pixels8(PtHandle *h, const uint8_t **pixels);

/* Parity hook: primitive id hit by the primary ray of every pixel for one global sample index. */
int ptgpu_render_primary_ids(PtHandle *h, uint32_t sample_index, uint32_t *out);

/* Multi-GPU combine: the device address of this handle's u32 accumulator (width*height*3 words) so a
 * host can all-reduce it in place (RCCL / torch.distributed), then set the global sample count. */
int ptgpu_device_pixels32(PtHandle *h, void **device_ptr, uint64_t *n_words);
int ptgpu_set_num_samples(PtHandle *h, uint64_t num_samples);
/* The same combine for hosts that own the exchange buffer: copy the accumulator to / from device memory
 * the caller provides (width*height*3 u32 words, on this handle's device). */
int ptgpu_export_pixels32(PtHandle *h, void *dst_device);
int ptgpu_import_pixels32(PtHandle *h, const void *src_device, uint64_t num_samples);
/* The handle's hipStream_t (as void*), for event timing by the caller.  Everything a call of this library enqueues is
 * ordered on this stream by the time the call returns (up to three more internal streams carry the other passes in
 * flight, PtConfig.concurrent_passes, and are joined into it at the end of ptgpu_render); work the caller enqueues here
 * itself is not seen by those streams until ptgpu_sync. */
int ptgpu_stream(PtHandle *h, void **stream);

int ptgpu_get_stats(PtHandle *h, PtStats *out);
/* The acceleration structure ptgpu_upload_scene would build for this scene, built on the host only (no device is
 * touched): fills the bvh_* fields of *out, zeroes the rest.  The reference has no such structure (find_scene_hit,
 * src/scene.rs:263-297, scans every primitive); this is how a host or a test checks its size and depth bounds.  The
 * structure is also checked against itself - every primitive inside the box of every node above it, each reachable
 * exactly once -: PT_ERR_INTERNAL if that fails. */
int ptgpu_bvh_build_stats(const PtSceneDesc *scene, PtStats *out);
/* The same structure walked ON THE HOST (no device is touched) by the traversal functions of the kernels, compiled for
 * the CPU, one ray after the other: rays = n_rays x (origin.xyz, direction.xyz); prims[r] = id of the nearest primitive
 * (0xFFFFFFFF: none - find_scene_hit's None, src/scene.rs:263-297), ts[r] its distance.  Every reference to a unit of the
 * structure and every stack slot is checked on the way: faults[0] = references outside the structure or walks that did
 * not end, [1] = references to a unit of the other kind (node / primitive), distances that are NaN from a primitive
 * inside the structure and hits whose shading row is wrong, [2] = stack slots outside the stack,
 * [3] = units fetched (faults points at four words).  A diagnostic for tests: it answers "which primitive" for any ray, including rays no scene
 * produces, without a GPU to fault. */
int ptgpu_bvh_host_trace(const PtSceneDesc *scene, uint64_t n_rays, const float *rays, uint32_t *prims, float *ts, uint64_t *faults);
/* The same with test switches.  PT_HOST_TRACE_UNGATED: the traversal normally answers rays it cannot walk safely by
 * arithmetic alone (NaN or infinite origins and directions, origins far outside the structure) with the linear scan
 * instead; with this flag EVERY ray is walked through the structure.  prims/ts may then differ from find_scene_hit's
 * (src/scene.rs:263-297) for such rays, but faults[0] and faults[2] must still be 0: the walk follows only references the
 * builder wrote (each node carries its child count), so it cannot leave the structure whatever the arithmetic yields. */
enum { PT_HOST_TRACE_UNGATED = 1 };
int ptgpu_bvh_host_trace_flags(const PtSceneDesc *scene, uint64_t n_rays, const float *rays, uint32_t *prims, float *ts, uint64_t *faults,
                               uint32_t flags);
const char *ptgpu_last_error(PtHandle *h);

/* ---- Several GPUs of one node behind one handle ----------------------------------------------------------------
 * Replaces the same pair as ptgpu_create/ptgpu_render - WorkerPool::new(8, ...) (src/window.rs:109-116) and the
 * per-frame process(tile) x tiles + wait() (src/window.rs:166-196) - for a single-process host with more than one
 * GPU: the devices take the place of the pool's worker threads.  The group owns one PtHandle and one host thread
 * per device; device k renders the global sample indices s with s % n == k for all pixels (scene replicated), and
 * the u32 accumulators are summed onto the first device with RCCL (ncclReduce, ncclUint32/ncclSum, over xGMI) when
 * the image is read - the only exchange.  Integer sums and an RNG keyed on the global sample index make pixels32 /
 * pixels8 bit-identical to what one handle renders for the same total number of samples.
 * cfg->device, shard_index and shard_count are ignored (set per member).  devices == NULL: ordinals 0..n-1;
 * n_devices == 0: every visible device.  Calls on one group come from one host thread at a time. */
typedef struct PtGroup PtGroup;
int ptgpu_group_create(const PtConfig *cfg, const int32_t *devices, uint32_t n_devices, PtGroup **out);
void ptgpu_group_destroy(PtGroup *g);
int ptgpu_group_upload_scene(PtGroup *g, const PtSceneDesc *scene);   /* every `*scene = ...; backbuffer.clear()` site, src/window.rs:130-163 */
int ptgpu_group_clear(PtGroup *g);                                    /* Backbuffer::clear, src/tracer.rs:242-256 */
/* n_samples more samples per pixel IN TOTAL (src/window.rs:166-196 x n_samples), split over the devices by sample index
 * - or, when there are fewer samples than devices (the window loop's one sample per frame), by image tile: every device
 * renders all of them for its own 32x32 tiles (ptgpu_render_tiles), so no device idles.  Returns when the work is
 * enqueued on every device. */
int ptgpu_group_render(PtGroup *g, uint32_t n_samples);
/* ptgpu_frame for the group: one iteration of the window loop (src/window.rs:166-213) on every GPU of the node.  The
 * frame's n_samples are split by image tile, each device's last pass writes the display bytes of its own tiles, one RCCL
 * exchange of bytes (ncclReduce of ncclUint8: disjoint tiles, a quarter of the size of the sums) gathers them on the first
 * device, and the call returns the newest frame that has arrived in page-locked host memory (at the oldest the frame two
 * before this call's, as ptgpu_frame); no device waits for the host, no device waits for another's sums.  The images are exactly Backbuffer.pixels8 after *num_samples samples.  After a
 * ptgpu_group_render that split by sample index (until the next clear) the frames are not pipelined: each call then
 * renders, reduces the sums and returns its own frame. */
int ptgpu_group_frame(PtGroup *g, uint32_t n_samples, const uint8_t **pixels, uint64_t *num_samples);
int ptgpu_group_sync(PtGroup *g);
/* Backbuffer.pixels32 / pixels8 / the bytes for glDrawPixels (src/window.rs:205-210) of everything rendered so far;
 * each call synchronises the devices and runs the one reduction. */
int ptgpu_group_read_pixels32(PtGroup *g, uint32_t *out, uint64_t *num_samples);
int ptgpu_group_read_pixels8(PtGroup *g, uint8_t *out);
int ptgpu_group_present_pixels8(PtGroup *g, const uint8_t **pixels);
int ptgpu_group_get_stats(PtGroup *g, PtStats *out);                  /* counts and times summed over the devices */
int ptgpu_group_exchange_stats(PtGroup *g, uint64_t *bytes, double *ms);  /* size and device time of the last reduction */
/* The split rule itself (pure arithmetic, no device): how many of the global sample indices first .. first+count-1
 * are congruent to k modulo n, i.e. go to device k of n. */
uint64_t ptgpu_group_shard_samples(uint64_t first, uint64_t count, uint32_t k, uint32_t n);
uint32_t ptgpu_group_size(const PtGroup *g);
PtHandle *ptgpu_group_member(PtGroup *g, uint32_t k);                 /* device k's handle (stats, stream); owned by the group */
const char *ptgpu_group_last_error(PtGroup *g);

/* What the Rust host computes with hmath before the scene reaches this library; provided for C/C++/
 * Python hosts.  Camera::new (src/tracer.rs:69-96) and Plane::new's normal (src/scene.rs:106). */
void ptgpu_camera_new(PtVec3 position, PtVec3 target, PtVec3 up, float width, float height, float z_near,
                      int32_t tone_mapping, float exposure, float iso, PtCamera *out);
void ptgpu_plane_normal(PtVec3 u, PtVec3 v, PtVec3 *out);
void ptgpu_plane_normals(PtPlane *planes, uint64_t n);   /* fills planes[i].normal from u and v */

/* Device-side evaluation of one deterministic math function over n arguments (test hook for the
 * sin/cos/pow/exp/acos/atan2 replacements the kernels use): fn 0 sin, 1 cos, 2 pow, 3 exp, 4 acos, 5 atan2; and for the
 * basic operations that compile to instruction sequences: 6 sqrt(a), 7 a / b, 8 random32 of a's bit pattern; 9, 10, 11: the
 * x, y, z component of normalize((a, b, a * 0.37 + b)) as the kernels evaluate it. */
int ptgpu_eval_math(PtHandle *h, int32_t fn, const float *a, const float *b, float *out, uint64_t n);
/* Device self-test of the gamma quantiser's lookup table (the resolve kernel's fast path for
 * Pixel32::from_unit(gamma_correction(x)), src/tracer.rs:609-616 + src/common.rs:78-82): compares it with the direct
 * powf evaluation for EVERY float in [0,1] plus ranges around the edges; *mismatches must come back 0. */
int ptgpu_selftest_gamma(PtHandle *h, uint64_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif
