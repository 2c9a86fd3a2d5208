/*
 * ptref.h -- CPU oracle for the hpatjens/Pathtracer hot path (C ABI, plain C99).
 *
 * TEST INFRASTRUCTURE ONLY.  This is a restatement, function by function, of the reference's
 * src/common.rs, src/scene.rs and src/tracer.rs (plus the tile fan-out of src/worker.rs +
 * src/window.rs:166-196) used to CHECK the GPU library.  Nothing under pathtracer_amd/ may include,
 * link or call it; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * PARITY: PINNED AT IMAGE LEVEL ONLY, UNPINNED AT VALUE LEVEL.  The reference ships no tests, golden vectors
 * or fixtures, its vector math lives in the un-vendored `hmath` crate (Cargo.toml:14, git
 * 82996f82319240c521a81dc1b60228281ab3554d) and it cannot be compiled here (no rustc).  Its only output on
 * record is docs/5.jpg, its own 1024x1024 render of scenes/sample/sample.scene: this oracle (xorshift32 stream,
 * platform libm) converges to that image - RMSE 4.9/255 at 2048 spp, per-channel bias 0.2, median per-pixel error
 * 2.0, the residual on one-pixel outlines of the light (tests/test_reference_image.py, fixture
 * tests/golden/reference_render_256.npz).  No individual value (a hit distance, a pixel sum) is pinned by the
 * reference; those are pinned only by known-answer values derived by hand from the source
 * (tests/test_oracle_kat.py).  Vector-op definitions assumed for hmath: dot = (x*x' + y*y') + z*z',
 * length = sqrt(dot(v,v)), normalize = v / length (3 divides), right-handed cross, component-wise * and /.
 */
#ifndef PTREF_H
#define PTREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float x, y, z; } pr_vec3;

/* scene.rs:84-92 `enum Material` */
enum { PR_MAT_NONE = 0, PR_MAT_EMISSIVE = 1, PR_MAT_MIRROR = 2, PR_MAT_TRANSLUCENT = 3, PR_MAT_PHYSICALLY = 4 };
typedef struct {
    int32_t kind;
    pr_vec3 color;      /* Emissive: radiance; Physically: reflectivity */
    float roughness;    /* Physically */
    float metalness;    /* Physically */
    float ior;          /* Translucent */
} pr_material;

/* scene.rs:128-133 */
typedef struct { pr_vec3 origin; float radius; pr_material material; } pr_sphere;
/* scene.rs:95-102 */
typedef struct { pr_vec3 origin, u, v; pr_material material; pr_vec3 normal; } pr_plane;

/* tracer.rs:43-48 */
enum { PR_TONE_CLAMP = 0, PR_TONE_REINHARD = 1, PR_TONE_EXPOSURE = 2 };
/* tracer.rs:50-66 `CameraPlane` + `Camera` */
typedef struct {
    pr_vec3 plane_origin, plane_u, plane_v;
    pr_vec3 position;
    float width, height, z_near;
    int32_t tone_mapping;
    float exposure;
    float iso;
} pr_camera;

/* scene.rs:71-75 `enum Sky` */
enum { PR_SKY_CONSTANT = 0, PR_SKY_HDRI = 1 };
typedef struct {
    int32_t kind;
    pr_vec3 radiance;       /* Constant */
    const float *pixels;    /* HDRI: width*height*3 f32, NULL = "not loaded yet" (tracer.rs:579) */
    uint64_t width, height;
} pr_sky;

/* scene.rs:151-160 `Scene`, tracer.rs:8-13 `ImageSettings` */
typedef struct {
    uint32_t width, height;
    int32_t sub_pixel_sampling;
    pr_camera camera;
    pr_sky sky;
    const pr_sphere *spheres;          uint64_t n_spheres;
    const pr_plane *planes;            uint64_t n_planes;
    const pr_sphere *emissive_spheres; uint64_t n_emissive_spheres;
    const pr_plane *emissive_planes;   uint64_t n_emissive_planes;
} pr_scene;

enum { PR_RNG_XORSHIFT = 0,  /* the reference's global xorshift32 stream, serial draw order */
       PR_RNG_COUNTER = 1 }; /* counter RNG keyed on (seed, pixel, sample, path node, dimension) */
enum { PR_MATH_LIBM = 0,     /* sinf/cosf/powf/expf/acosf/atan2f from the platform libm, as Rust does */
       PR_MATH_DET = 1 };    /* ptref_detmath.h */

typedef struct {
    int32_t rng_mode;
    int32_t math_mode;
    uint32_t seed;            /* PR_RNG_COUNTER key / PR_RNG_XORSHIFT initial state (reference: 314159265) */
    int32_t max_depth;        /* reference: 4 (tracer.rs:633) */
    int32_t sub_pixel_override; /* -1: use scene.sub_pixel_sampling, 0/1: force */
    int32_t camera_model;     /* 0: pinhole_ray (what render() calls, tracer.rs:629); 1: thin_lens_ray (tracer.rs:144-197) */
    float focus_distance;     /* thin lens; reference constant 10.0 (tracer.rs:155) */
    float lens_aperture;      /* thin lens; reference constant 4.0 (tracer.rs:177) */
    int32_t hoist_plane_axes; /* timing variant only: 1 = normalize(u), length(u), normalize(v), length(v) of every plane
                                 once per pass instead of once per ray and plane (scene.rs:244-253); same image */
} pr_options;

typedef struct {
    uint64_t rays;            /* find_scene_hit calls (primary + bounce + shadow) */
    uint64_t prim_tests;      /* intersect_sphere + intersect_plane calls */
} pr_stats;

/* tracer.rs:200-216 `Backbuffer` */
typedef struct {
    uint32_t width, height;
    uint64_t num_samples;     /* starts at 1 (tracer.rs:227), bumped by the caller after each pass */
    uint32_t *pixels32;       /* width*height*3 */
    uint8_t *pixels8;         /* width*height*3 */
} pr_backbuffer;

typedef struct {
    float t;
    pr_vec3 position, normal;
    int32_t transition;       /* 0 = In, 1 = Out (tracer.rs:22-23) */
    uint32_t prim_id;         /* spheres ++ emissive_spheres ++ planes ++ emissive_planes; 0xFFFFFFFF = miss */
} pr_hit;

/* --- leaf utilities (common.rs) --- */
uint32_t ptref_xorshift32_next(uint32_t *state);                       /* common.rs:28-36 */
float ptref_random32_from_u32(uint32_t r);                             /* common.rs:38-41 */
uint32_t ptref_counter_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t dim);
float ptref_math_eval(int32_t math_mode, int32_t fn, float a, float b); /* fn: 0 sin 1 cos 2 pow 3 exp 4 acos 5 atan2 6 sqrt 7 a/b 8 random32(bits of a) */
void ptref_math_eval_array(int32_t math_mode, int32_t fn, const float *a, const float *b, float *out, uint64_t n);
float ptref_sqrtf_ieee(float x);

/* --- constructors the Rust host runs (they use hmath) --- */
void ptref_camera_new(pr_vec3 position, pr_vec3 target, pr_vec3 up, float width, float height, float z_near,
                      int32_t tone_mapping, float exposure, float iso, pr_camera *out);       /* tracer.rs:69-96 */
void ptref_plane_normal(pr_vec3 u, pr_vec3 v, pr_vec3 *out);                                  /* scene.rs:106 */
void ptref_plane_normals(pr_plane *planes, uint64_t n);

/* --- the path --- */
int32_t ptref_find_scene_hit(const pr_scene *scene, pr_vec3 origin, pr_vec3 direction, pr_hit *out); /* scene.rs:263-297 */
void ptref_pinhole_ray(const pr_scene *scene, const pr_options *opt, uint32_t x, uint32_t y, float jitter_u,
                       float jitter_v, int32_t jitter, pr_vec3 *origin, pr_vec3 *direction);   /* tracer.rs:125-142 */
void ptref_thin_lens_ray(const pr_scene *scene, const pr_options *opt, float r1, float r2, pr_vec3 *origin,
                         pr_vec3 *direction);                                                   /* tracer.rs:144-197 */
/* radiance of one ray, recursion as tracer.rs:448-584; RNG state as given by opt (xorshift state = *state) */
pr_vec3 ptref_trace_radiance(const pr_scene *scene, const pr_options *opt, uint32_t *xorshift_state,
                             uint32_t pixel, uint32_t sample, pr_vec3 origin, pr_vec3 direction, int32_t depth,
                             pr_stats *stats);
pr_vec3 ptref_tone_map_gamma(const pr_scene *scene, const pr_options *opt, pr_vec3 hdr);       /* tracer.rs:586-616,634-640 */
void ptref_pixel32_from_unit(pr_vec3 c, uint32_t out[3]);                                      /* common.rs:78-82 */

/* One full-image pass (1 spp): 32x32 tiles in row-major order, clipped to the image, pixels y-then-x
 * inside a tile (window.rs:166-179, tracer.rs:625-646), single thread.  `sample_index` keys the counter
 * RNG; the xorshift state is threaded through *xorshift_state.  Uses bb->num_samples for pixels8 and does
 * NOT bump it (the caller does, as window.rs:211). */
void ptref_render_pass(const pr_scene *scene, const pr_options *opt, pr_backbuffer *bb, uint32_t *xorshift_state,
                       uint32_t sample_index, pr_stats *stats);

/* Same pass on a pool of `n_threads` worker threads pulling 32x32 tiles from a FIFO, with a barrier at the end
 * (worker.rs:18-130).  PR_RNG_XORSHIFT uses one private stream per thread seeded seed+thread_id (the
 * reference's shared static is a data race); PR_RNG_COUNTER gives the same pixels as the serial pass. */
void ptref_render_pass_pool(const pr_scene *scene, const pr_options *opt, pr_backbuffer *bb, int32_t n_threads,
                            uint32_t sample_index, pr_stats *stats);

/* Primary-hit primitive ids for one sample index (row-major y*width+x), 0xFFFFFFFF = miss. */
void ptref_primary_ids(const pr_scene *scene, const pr_options *opt, uint32_t sample_index, uint32_t *ids);

/* Test hook: log every find_scene_hit call made from now on into rows[capacity][8] (origin.xyz, direction.xyz, bits of
 * the primitive id found - 0xFFFFFFFF none -, t); NULL stops.  ptref_ray_log_count: calls since the log was set. */
void ptref_ray_log(float *rows, uint64_t capacity);
uint64_t ptref_ray_log_count(void);

/* pixels8 from pixels32 (tracer.rs:271-282) for every pixel with the given num_samples. */
void ptref_assign_pixels8(const pr_scene *scene, const uint32_t *pixels32, uint64_t num_samples, uint8_t *pixels8);

#ifdef __cplusplus
}
#endif
#endif
