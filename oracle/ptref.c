/*
 * ptref.c -- CPU oracle: restatement of the reference's ray/scene intersection and shading loop.
 * TEST INFRASTRUCTURE ONLY; see ptref.h.  Every function cites the reference lines it follows
 * (paths relative to the reference checkout: src/common.rs, src/scene.rs, src/tracer.rs,
 * src/worker.rs, src/window.rs).
 *
 * Build: gcc -O2 -std=c99 -ffp-contract=off -fno-fast-math  (no FMA contraction, IEEE f32 on SSE).
 */
#define _POSIX_C_SOURCE 200809L
#include "ptref.h"
#include "ptref_detmath.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------
 * hmath stand-ins (Vector3<f32>).  Operation order is this build's definition (ptref.h header).
 * ---------------------------------------------------------------------------------------- */
static inline pr_vec3 v3(float x, float y, float z) { pr_vec3 r = {x, y, z}; return r; }
static inline pr_vec3 vadd(pr_vec3 a, pr_vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline pr_vec3 vsub(pr_vec3 a, pr_vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline pr_vec3 vneg(pr_vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline pr_vec3 vmul(pr_vec3 a, pr_vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline pr_vec3 vdiv(pr_vec3 a, pr_vec3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline pr_vec3 vscale(pr_vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }   /* Vec*f32 and f32*Vec */
static inline pr_vec3 vdivs(pr_vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }    /* Vec/f32 */
static inline float vdot(pr_vec3 a, pr_vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline pr_vec3 vcross(pr_vec3 a, pr_vec3 b) {
    return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(pr_vec3 a) { return sqrtf(vdot(a, a)); }
static inline pr_vec3 vnormalize(pr_vec3 a) { return vdivs(a, vlength(a)); }

float ptref_sqrtf_ieee(float x) { return sqrtf(x); }

/* f32::max / f32::min: return the non-NaN operand (Rust std); for equal operands the second. */
static inline float fmax_rs(float a, float b) { if (b != b) return a; if (a != a) return b; return a > b ? a : b; }
static inline float fmin_rs(float a, float b) { if (b != b) return a; if (a != a) return b; return a < b ? a : b; }

/* `x as u32` for f32 (saturating since Rust 1.45): NaN -> 0, negative -> 0, >= 2^32 -> u32::MAX. */
static inline uint32_t f32_as_u32(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
static inline uint8_t f32_as_u8(float x) {
    if (!(x > 0.0f)) return 0;
    if (x >= 255.0f) return 255;
    return (uint8_t)x;
}

/* ------------------------------------------------------------------------------------------
 * common.rs
 * ---------------------------------------------------------------------------------------- */

/* common.rs:28-36 */
uint32_t ptref_xorshift32_next(uint32_t *state) {
    uint32_t x = *state;
    x ^= x << 13;
    x ^= x >> 17;
    x ^= x << 5;
    *state = x;
    return x;
}

/* common.rs:38-41: `r as f32 / std::u32::MAX as f32`; u32::MAX as f32 rounds to 2^32. */
float ptref_random32_from_u32(uint32_t r) { return (float)r / 4294967296.0f; }

static inline uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

/* Counter RNG (no reference counterpart; replaces the racy global stream, common.rs:28).  Output is
 * never 0, like xorshift32, so random32 stays in (0, 1]. */
uint32_t ptref_counter_u32(uint32_t seed, uint32_t pixel, uint32_t sample, uint32_t node, uint32_t dim) {
    uint32_t h = mix32(seed + 0x9E3779B9U);
    h = mix32((h ^ pixel) + 0x85EBCA6BU);
    h = mix32((h ^ sample) + 0xC2B2AE35U);
    h = mix32((h ^ node) + 0x27D4EB2FU);
    h = mix32((h ^ dim) + 0x165667B1U);
    return h ? h : 1u;
}

/* common.rs:44-57 */
static inline float clampf32(float mn, float mx, float x) { if (x > mx) return mx; else if (x < mn) return mn; else return x; }
static inline float saturatef32(float x) { return clampf32(0.0f, 1.0f, x); }
/* common.rs:68-70 */
static inline pr_vec3 mix_vec3(pr_vec3 a, pr_vec3 b, float x) { return vadd(vscale(a, 1.0f - x), vscale(b, x)); }

/* common.rs:78-82 */
void ptref_pixel32_from_unit(pr_vec3 c, uint32_t out[3]) {
    out[0] = f32_as_u32(c.x * 255.0f);
    out[1] = f32_as_u32(c.y * 255.0f);
    out[2] = f32_as_u32(c.z * 255.0f);
}

/* ------------------------------------------------------------------------------------------
 * math dispatch
 * ---------------------------------------------------------------------------------------- */
static inline float m_sin(int mode, float x) { return mode == PR_MATH_DET ? ptd_sinf(x) : sinf(x); }
static inline float m_cos(int mode, float x) { return mode == PR_MATH_DET ? ptd_cosf(x) : cosf(x); }
static inline float m_pow(int mode, float x, float y) { return mode == PR_MATH_DET ? ptd_powf(x, y) : powf(x, y); }
static inline float m_exp(int mode, float x) { return mode == PR_MATH_DET ? ptd_expf(x) : expf(x); }
static inline float m_acos(int mode, float x) { return mode == PR_MATH_DET ? ptd_acosf(x) : acosf(x); }
static inline float m_atan2(int mode, float y, float x) { return mode == PR_MATH_DET ? ptd_atan2f(y, x) : atan2f(y, x); }

float ptref_math_eval(int32_t mode, int32_t fn, float a, float b) {
    switch (fn) {
    case 0: return m_sin(mode, a);
    case 1: return m_cos(mode, a);
    case 2: return m_pow(mode, a, b);
    case 3: return m_exp(mode, a);
    case 4: return m_acos(mode, a);
    case 5: return m_atan2(mode, a, b);
    case 6: return sqrtf(a);            /* IEEE: the host's is correctly rounded */
    case 7: return a / b;
    case 8: { uint32_t u; memcpy(&u, &a, 4); return ptref_random32_from_u32(u); }
    default: return 0.0f;
    }
}

void ptref_math_eval_array(int32_t mode, int32_t fn, const float *a, const float *b, float *out, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) out[i] = ptref_math_eval(mode, fn, a[i], b ? b[i] : 0.0f);
}

#define PR_PI 3.14159274101257324219f /* std::f32::consts::PI (common.rs:5) */

/* ------------------------------------------------------------------------------------------
 * tracer.rs: camera
 * ---------------------------------------------------------------------------------------- */

/* tracer.rs:69-96 Camera::new + construct_projection_plane */
void ptref_camera_new(pr_vec3 position, pr_vec3 target, pr_vec3 up, float width, float height, float z_near,
                      int32_t tone_mapping, float exposure, float iso, pr_camera *out) {
    pr_vec3 z = vnormalize(vsub(position, target));
    pr_vec3 x = vnormalize(vcross(up, z));
    pr_vec3 y = vnormalize(vcross(z, x));
    float half_width = width / 2.0f;
    float half_height = height / 2.0f;
    pr_vec3 to_plane_center = vscale(z, -z_near);
    /* position + to_plane_center - x*half_width - y*half_height, left to right */
    pr_vec3 origin = vsub(vsub(vadd(position, to_plane_center), vscale(x, half_width)), vscale(y, half_height));
    out->plane_origin = origin;
    out->plane_u = vscale(x, width);
    out->plane_v = vscale(y, height);
    out->position = position;
    out->width = width; out->height = height; out->z_near = z_near;
    out->tone_mapping = tone_mapping; out->exposure = exposure; out->iso = iso;
}

/* scene.rs:106 Plane::new: normal = u.cross(v).normalize() */
void ptref_plane_normal(pr_vec3 u, pr_vec3 v, pr_vec3 *out) { *out = vnormalize(vcross(u, v)); }
void ptref_plane_normals(pr_plane *planes, uint64_t n) {
    for (uint64_t i = 0; i < n; ++i) ptref_plane_normal(planes[i].u, planes[i].v, &planes[i].normal);
}

/* tracer.rs:104-111 (sampler step vectors) + 125-142 (pinhole_ray) */
void ptref_pinhole_ray(const pr_scene *scene, const pr_options *opt, uint32_t x, uint32_t y, float jitter_u,
                       float jitter_v, int32_t jitter, pr_vec3 *origin, pr_vec3 *direction) {
    (void)opt;
    const pr_camera *cam = &scene->camera;
    pr_vec3 su = vdivs(cam->plane_u, (float)scene->width);
    pr_vec3 sv = vdivs(cam->plane_v, (float)scene->height);
    float offset_u = (float)x;
    float offset_v = (float)y;
    if (jitter) { offset_u += jitter_u; offset_v += jitter_v; }
    pr_vec3 du = vscale(su, offset_u);
    pr_vec3 dv = vscale(sv, offset_v);
    pr_vec3 o = vadd(vadd(cam->plane_origin, du), dv);
    *origin = o;
    *direction = vnormalize(vsub(o, cam->position));
}

/* tracer.rs:144-197 thin_lens_ray, given the pinhole ray (in/out) and the two random numbers it draws */
void ptref_thin_lens_ray(const pr_scene *scene, const pr_options *opt, float r1, float r2, pr_vec3 *origin, pr_vec3 *direction) {
    const pr_camera *cam = &scene->camera;
    pr_vec3 su = vdivs(cam->plane_u, (float)scene->width);   /* CameraSampler.u / .v, tracer.rs:107-108 */
    pr_vec3 sv = vdivs(cam->plane_v, (float)scene->height);
    pr_vec3 point_on_focus_plane = vadd(*origin, vscale(*direction, opt->focus_distance));  /* :160 */
    float circle_of_confusion = 22.0f / opt->lens_aperture;                                 /* :181 */
    float alpha = 2.0f * PR_PI * r1;                                                        /* :186 */
    float r = circle_of_confusion * r2;                                                     /* :187 */
    float l_u = r * m_sin(opt->math_mode, alpha);                                           /* :189 */
    float l_v = r * m_cos(opt->math_mode, alpha);                                           /* :190 */
    pr_vec3 lens_origin = vadd(vadd(*origin, vscale(su, l_u)), vscale(sv, l_v));            /* :192 */
    *direction = vnormalize(vsub(point_on_focus_plane, lens_origin));                       /* :193 */
    *origin = lens_origin;
}

/* ------------------------------------------------------------------------------------------
 * scene.rs: intersection
 * ---------------------------------------------------------------------------------------- */

/* scene.rs:169-221 */
static int intersect_sphere(const pr_sphere *sphere, pr_vec3 p, pr_vec3 d, pr_hit *hit) {
    pr_vec3 s = sphere->origin;
    float r = sphere->radius;
    pr_vec3 c = vsub(s, p);
    float e_len = vdot(d, c);
    if (e_len < -r) return 0;
    pr_vec3 e = vscale(d, e_len);
    pr_vec3 v = vsub(e, c);
    float v_len = vlength(v);
    if (v_len > r) return 0;
    float b = sqrtf(r * r - v_len * v_len);
    float t1 = e_len - b;
    float t2 = e_len + b;
    if (t1 >= 0.0f) {
        pr_vec3 position = vadd(p, vscale(d, t1));
        hit->t = t1; hit->position = position;
        hit->normal = vnormalize(vsub(position, sphere->origin));
        hit->transition = 0;
        return 1;
    } else if (t2 >= 0.0f) {
        pr_vec3 position = vadd(p, vscale(d, t2));
        hit->t = t2; hit->position = position;
        hit->normal = vnormalize(vsub(sphere->origin, position));
        hit->transition = 1;
        return 1;
    }
    return 0;
}

/* The four per-plane values scene.rs:244-253 recomputes for every ray: normalize(u), length(u), normalize(v),
 * length(v).  The timing variant `hoist_plane_axes` evaluates them once per pass instead (same operations, same
 * values, so the image does not change); the faithful path passes NULL and recomputes them as the reference does. */
typedef struct { pr_vec3 u_hat, v_hat; float u_len, v_len; } pr_plane_axes;

/* scene.rs:223-261 */
static int intersect_plane(const pr_plane *plane, const pr_plane_axes *ax, pr_vec3 p, pr_vec3 d, pr_hit *hit) {
    pr_vec3 n = plane->normal;
    pr_vec3 s = plane->origin;
    float num = n.x * (s.x - p.x) + n.y * (s.y - p.y) + n.z * (s.z - p.z);
    float denum = n.x * d.x + n.y * d.y + n.z * d.z;
    if (fabsf(denum) < 0.0000001f) return 0;
    float parameter = num / denum;
    if (parameter < 0.0f) return 0;
    pr_vec3 position = vadd(p, vscale(d, parameter));
    pr_vec3 in_plane = vsub(position, s);
    float on_u = vdot(in_plane, ax ? ax->u_hat : vnormalize(plane->u));
    if (on_u > (ax ? ax->u_len : vlength(plane->u)) || on_u < 0.0f) return 0;
    float on_v = vdot(in_plane, ax ? ax->v_hat : vnormalize(plane->v));
    if (on_v > (ax ? ax->v_len : vlength(plane->v)) || on_v < 0.0f) return 0;
    hit->t = parameter; hit->position = position;
    if (vdot(d, n) < 0.0f) { hit->transition = 0; hit->normal = n; }
    else { hit->transition = 1; hit->normal = vneg(n); }
    return 1;
}

/* Test hook: while a log is set, every find_scene_hit call appends its ray and its answer (8 words: origin.xyz,
 * direction.xyz, bits of the primitive id, t) - the exact rays a render traces, for tests that replay them elsewhere.
 * Rows beyond the capacity are counted but not stored.  Calls from the pool's threads interleave in any order. */
static float *g_ray_log = NULL;
static uint64_t g_ray_log_cap = 0, g_ray_log_n = 0;
void ptref_ray_log(float *rows, uint64_t capacity) { g_ray_log = rows; g_ray_log_cap = rows ? capacity : 0; g_ray_log_n = 0; }
uint64_t ptref_ray_log_count(void) { return g_ray_log_n; }
static void log_ray(pr_vec3 p, pr_vec3 d, uint32_t prim_id, float t) {
    uint64_t k = __atomic_fetch_add(&g_ray_log_n, 1, __ATOMIC_RELAXED);
    if (k >= g_ray_log_cap) return;
    float *r = g_ray_log + 8 * k;
    r[0] = p.x; r[1] = p.y; r[2] = p.z; r[3] = d.x; r[4] = d.y; r[5] = d.z;
    memcpy(&r[6], &prim_id, 4);
    r[7] = t;
}

/* scene.rs:263-297.  Iteration (and id) order: spheres, emissive_spheres, planes, emissive_planes;
 * a later hit replaces the nearest only when strictly closer; planes count only with Transition::In. */
static int find_scene_hit(const pr_scene *scene, const pr_plane_axes *axes, pr_vec3 p, pr_vec3 d, pr_hit *out,
                          pr_stats *stats) {
    int found = 0;
    pr_hit best, h;
    uint32_t id = 0;
    memset(&best, 0, sizeof best);
    if (stats) stats->rays++;
    for (int list = 0; list < 2; ++list) {
        const pr_sphere *arr = list == 0 ? scene->spheres : scene->emissive_spheres;
        uint64_t n = list == 0 ? scene->n_spheres : scene->n_emissive_spheres;
        for (uint64_t i = 0; i < n; ++i, ++id) {
            if (stats) stats->prim_tests++;
            if (intersect_sphere(&arr[i], p, d, &h)) {
                if (!found || h.t < best.t) { best = h; best.prim_id = id; found = 1; }
            }
        }
    }
    for (int list = 0; list < 2; ++list) {
        const pr_plane *arr = list == 0 ? scene->planes : scene->emissive_planes;
        uint64_t n = list == 0 ? scene->n_planes : scene->n_emissive_planes;
        for (uint64_t i = 0; i < n; ++i, ++id) {
            if (stats) stats->prim_tests++;
            const pr_plane_axes *ax = axes ? &axes[(list == 0 ? 0 : scene->n_planes) + i] : NULL;
            if (intersect_plane(&arr[i], ax, p, d, &h)) {
                if (h.transition == 0) {
                    if (!found || h.t < best.t) { best = h; best.prim_id = id; found = 1; }
                }
            }
        }
    }
    if (found) *out = best;
    else out->prim_id = 0xFFFFFFFFu;
    if (g_ray_log) log_ray(p, d, out->prim_id, found ? best.t : 0.0f);
    return found;
}

int32_t ptref_find_scene_hit(const pr_scene *scene, pr_vec3 origin, pr_vec3 direction, pr_hit *out) {
    return find_scene_hit(scene, NULL, origin, direction, out, NULL);
}

static const pr_material *prim_material(const pr_scene *sc, uint32_t id) {
    if (id < sc->n_spheres) return &sc->spheres[id].material;
    id -= (uint32_t)sc->n_spheres;
    if (id < sc->n_emissive_spheres) return &sc->emissive_spheres[id].material;
    id -= (uint32_t)sc->n_emissive_spheres;
    if (id < sc->n_planes) return &sc->planes[id].material;
    id -= (uint32_t)sc->n_planes;
    return &sc->emissive_planes[id].material;
}

/* ------------------------------------------------------------------------------------------
 * RNG plumbing
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const pr_scene *scene;
    const pr_options *opt;
    uint32_t *xs;          /* xorshift state (PR_RNG_XORSHIFT) */
    uint32_t pixel, sample;
    pr_stats *stats;
    int math;
    const pr_plane_axes *axes;  /* NULL: recompute per ray as the reference does */
} pr_ctx;

static inline uint32_t draw_u32(pr_ctx *c, uint32_t node, uint32_t dim) {
    if (c->opt->rng_mode == PR_RNG_COUNTER) return ptref_counter_u32(c->opt->seed, c->pixel, c->sample, node, dim);
    return ptref_xorshift32_next(c->xs);
}
static inline float draw_f32(pr_ctx *c, uint32_t node, uint32_t dim) { return ptref_random32_from_u32(draw_u32(c, node, dim)); }

/* ------------------------------------------------------------------------------------------
 * scene.rs: sky texture and light sampling
 * ---------------------------------------------------------------------------------------- */

/* scene.rs:32-38 */
static inline float repeat_f32(float x) {
    /* `x as i32` saturates; |x| here is < 2 so a plain cast is the same */
    float t = (float)(int32_t)x;
    if (x < 0.0f) return x - t + 1.0f;
    return x - t;
}

/* scene.rs:45-68 */
static pr_vec3 sample_equirectangular(const pr_sky *sky, pr_vec3 direction, int math) {
    float u = m_atan2(math, direction.z, direction.x) / (2.0f * PR_PI);
    float v = m_acos(math, direction.y) / PR_PI;
    u = repeat_f32(u);
    v = repeat_f32(v);
    float fx = u * (float)sky->width, fy = v * (float)sky->height;
    /* `as usize` saturating: NaN/negative -> 0 */
    uint64_t px = (fx > 0.0f) ? (uint64_t)fx : 0u;
    uint64_t py = (fy > 0.0f) ? (uint64_t)fy : 0u;
    uint64_t last = sky->width * sky->height * 3u - 3u;
    uint64_t index = 3u * (py * sky->width + px);
    if (index > last) index = last;
    return v3(sky->pixels[index + 0], sky->pixels[index + 1], sky->pixels[index + 2]);
}

/* scene.rs:116-125 Plane::sample; the facing test happens BEFORE any draw. */
static int plane_sample(pr_ctx *c, const pr_plane *pl, pr_vec3 position, uint32_t node, pr_vec3 *out) {
    if (vdot(vsub(pl->origin, position), pl->normal) < 0.0f) return 0;
    float r1 = draw_f32(c, node, 1);
    float r2 = draw_f32(c, node, 2);
    *out = vadd(vadd(pl->origin, vscale(pl->u, r1)), vscale(pl->v, r2));
    return 1;
}

/* scene.rs:136-148 Sphere::sample: a point on the UNIT sphere at the world origin (origin/radius ignored). */
static int sphere_sample(pr_ctx *c, uint32_t node, pr_vec3 *out) {
    float r1 = draw_f32(c, node, 1);
    float r2 = draw_f32(c, node, 2);
    float theta = 2.0f * PR_PI * r1;
    float phi = m_acos(c->math, 2.0f * r2 - 1.0f);
    *out = v3(m_sin(c->math, theta) * m_cos(c->math, phi), m_cos(c->math, theta), m_sin(c->math, theta) * m_sin(c->math, phi));
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * tracer.rs: optics, bases, BRDFs, samplers
 * ---------------------------------------------------------------------------------------- */

/* tracer.rs:285-288 */
static inline pr_vec3 reflect(pr_vec3 incoming, pr_vec3 normal) {
    return vsub(incoming, vscale(normal, 2.0f * vdot(incoming, normal)));
}

/* tracer.rs:290-318 */
static inline pr_vec3 refract(pr_vec3 incoming, pr_vec3 normal, float n1, float n2) {
    pr_vec3 v = vneg(incoming);
    float cos_theta1 = vdot(v, normal);
    float sin_theta1 = sqrtf(1.0f - cos_theta1 * cos_theta1);
    float sin_theta2 = (sin_theta1 * n1) / n2;
    float cos_theta2 = sqrtf(1.0f - sin_theta2 * sin_theta2);
    pr_vec3 p = vadd(incoming, vscale(normal, cos_theta1));
    return vadd(vscale(vneg(normal), cos_theta2), vscale(vnormalize(p), sin_theta2));
}

typedef struct { pr_vec3 x, y, z; } pr_basis;

/* tracer.rs:320-331 */
static inline pr_basis construct_coordinate_system(pr_vec3 normal) {
    const float EPS = 0.9999f;
    pr_vec3 other = (normal.y > EPS || normal.y < -EPS) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f);
    pr_basis b;
    b.y = normal;
    b.x = vcross(b.y, other);
    b.z = vcross(b.y, b.x);
    return b;
}

/* tracer.rs:443-446 */
static inline pr_vec3 to_basis(pr_basis b, pr_vec3 v) {
    return vadd(vadd(vscale(b.x, v.x), vscale(b.y, v.y)), vscale(b.z, v.z));
}

/* tracer.rs:370-389 */
static inline float geometry_schlick_ggx(pr_vec3 n, pr_vec3 a, float k) {
    float n_dot_a = fmax_rs(0.0f, vdot(n, a));
    float nom = n_dot_a;
    float denom = n_dot_a * (1.0f - k) + k;
    return nom / denom;
}
static inline float geometry_smith(pr_vec3 n, pr_vec3 v, pr_vec3 l, float k) {
    return geometry_schlick_ggx(n, l, k) * geometry_schlick_ggx(n, v, k);
}

/* tracer.rs:391-393 */
static inline pr_vec3 fresnel_schlick(int math, float cos_theta, pr_vec3 f0) {
    return vadd(f0, vscale(vsub(v3(1.0f, 1.0f, 1.0f), f0), m_pow(math, 1.0f - cos_theta, 5.0f)));
}

/* tracer.rs:395-415 */
static inline pr_vec3 brdf_cook_torrance(int math, pr_vec3 v, pr_vec3 l, pr_vec3 n, float roughness, pr_vec3 f0) {
    float alpha = roughness * roughness;
    pr_vec3 h = vnormalize(vadd(v, l));
    float cos_theta = fmax_rs(0.0f, vdot(h, v));
    float k = alpha / 2.0f;
    pr_vec3 num = vscale(fresnel_schlick(math, cos_theta, f0), geometry_smith(n, v, l, k));
    float denum = 4.0f * fmax_rs(0.0f, vdot(n, v));
    return vdivs(num, denum);
}

/* tracer.rs:419-430 */
static inline pr_vec3 importance_sample_ggx(int math, float xi_x, float xi_y, float roughness) {
    float a = roughness * roughness;
    float phi = 2.0f * PR_PI * xi_x;
    float sin_phi = m_sin(math, phi);
    float cos_phi = m_cos(math, phi);
    float cos_theta = sqrtf((1.0f - xi_y) / (xi_y * (a * a - 1.0f) + 1.0f));
    float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
    return v3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
}

/* tracer.rs:433-441 */
static inline pr_vec3 importance_sample_cos(int math, float xi_x, float xi_y) {
    float r = sqrtf(xi_x);
    float theta = 2.0f * PR_PI * xi_y;
    float x = r * m_cos(math, theta);
    float z = r * m_sin(math, theta);
    return v3(x, sqrtf(fmax_rs(0.0f, 1.0f - xi_x)), z);
}

/* ------------------------------------------------------------------------------------------
 * tracer.rs:448-584 trace_radiance.  `node` is the heap index of this call in the path tree (root 1,
 * single/refraction child 2n, glass reflection child 2n+1); it only keys the counter RNG.
 * ---------------------------------------------------------------------------------------- */
static pr_vec3 trace_radiance(pr_ctx *c, pr_vec3 origin, pr_vec3 direction, int depth, uint32_t node) {
    const pr_scene *scene = c->scene;
    const pr_vec3 zero = {0.0f, 0.0f, 0.0f}, one = {1.0f, 1.0f, 1.0f};

    if (depth <= 0) { /* tracer.rs:451-492 */
        uint64_t n_es = scene->n_emissive_spheres, n_ep = scene->n_emissive_planes;
        uint64_t num_light_sources = n_es + n_ep;
        if (num_light_sources == 0) return zero; /* the reference panics on `% 0`; defined as black here */
        uint64_t r = (uint64_t)draw_u32(c, node, 0) % num_light_sources;
        pr_vec3 sample_point, light_radiance;
        int some;
        if (r < n_es) {
            light_radiance = scene->emissive_spheres[r].material.color;
            some = sphere_sample(c, node, &sample_point);
        } else {
            /* tracer.rs:464 indexes emissive_planes[r] WITHOUT subtracting n_es; out of range panics in
             * the reference, defined as black here. */
            if (r >= n_ep) return zero;
            light_radiance = scene->emissive_planes[r].material.color;
            some = plane_sample(c, &scene->emissive_planes[r], origin, node, &sample_point);
        }
        if (!some) return zero;
        pr_vec3 light_ray = vsub(sample_point, origin);
        float light_distance = vlength(light_ray);
        pr_hit h;
        if (find_scene_hit(scene, c->axes, origin, vnormalize(light_ray), &h, c->stats)) {
            if (h.t > light_distance - 0.000001f) return light_radiance;
            return zero;
        }
        return zero;
    }

    pr_hit hit;
    if (find_scene_hit(scene, c->axes, origin, direction, &hit, c->stats)) {
        const float SHIFT_AMOUNT = 0.0001f;
        const float R = 0.04f;
        const pr_material *m = prim_material(scene, hit.prim_id);
        pr_vec3 outwards = vadd(hit.position, vscale(hit.normal, SHIFT_AMOUNT));
        pr_vec3 inwards = vsub(hit.position, vscale(hit.normal, SHIFT_AMOUNT));
        switch (m->kind) {
        case PR_MAT_NONE:
            return one;
        case PR_MAT_EMISSIVE:
            return m->color;
        case PR_MAT_MIRROR: {
            pr_vec3 rd = reflect(direction, hit.normal);
            return trace_radiance(c, outwards, vnormalize(rd), depth - 1, 2u * node);
        }
        case PR_MAT_TRANSLUCENT: { /* tracer.rs:511-536: refraction subtree first, then reflection */
            float n1 = hit.transition == 0 ? 1.0f : m->ior;
            float n2 = hit.transition == 0 ? m->ior : 1.0f;
            pr_vec3 refr_d = refract(direction, hit.normal, n1, n2);
            pr_vec3 l_refr = trace_radiance(c, inwards, vnormalize(refr_d), depth - 1, 2u * node);
            pr_vec3 refl_d = reflect(direction, hit.normal);
            pr_vec3 l_refl = trace_radiance(c, outwards, vnormalize(refl_d), depth - 1, 2u * node + 1u);
            float cos_theta = fmax_rs(0.0f, vdot(hit.normal, vneg(direction)));
            pr_vec3 fresnel = fresnel_schlick(c->math, cos_theta, v3(R, R, R));
            pr_vec3 k_refl = fresnel;
            pr_vec3 k_refr = vsub(one, fresnel);
            return vadd(vmul(k_refl, l_refl), vmul(k_refr, l_refr));
        }
        case PR_MAT_PHYSICALLY: { /* tracer.rs:537-569 */
            pr_basis ts = construct_coordinate_system(hit.normal);
            pr_vec3 view = vneg(direction);
            pr_vec3 normal = hit.normal;
            pr_vec3 f0 = mix_vec3(v3(R, R, R), m->color, m->metalness);
            float xi_x = draw_f32(c, node, 0);
            float xi_y = draw_f32(c, node, 1);
            float a = draw_f32(c, node, 2);              /* left operand of `<` first ... */
            uint32_t ch = draw_u32(c, node, 3) % 3u;     /* ... then the channel index */
            float f0c = ch == 0 ? f0.x : (ch == 1 ? f0.y : f0.z);
            if (a < f0c) {
                pr_vec3 h = to_basis(ts, importance_sample_ggx(c->math, xi_x, xi_y, m->roughness));
                pr_vec3 light = reflect(direction, h);
                pr_vec3 L = trace_radiance(c, outwards, vnormalize(light), depth - 1, 2u * node);
                pr_vec3 brdf = brdf_cook_torrance(c->math, view, light, normal, m->roughness, f0);
                return vmul(vscale(brdf, PR_PI), L);
            } else {
                pr_vec3 light = to_basis(ts, importance_sample_cos(c->math, xi_x, xi_y));
                float light_cos_theta = vdot(light, normal);
                pr_vec3 L = trace_radiance(c, outwards, vnormalize(light), depth - 1, 2u * node);
                pr_vec3 lambert = vdivs(m->color, PR_PI);   /* tracer.rs:334-337 */
                return vscale(vmul(vscale(lambert, PR_PI), L), light_cos_theta);
            }
        }
        default:
            return zero;
        }
    }
    /* tracer.rs:571-583 */
    if (scene->sky.kind == PR_SKY_CONSTANT) return scene->sky.radiance;
    if (scene->sky.pixels) return sample_equirectangular(&scene->sky, direction, c->math);
    return one;
}

pr_vec3 ptref_trace_radiance(const pr_scene *scene, const pr_options *opt, uint32_t *xorshift_state,
                             uint32_t pixel, uint32_t sample, pr_vec3 origin, pr_vec3 direction, int32_t depth,
                             pr_stats *stats) {
    pr_ctx c = {scene, opt, xorshift_state, pixel, sample, stats, opt->math_mode, NULL};
    return trace_radiance(&c, origin, direction, depth, 1u);
}

/* ------------------------------------------------------------------------------------------
 * tracer.rs:586-616 tone mapping and gamma
 * ---------------------------------------------------------------------------------------- */
pr_vec3 ptref_tone_map_gamma(const pr_scene *scene, const pr_options *opt, pr_vec3 hdr) {
    int math = opt->math_mode;
    pr_vec3 ldr;
    switch (scene->camera.tone_mapping) {
    case PR_TONE_CLAMP:
        ldr = v3(saturatef32(hdr.x), saturatef32(hdr.y), saturatef32(hdr.z));
        break;
    case PR_TONE_REINHARD:
        ldr = vdiv(hdr, vadd(v3(1.0f, 1.0f, 1.0f), hdr));
        break;
    default: {
        float e = scene->camera.exposure;
        ldr = v3(1.0f - m_exp(math, -e * hdr.x), 1.0f - m_exp(math, -e * hdr.y), 1.0f - m_exp(math, -e * hdr.z));
    }
    }
    const float GAMMA = 1.0f / 2.2f;
    return v3(m_pow(math, ldr.x, GAMMA), m_pow(math, ldr.y, GAMMA), m_pow(math, ldr.z, GAMMA));
}

/* tracer.rs:271-282 */
static inline void assign_pixel8(const uint32_t *p32, uint64_t num_samples, float iso_factor, uint8_t *p8) {
    float n = (float)(uint32_t)num_samples;
    for (int k = 0; k < 3; ++k) p8[k] = f32_as_u8(fmin_rs(255.0f, iso_factor * (float)p32[k] / n));
}

void ptref_assign_pixels8(const pr_scene *scene, const uint32_t *pixels32, uint64_t num_samples, uint8_t *pixels8) {
    float iso_factor = scene->camera.iso / 100.0f;
    uint64_t n = (uint64_t)scene->width * scene->height;
    for (uint64_t i = 0; i < n; ++i) assign_pixel8(pixels32 + 3 * i, num_samples, iso_factor, pixels8 + 3 * i);
}

/* ------------------------------------------------------------------------------------------
 * tracer.rs:618-647 render(tile)
 * ---------------------------------------------------------------------------------------- */
static int want_jitter(const pr_scene *scene, const pr_options *opt) {
    return opt->sub_pixel_override < 0 ? (scene->sub_pixel_sampling != 0) : (opt->sub_pixel_override != 0);
}

static void primary_ray(pr_ctx *c, uint32_t x, uint32_t y, pr_vec3 *o, pr_vec3 *d) {
    int jitter = want_jitter(c->scene, c->opt);
    float ju = 0.0f, jv = 0.0f;
    if (jitter) { ju = draw_f32(c, 0, 0); jv = draw_f32(c, 0, 1); }
    ptref_pinhole_ray(c->scene, c->opt, x, y, ju, jv, jitter, o, d);
    if (c->opt->camera_model == 1) {  /* two more draws after the pinhole ray's (tracer.rs:183-184) */
        float r1 = draw_f32(c, 0, 2);
        float r2 = draw_f32(c, 0, 3);
        ptref_thin_lens_ray(c->scene, c->opt, r1, r2, o, d);
    }
}

static pr_plane_axes *hoisted_axes(const pr_scene *scene, const pr_options *opt) {
    if (!opt->hoist_plane_axes) return NULL;
    uint64_t n = scene->n_planes + scene->n_emissive_planes;
    pr_plane_axes *ax = (pr_plane_axes *)malloc(sizeof(pr_plane_axes) * (size_t)(n ? n : 1));
    for (uint64_t i = 0; i < n; ++i) {
        const pr_plane *pl = i < scene->n_planes ? &scene->planes[i] : &scene->emissive_planes[i - scene->n_planes];
        ax[i].u_hat = vnormalize(pl->u); ax[i].u_len = vlength(pl->u);
        ax[i].v_hat = vnormalize(pl->v); ax[i].v_len = vlength(pl->v);
    }
    return ax;
}

static void render_tile(const pr_scene *scene, const pr_options *opt, const pr_plane_axes *axes, pr_backbuffer *bb,
                        uint32_t *xs, uint32_t sample_index, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1,
                        pr_stats *stats) {
    float iso_factor = scene->camera.iso / 100.0f;
    pr_ctx c = {scene, opt, xs, 0, sample_index, stats, opt->math_mode, axes};
    for (uint32_t y = y0; y < y1; ++y) {
        for (uint32_t x = x0; x < x1; ++x) {
            uint32_t index = y * bb->width + x;
            c.pixel = index;
            pr_vec3 o, d;
            primary_ray(&c, x, y, &o, &d);
            pr_vec3 hdr = trace_radiance(&c, o, d, opt->max_depth, 1u);
            pr_vec3 g = ptref_tone_map_gamma(scene, opt, hdr);
            uint32_t q[3];
            ptref_pixel32_from_unit(g, q);
            uint32_t *p32 = bb->pixels32 + 3u * (size_t)index;  /* tracer.rs:258-269, wrapping add */
            p32[0] += q[0]; p32[1] += q[1]; p32[2] += q[2];
            assign_pixel8(p32, bb->num_samples, iso_factor, bb->pixels8 + 3u * (size_t)index);
        }
    }
}

#define TILE_SIZE 32u /* window.rs:167 */

void ptref_render_pass(const pr_scene *scene, const pr_options *opt, pr_backbuffer *bb, uint32_t *xorshift_state,
                       uint32_t sample_index, pr_stats *stats) {
    uint32_t ntx = (bb->width + TILE_SIZE - 1) / TILE_SIZE, nty = (bb->height + TILE_SIZE - 1) / TILE_SIZE;
    pr_plane_axes *axes = hoisted_axes(scene, opt);
    for (uint32_t ty = 0; ty < nty; ++ty)
        for (uint32_t tx = 0; tx < ntx; ++tx) {
            uint32_t x0 = tx * TILE_SIZE, y0 = ty * TILE_SIZE;
            uint32_t x1 = x0 + TILE_SIZE, y1 = y0 + TILE_SIZE;
            /* the reference never clips (window.rs:174-175) and panics past the image; clipped here */
            if (x1 > bb->width) x1 = bb->width;
            if (y1 > bb->height) y1 = bb->height;
            render_tile(scene, opt, axes, bb, xorshift_state, sample_index, x0, y0, x1, y1, stats);
        }
    free(axes);
}

void ptref_primary_ids(const pr_scene *scene, const pr_options *opt, uint32_t sample_index, uint32_t *ids) {
    uint32_t xs = opt->seed ? opt->seed : 314159265u;
    pr_ctx c = {scene, opt, &xs, 0, sample_index, NULL, opt->math_mode, NULL};
    for (uint32_t y = 0; y < scene->height; ++y)
        for (uint32_t x = 0; x < scene->width; ++x) {
            uint32_t index = y * scene->width + x;
            c.pixel = index;
            pr_vec3 o, d;
            primary_ray(&c, x, y, &o, &d);
            pr_hit h;
            find_scene_hit(scene, NULL, o, d, &h, NULL);
            ids[index] = h.prim_id;
        }
}

/* ------------------------------------------------------------------------------------------
 * worker.rs:18-130 WorkerPool: FIFO of tiles under one mutex, n threads, barrier = queue drained
 * and every worker idle.  Threads live for one pass here (the reference keeps them forever).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    const pr_scene *scene; const pr_options *opt; pr_backbuffer *bb;
    const pr_plane_axes *axes;
    uint32_t sample_index;
    pthread_mutex_t mu;
    uint32_t next_tile, n_tiles, ntx;
} pool_shared;

typedef struct { pool_shared *sh; uint32_t xs; pr_stats stats; } pool_thread;

static void *pool_main(void *arg) {
    pool_thread *t = (pool_thread *)arg;
    pool_shared *sh = t->sh;
    for (;;) {
        pthread_mutex_lock(&sh->mu);   /* worker.rs:72-73 pop_front under the queue mutex */
        uint32_t tile = sh->next_tile;
        if (tile < sh->n_tiles) sh->next_tile = tile + 1;
        pthread_mutex_unlock(&sh->mu);
        if (tile >= sh->n_tiles) break;
        uint32_t tx = tile % sh->ntx, ty = tile / sh->ntx;
        uint32_t x0 = tx * TILE_SIZE, y0 = ty * TILE_SIZE, x1 = x0 + TILE_SIZE, y1 = y0 + TILE_SIZE;
        if (x1 > sh->bb->width) x1 = sh->bb->width;
        if (y1 > sh->bb->height) y1 = sh->bb->height;
        render_tile(sh->scene, sh->opt, sh->axes, sh->bb, &t->xs, sh->sample_index, x0, y0, x1, y1, &t->stats);
    }
    return NULL;
}

void ptref_render_pass_pool(const pr_scene *scene, const pr_options *opt, pr_backbuffer *bb, int32_t n_threads,
                            uint32_t sample_index, pr_stats *stats) {
    if (n_threads < 1) n_threads = 1;
    pool_shared sh;
    sh.scene = scene; sh.opt = opt; sh.bb = bb; sh.sample_index = sample_index;
    pr_plane_axes *axes = hoisted_axes(scene, opt);
    sh.axes = axes;
    sh.ntx = (bb->width + TILE_SIZE - 1) / TILE_SIZE;
    sh.n_tiles = sh.ntx * ((bb->height + TILE_SIZE - 1) / TILE_SIZE);
    sh.next_tile = 0;
    pthread_mutex_init(&sh.mu, NULL);
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * (size_t)n_threads);
    pool_thread *pt = (pool_thread *)calloc((size_t)n_threads, sizeof(pool_thread));
    for (int i = 0; i < n_threads; ++i) {
        pt[i].sh = &sh;
        /* one private stream per worker and pass; never 0 (xorshift fixed point) */
        uint32_t s = mix32(opt->seed + 0x9E3779B9u * (uint32_t)i + 0x85EBCA6Bu * sample_index);
        pt[i].xs = s ? s : 1u;
        pthread_create(&th[i], NULL, pool_main, &pt[i]);
    }
    for (int i = 0; i < n_threads; ++i) {
        pthread_join(th[i], NULL);
        if (stats) { stats->rays += pt[i].stats.rays; stats->prim_tests += pt[i].stats.prim_tests; }
    }
    pthread_mutex_destroy(&sh.mu);
    free(th); free(pt); free(axes);
}
