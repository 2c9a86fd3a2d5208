// pt_scene.h -- flattened scene as the kernels see it, ray/primitive intersection and surface shading.
//
// Follows the reference function by function (paths relative to the reference checkout):
//   src/scene.rs:169-221   intersect_sphere          -> sphere_test / sphere_hit
//   src/scene.rs:223-261   intersect_plane           -> plane_test / plane_hit
//   src/scene.rs:263-297   find_scene_hit            -> nearest_brute (linear scan) / pt_bvh.h (BVH)
//   src/tracer.rs:285-446  reflect, refract, bases, BRDFs, samplers
//   src/tracer.rs:448-584  trace_radiance            -> shade_hit (one call level, children returned instead
//                                                       of recursed into) + node_value (the combine step)
//   src/tracer.rs:586-616  tone mapping, gamma       -> tonemap_gamma
// All arithmetic keeps the reference's evaluation order (see pt_math.h for the vector-op definitions).
#pragma once

#include "pt_math.h"

namespace pt {

constexpr uint32_t MISS = 0xFFFFFFFFu;

// Material kinds, reference src/scene.rs:84-92
enum : uint32_t { MAT_NONE = 0, MAT_EMISSIVE = 1, MAT_MIRROR = 2, MAT_TRANSLUCENT = 3, MAT_PHYSICALLY = 4 };
enum : int { TONE_CLAMP = 0, TONE_REINHARD = 1, TONE_EXPOSURE = 2 };
enum : int { SKY_CONSTANT = 0, SKY_HDRI = 1 };

// Device view of a Scene (reference src/scene.rs:151-160) after host-side flattening.  Primitive ids run
// over spheres, emissive spheres, planes, emissive planes in that order (src/scene.rs:266,280).
struct DevScene {
    uint32_t n_sph;          // spheres + emissive spheres
    uint32_t n_pl;           // planes + emissive planes
    uint32_t n_es, n_ep;     // emissive spheres / emissive planes (the lights of tracer.rs:453)
    uint32_t n_pl_plain;     // non-emissive planes: emissive plane k has plane slot n_pl_plain + k
    const float4 *sph;       // [n_sph]      (origin.xyz, radius)
    const float4 *pln;       // [n_pl * 4]   (n.xyz, s.x) (s.yz, un.xy) (un.z, |u|, vn.xy) (vn.z, |v|, 0, 0)
    const float4 *mat;       // [(n_sph + n_pl) * 4]  (kind, color.xyz) (roughness, metalness, ior, 0) (f0.xyz, 0) (PI*lambert.xyz, 0)
    const float4 *lights;    // [n_es] radiance of emissive spheres, then [n_ep * 4] emissive planes:
                             //   (origin.xyz, n.x) (n.yz, u.xy) (u.z, v.xyz) (radiance.xyz, 0)
    const float4 *bvh_nodes;    // null: linear scan.  [units * 4] the tree: 64-byte units, each a node or a leaf primitive (pt_bvh.h)
    uint2 *bvh_spill;           // traversal stack entries beyond the LDS part (see pt_bvh.h)
    const uint32_t *bvh_always; // primitives that cannot be bounded: tested for every ray
    uint32_t n_always;
    const float4 *shade_rows;   // [(units + n_always) * 4] what shading needs of a primitive, one 64-byte line each: row u belongs to
                                //   the leaf primitive in unit u of the tree (rows of node units are unused), the `always` list follows:
                                //   (kind, color.xyz) (roughness, metalness, ior, bits(primitive id)) (f0.xyz, 0) (plane normal or sphere
                                //   centre .xyz, 0); the BVH trace kernel reports hits by this row
    uint32_t n_bvh_units;       // units of the tree (the rows of the `always` primitives follow)
    const uint32_t *prim_rows;  // [n_sph + n_pl] the shading row of every primitive (the inverse of the rows' order)
    float bvh_reach;            // ray origins with every |coordinate| at or below this are "tame" (pt_bvh.h, bvh_begin)
    // camera (reference src/tracer.rs:104-111: per-pixel step vectors of CameraSampler)
    V3 cam_origin, cam_su, cam_sv, cam_pos;
    int tone; float exposure; float iso_factor;
    int sky_kind; V3 sky_radiance; const float *sky_pixels; uint32_t sky_w, sky_h;
    int has_bounds;                // bounds_lo/hi hold the padded box of all primitives (linear-scan scenes): a primary ray that
    V3 bounds_lo, bounds_hi;       // misses it misses everything
    const float *gamma_table;  // [256] smallest x with trunc(255 * x^(1/2.2)) >= k (see gamma_quantise); may be null
    uint32_t width, height;
    int jitter;
    uint32_t seed;
    int max_depth;
    int thin_lens;                 // CameraSampler::thin_lens_ray instead of pinhole_ray
    float focus_distance, circle_of_confusion;
};

struct Hit { float t; V3 position; V3 normal; int transition; };  // transition 0 = In, 1 = Out

// ---- spheres: reference src/scene.rs:169-221 ---------------------------------------------------------
// Accept/reject and distance only (what the nearest-hit search needs).
// `out` = 1 when the accepted root is the far one (Transition::Out: the ray starts inside the sphere).
template <class T4>
PT_HD bool sphere_test(T4 sp, V3 p, V3 d, float &t, int &out) {
    V3 s = v3(sp.x, sp.y, sp.z);
    float r = sp.w;
    V3 c = s - p;
    float e_len = dot(d, c);
    if (e_len < -r) return false;
    V3 e = e_len * d;
    V3 v = e - c;
    // `v_len > r` decided without the square root when it is not close: v2 > RN(r*r) * (1 + 2^-19) implies
    // sqrt(v2) > r * (1 + 2^-21), more than an ulp above r, so the rounded length exceeds r as well.  A wave whose
    // rays all pass a sphere at a distance skips both roots; NaN and overflow fall through to the exact test.
    float v2 = dot(v, v);
    float r2 = r * r;
    if (v2 > r2 * 1.000002f) return false;
    float v_len = sqrt_ieee(v2);  // = length(v)
    if (v_len > r) return false;
    float b = sqrt_ieee(r2 - v_len * v_len);
    float t1 = e_len - b;
    float t2 = e_len + b;
    if (t1 >= 0.0f) { t = t1; out = 0; return true; }
    if (t2 >= 0.0f) { t = t2; out = 1; return true; }
    return false;
}
template <class T4>
PT_HD bool sphere_test(T4 sp, V3 p, V3 d, float &t) { int out; return sphere_test(sp, p, d, t, out); }
// The first two rejections of sphere_test alone (the same operations on the same operands): false = certainly no hit.
template <class T4>
PT_HD bool sphere_may_hit(T4 sp, V3 p, V3 d) {
    V3 s = v3(sp.x, sp.y, sp.z);
    float r = sp.w;
    V3 c = s - p;
    float e_len = dot(d, c);
    if (e_len < -r) return false;
    V3 v = e_len * d - c;
    return !(dot(v, v) > (r * r) * 1.000002f);
}
// Full hit record of a sphere already known to be hit.
PT_HD bool sphere_hit(float4 sp, V3 p, V3 d, Hit &h) {
    V3 s = v3(sp.x, sp.y, sp.z);
    float r = sp.w;
    V3 c = s - p;
    float e_len = dot(d, c);
    if (e_len < -r) return false;
    V3 e = e_len * d;
    V3 v = e - c;
    float v_len = length(v);
    if (v_len > r) return false;
    float b = sqrt_ieee(r * r - v_len * v_len);
    float t1 = e_len - b;
    float t2 = e_len + b;
    if (t1 >= 0.0f) {
        h.t = t1; h.position = p + t1 * d; h.normal = normalize(h.position - s); h.transition = 0;
        return true;
    }
    if (t2 >= 0.0f) {
        h.t = t2; h.position = p + t2 * d; h.normal = normalize(s - h.position); h.transition = 1;
        return true;
    }
    return false;
}

// ---- planes: reference src/scene.rs:223-261 ----------------------------------------------------------
// The reference recomputes normalize(u), length(u), normalize(v), length(v) per ray; they depend on the
// plane only, so the host stores them (same operations, same bits).  find_scene_hit keeps a plane hit only
// when its transition is In (src/scene.rs:282), i.e. dot(d, n) < 0; dot(d, n) and `denum` are the same sum.
template <class T4>
PT_HD bool plane_test(T4 q0, T4 q1, T4 q2, T4 q3, V3 p, V3 d, float &t, V3 &position) {
    V3 n = v3(q0.x, q0.y, q0.z);
    V3 s = v3(q0.w, q1.x, q1.y);
    float denum = n.x * d.x + n.y * d.y + n.z * d.z;
    // The last condition of the reference's chain (Transition::In, dot(d, n) < 0: the same three products and sums
    // as denum) is tested first: a plane seen from behind costs five instructions instead of a division and two
    // more dot products.  All conditions are pure, so the order does not change which hits are accepted.
    if (!(denum < 0.0f)) return false;
    float num = n.x * (s.x - p.x) + n.y * (s.y - p.y) + n.z * (s.z - p.z);
    if (__builtin_fabsf(denum) < 0.0000001f) return false;
    float parameter = num / denum;
    if (parameter < 0.0f) return false;
    position = p + parameter * d;
    V3 in_plane = position - s;
    float on_u = dot(in_plane, v3(q1.z, q1.w, q2.x));
    if (on_u > q2.y || on_u < 0.0f) return false;
    float on_v = dot(in_plane, v3(q2.z, q2.w, q3.x));
    if (on_v > q3.y || on_v < 0.0f) return false;
    t = parameter;
    return true;
}

// The same test for a rectangle whose normal and both edges run along coordinate axes (word q3.z names the
// axes, 0 = any other plane: see plane_axis_code).  With n, normalize(u), normalize(v) each made of one +-1 and two
// zeros, every three-term sum of plane_test has at most one non-zero term, and adding the +-0 of the others does not
// change a non-zero value - PROVIDED the ray is finite (0 * inf or 0 * NaN would poison the sum: the caller checks
// the ray and sends non-finite ones through plane_test) and small enough that nothing overflows on the way (|p| <
// 1e29 and, checked on the host, |s| < 1e29 give |t| < 2e36).  Where such a sum IS zero its sign can depend on the
// dropped terms: for denum, on_u and on_v only comparisons follow, which do not see the sign of a zero; num feeds the
// division, so a zero num is recomputed in full.  Everything else is the same operation on the same operands, which
// makes the accepted hits and their t identical to plane_test's; the test drops from ~60 to ~40 instructions.
template <int A> PT_HD float axis_of(V3 v) { return A == 0 ? v.x : (A == 1 ? v.y : v.z); }
// u runs along axis (A+1)%3 and v along (A+2)%3: the host stores the two edges of such a plane in that order (both
// edge conditions must hold, so which is called u does not change the outcome).
template <int A, class T4>
PT_HD bool plane_test_axis(T4 q0, T4 q1, T4 q2, T4 q3, V3 p, V3 d, float &t) {
    constexpr int A1 = (A + 1) % 3, A2 = (A + 2) % 3;
    V3 n = v3(q0.x, q0.y, q0.z);
    V3 s = v3(q0.w, q1.x, q1.y);
    V3 un = v3(q1.z, q1.w, q2.x), vn = v3(q2.z, q2.w, q3.x);
    float sign = axis_of<A>(n);
    float denum = sign * axis_of<A>(d);
    if (!(denum < 0.0f)) return false;
    float num = sign * (axis_of<A>(s) - axis_of<A>(p));
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(num == 0.0f) != 0ull) {
        asm volatile("" ::: "memory");  // keeps this a branch: the full sum stays out of the common path
#else
    {
#endif
        if (num == 0.0f) num = n.x * (s.x - p.x) + n.y * (s.y - p.y) + n.z * (s.z - p.z);
    }
    // From here on no lane leaves early: a wave runs every instruction below for as long as one of its lanes needs
    // it, so returning early saves nothing, while each nested exit costs scalar instructions for the EXEC mask.
    bool ok = !(__builtin_fabsf(denum) < 0.0000001f);
    float parameter = num / denum;
    ok &= !(parameter < 0.0f);
    float on1 = ((axis_of<A1>(p) + parameter * axis_of<A1>(d)) - axis_of<A1>(s)) * axis_of<A1>(un);
    ok &= !(on1 > q2.y) & !(on1 < 0.0f);
    float on2 = ((axis_of<A2>(p) + parameter * axis_of<A2>(d)) - axis_of<A2>(s)) * axis_of<A2>(vn);
    ok &= !(on2 > q3.y) & !(on2 < 0.0f);
    t = parameter;
    return ok;
}

// Host side of the above: 0 when the plane does not qualify, else 1 + the normal's axis A; `swap_edges` tells the
// caller to store v in u's place and u in v's so that the stored u runs along (A+1)%3.
inline uint32_t plane_axis_code(V3 n, V3 s, V3 un, float ul, V3 vn, float vl, bool &swap_edges) {
    swap_edges = false;
    auto axis = [](V3 v) -> int {  // the axis of a vector made of one +-1 and two zeros, else -1
        float c[3] = {v.x, v.y, v.z};
        int a = -1;
        for (int k = 0; k < 3; ++k) {
            if (c[k] == 1.0f || c[k] == -1.0f) { if (a >= 0) return -1; a = k; }
            else if (c[k] != 0.0f) return -1;   // NaN lands here too
        }
        return a;
    };
    int an = axis(n), au = axis(un), av = axis(vn);
    if (an < 0 || au < 0 || av < 0 || an == au || an == av || au == av) return 0;
    const float lim = 1e29f;
    if (!(__builtin_fabsf(s.x) < lim && __builtin_fabsf(s.y) < lim && __builtin_fabsf(s.z) < lim && ul < lim && vl < lim)) return 0;
    swap_edges = au != (an + 1) % 3;
    return 1u + (uint32_t)an;
}

// ---- nearest hit by linear scan: reference src/scene.rs:263-297 ---------------------------------------
// A later primitive replaces the current nearest only when strictly closer.  P4 is any pointer to a 4-float
// vector (the kernels pass constant-address-space pointers so the wave-uniform primitive fetches become
// scalar loads into SGPRs).
// DEFER (device only, a kernel-level choice for scenes with more than DEFER_SPHERES spheres): see below.
constexpr uint32_t DEFER_SPHERES = 8;
// `own_sph`: where a lane reads ITS candidate sphere from (the same table; the kernels pass their LDS copy, P4 itself is
// a scalar-load pointer that suits fetches every lane shares).
template <bool DEFER = false, class P4, class Q4 = P4>
PT_HD uint32_t nearest_brute(P4 sph, uint32_t n_sph, P4 pln, uint32_t n_pl, V3 p, V3 d, float &best_t, int &best_out, Q4 own_sph = Q4()) {
    uint32_t best = MISS;
    best_t = 0.0f;
    best_out = 0;  // Transition of the winner (planes are only ever accepted as In)
#if defined(__HIP_DEVICE_COMPILE__)
    // Every sphere's cheap rejections first, for all lanes at once; what survives (a few per cent of the lanes per
    // sphere) is then tested in full lane by lane, each lane on ITS next candidate: the square roots run once or twice per
    // wave instead of once per sphere that any lane might hit.  A lane takes its candidates in ascending order and
    // replaces only when strictly closer, as the scan over all spheres does.  Measured at 1080p: the room with 20
    // spheres +9 % (candidates read from the block's LDS copy of the scene; +4 % when read from global memory), the
    // sample scene with 3 spheres -1.3 % (the bookkeeping costs more than two skipped rounds of roots save), hence the
    // threshold - and a template parameter, because the mere presence of this loop in the sample scene's kernel costs
    // it 0.5 %.
    const bool deferred = DEFER;
    for (uint32_t first = 0; deferred && first < n_sph; first += 32u) {
        const uint32_t n = n_sph - first < 32u ? n_sph - first : 32u;
        uint32_t cand = 0;
        for (uint32_t i = 0; i < n; ++i) cand |= sphere_may_hit(sph[first + i], p, d) ? 1u << i : 0u;
        while (__builtin_amdgcn_ballot_w64(cand != 0u) != 0ull) {
            if (cand != 0u) {
                const uint32_t i = first + (uint32_t)__builtin_ctz(cand);
                cand &= cand - 1u;
                float t;
                int out;
                if (sphere_test(own_sph[i], p, d, t, out)) {
                    if (best == MISS || t < best_t) { best = i; best_t = t; best_out = out; }
                }
            }
        }
    }
#else
    const bool deferred = false;
#endif
    for (uint32_t i = 0; !deferred && i < n_sph; ++i) {
        float t;
        int out;
        if (sphere_test(sph[i], p, d, t, out)) {
            if (best == MISS || t < best_t) { best = i; best_t = t; best_out = out; }
        }
    }
    // axis-aligned rectangles take the short test when every ray of the wave is finite (see plane_test_axis)
#if defined(__HIP_DEVICE_COMPILE__)
    const bool finite_ray = (__builtin_fabsf(d.x) + __builtin_fabsf(d.y)) + __builtin_fabsf(d.z) < 4.0f &&
                            (__builtin_fabsf(p.x) + __builtin_fabsf(p.y)) + __builtin_fabsf(p.z) < 1e29f;
    const bool short_ok = __builtin_amdgcn_ballot_w64(!finite_ray) == 0ull;
#else
    const bool short_ok = false;
#endif
    for (uint32_t i = 0; i < n_pl; ++i) {
        float t; V3 pos;
        auto q0 = pln[4 * i], q1 = pln[4 * i + 1], q2 = pln[4 * i + 2], q3 = pln[4 * i + 3];
        const uint32_t code = short_ok ? __builtin_bit_cast(uint32_t, (float)q3.z) : 0u;
        bool hit;
        if (code == 1u) hit = plane_test_axis<0>(q0, q1, q2, q3, p, d, t);
        else if (code == 2u) hit = plane_test_axis<1>(q0, q1, q2, q3, p, d, t);
        else if (code == 3u) hit = plane_test_axis<2>(q0, q1, q2, q3, p, d, t);
        else hit = plane_test(q0, q1, q2, q3, p, d, t, pos);
        // (selects instead of a nested branch: the same three writes without two EXEC-mask round trips)
        const bool take = hit && (best == MISS || t < best_t);
        best = take ? n_sph + i : best;
        best_t = take ? t : best_t;
        best_out = take ? 0 : best_out;
    }
    return best;
}

// Rebuild the full Hit of primitive `prim` for ray (p, d); `prim` must be what the nearest-hit search returned.
PT_HD bool prim_hit(const DevScene &sc, uint32_t prim, V3 p, V3 d, Hit &h) {
    if (prim < sc.n_sph) return sphere_hit(sc.sph[prim], p, d, h);
    uint32_t k = prim - sc.n_sph;
    float4 q0 = sc.pln[4 * k];
    float t; V3 pos;
    if (!plane_test(q0, sc.pln[4 * k + 1], sc.pln[4 * k + 2], sc.pln[4 * k + 3], p, d, t, pos)) return false;
    h.t = t; h.position = pos; h.normal = v3(q0.x, q0.y, q0.z); h.transition = 0;
    return true;
}

// The same Hit from what the nearest-hit search already knows (t and, for spheres, which root): position is the
// expression both tests use (p + t*d), so only the sphere normal is left to compute.
PT_HD void prim_hit_known(const DevScene &sc, uint32_t prim, V3 p, V3 d, float t, int transition, Hit &h) {
    h.t = t;
    h.position = p + t * d;
    if (prim < sc.n_sph) {
        float4 sp = sc.sph[prim];
        V3 s = v3(sp.x, sp.y, sp.z);
        h.transition = transition;
        h.normal = transition == 0 ? normalize(h.position - s) : normalize(s - h.position);
    } else {
        float4 q0 = sc.pln[4 * (prim - sc.n_sph)];
        h.transition = 0;
        h.normal = v3(q0.x, q0.y, q0.z);
    }
}

// ---- optics and BRDFs: reference src/tracer.rs:285-446 ------------------------------------------------
PT_HD V3 reflect(V3 incoming, V3 normal) { return incoming - (2.0f * dot(incoming, normal)) * normal; }  // :285-288

PT_HD V3 refract(V3 incoming, V3 normal, float n1, float n2) {  // :290-318
    V3 v = -incoming;
    float cos_theta1 = dot(v, normal);
    float sin_theta1 = sqrt_ieee(1.0f - cos_theta1 * cos_theta1);
    float sin_theta2 = (sin_theta1 * n1) / n2;
    float cos_theta2 = sqrt_ieee(1.0f - sin_theta2 * sin_theta2);
    V3 p = incoming + cos_theta1 * normal;
    return (-normal) * cos_theta2 + normalize(p) * sin_theta2;
}

struct Basis { V3 x, y, z; };
PT_HD Basis construct_coordinate_system(V3 normal) {  // :320-331 (x is NOT normalised)
    const float EPS = 0.9999f;
    V3 other = (normal.y > EPS || normal.y < -EPS) ? v3(1.0f, 0.0f, 0.0f) : v3(0.0f, 1.0f, 0.0f);
    Basis b;
    b.y = normal;
    b.x = cross(b.y, other);
    b.z = cross(b.y, b.x);
    return b;
}
PT_HD V3 to_basis(const Basis &b, V3 v) { return (b.x * v.x + b.y * v.y) + b.z * v.z; }  // :443-446

PT_HD float geometry_schlick_ggx(V3 n, V3 a, float k) {  // :382-387
    float n_dot_a = max_rs(0.0f, dot(n, a));
    float denom = n_dot_a * (1.0f - k) + k;
    return n_dot_a / denom;
}
PT_HD V3 fresnel_schlick(float cos_theta, V3 f0) {  // :391-393
    return f0 + (v3(1.0f, 1.0f, 1.0f) - f0) * det::powf_(1.0f - cos_theta, 5.0f);
}
PT_HD V3 brdf_cook_torrance(V3 v, V3 l, V3 n, float roughness, V3 f0) {  // :395-415
    float alpha = roughness * roughness;
    V3 h = normalize(v + l);
    float cos_theta = max_rs(0.0f, dot(h, v));
    float k = alpha / 2.0f;
    float g = geometry_schlick_ggx(n, l, k) * geometry_schlick_ggx(n, v, k);  // :370-389
    V3 num = fresnel_schlick(cos_theta, f0) * g;
    float denum = 4.0f * max_rs(0.0f, dot(n, v));
    return num / denum;
}
PT_HD V3 importance_sample_ggx(float xi_x, float xi_y, float roughness) {  // :419-430
    float a = roughness * roughness;
    float phi = 2.0f * PI_F * xi_x;
    float sin_phi, cos_phi;
    det::sincosf_(phi, sin_phi, cos_phi);
    float cos_theta = sqrt_ieee((1.0f - xi_y) / (xi_y * (a * a - 1.0f) + 1.0f));
    float sin_theta = sqrt_ieee(1.0f - cos_theta * cos_theta);
    return v3(sin_theta * cos_phi, cos_theta, sin_theta * sin_phi);
}
PT_HD V3 importance_sample_cos(float xi_x, float xi_y) {  // :433-441
    float r = sqrt_ieee(xi_x);
    float theta = 2.0f * PI_F * xi_y;
    float s, c;
    det::sincosf_(theta, s, c);
    return v3(r * c, sqrt_ieee(max_rs(0.0f, 1.0f - xi_x)), r * s);
}

// ---- one level of trace_radiance: reference src/tracer.rs:494-583 --------------------------------------
// The recursion `L = f(L_child...)` is split: shade_hit decides what kind of node this call is, returns the
// coefficients of f and the child rays; node_value applies f once the children's radiance is known.
enum : uint32_t { NK_LEAF = 0, NK_MIRROR = 1, NK_SPECULAR = 2, NK_DIFFUSE = 3, NK_GLASS = 4 };

struct NodeOut {
    uint32_t kind;
    V3 coef;       // LEAF: the radiance; SPECULAR: PI*brdf; DIFFUSE: PI*lambert; GLASS: fresnel (k_refl)
    float coef_w;  // DIFFUSE: light_cos_theta
    int n_children;
    V3 o0, d0;     // child 0: mirror / specular / diffuse bounce, or the glass REFRACTION ray (path node 2n)
    V3 o1, d1;     // child 1: the glass REFLECTION ray (path node 2n+1)
};

PT_HD V3 sky_radiance(const DevScene &sc, V3 direction);

// shade_hit is split in two so a kernel can learn how many child rays a hit spawns (and where they start) before
// paying for the sampling/BRDF arithmetic:
//   shade_prepare : rebuild the Hit of the nearest primitive, fetch its material, decide the match arm
//                   (reference src/tracer.rs:501-569; for Physically the arm depends on the node's third and fourth
//                   draws only: `random32() < f0[xorshift32() % 3]`, :549) and the child origins (:496-497)
//   shade_finish  : the arm's arithmetic -- child directions and node coefficients
struct ShadeCtx {
    Hit hit;
    float4 m0, m1;     // material rows
    V3 pi_lambert;     // Physically: PI * brdf_lambert(reflectivity)
    V3 f0;             // Physically: mix(0.04, reflectivity, metalness)
    uint32_t kind;     // NK_* arm
    int n_children;
    V3 o0, o1;         // child origins (o1: glass reflection)
    V3 leaf;           // NK_LEAF: the radiance
};


// `transition` < 0: only the primitive id is known, the Hit is rebuilt by re-running its test; otherwise (t,
// transition) come from the nearest-hit search.
// ROWS: `prim` is a slot of sc.shade_rows (what the BVH trace kernel reports) and everything shading needs of the
// primitive comes from that one 64-byte line instead of from two tables indexed by primitive id; (t, transition) are
// known.  PI * brdf_lambert(reflectivity) is then evaluated here with the operations the host uses for the table.
template <bool ROWS = false>
PT_HD void shade_prepare(const DevScene &sc, V3 origin, V3 direction, uint32_t prim, float t, int transition, uint32_t node_key, ShadeCtx &c) {
    c.n_children = 0;
    c.kind = NK_LEAF;
    bool hit = prim != MISS;
    float4 row2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ROWS) {
        if (hit) {
            const float4 *row = sc.shade_rows + 4 * (size_t)prim;
            c.m0 = row[0];
            c.m1 = row[1];
            row2 = row[2];
            float4 geo = row[3];
            c.hit.t = t;
            c.hit.position = origin + t * direction;
            c.hit.transition = 0;
            c.hit.normal = v3(geo.x, geo.y, geo.z);
            if (f2u(c.m1.w) < sc.n_sph) {   // a sphere: geo is its centre (reference src/scene.rs:212,216)
                V3 s = v3(geo.x, geo.y, geo.z);
                c.hit.transition = transition;
                c.hit.normal = transition == 0 ? normalize(c.hit.position - s) : normalize(s - c.hit.position);
            }
        }
    } else if (hit) {
        if (transition >= 0) prim_hit_known(sc, prim, origin, direction, t, transition, c.hit);
        else hit = prim_hit(sc, prim, origin, direction, c.hit);
    }
    if (!hit) {  // :571-583
        c.leaf = sky_radiance(sc, direction);
        return;
    }
    const float SHIFT_AMOUNT = 0.0001f;  // :495
    if (!ROWS) {
        c.m0 = sc.mat[4 * prim];
        c.m1 = sc.mat[4 * prim + 1];
    }
    uint32_t mat = f2u(c.m0.x);
    V3 color = v3(c.m0.y, c.m0.z, c.m0.w);
    V3 outwards = c.hit.position + SHIFT_AMOUNT * c.hit.normal;
    c.o0 = outwards;
    switch (mat) {
    case MAT_NONE: c.leaf = v3(1.0f, 1.0f, 1.0f); return;   // :502
    case MAT_EMISSIVE: c.leaf = color; return;               // :503-505
    case MAT_MIRROR: c.kind = NK_MIRROR; c.n_children = 1; return;  // :506-510
    case MAT_TRANSLUCENT:                                    // :511-536
        c.kind = NK_GLASS;
        c.n_children = 2;
        c.o0 = c.hit.position - SHIFT_AMOUNT * c.hit.normal;
        c.o1 = outwards;
        return;
    default: {                                               // MAT_PHYSICALLY, :537-569
        float4 m2 = ROWS ? row2 : sc.mat[4 * prim + 2];
        c.f0 = v3(m2.x, m2.y, m2.z);  // mix_vec3(0.04, reflectivity, metalness), :543: a constant of the material
        float a = random32(rng_draw(node_key, 2));
        uint32_t ch = rng_draw(node_key, 3) % 3u;
        float f0c = ch == 0 ? c.f0.x : (ch == 1 ? c.f0.y : c.f0.z);
        c.kind = a < f0c ? NK_SPECULAR : NK_DIFFUSE;
        c.n_children = 1;
        if (ROWS) c.pi_lambert = PI_F * (color / PI_F);   // PI * brdf_lambert(reflectivity), :334-337,567
        else {
            float4 m3 = sc.mat[4 * prim + 3];
            c.pi_lambert = v3(m3.x, m3.y, m3.z);
        }
        return;
    }
    }
}

PT_HD void shade_finish(const ShadeCtx &c, V3 direction, uint32_t node_key, NodeOut &out) {
    const float R = 0.04f;
    out.kind = c.kind;
    out.n_children = c.n_children;
    out.coef_w = 0.0f;
    out.o0 = c.o0;
    out.o1 = c.o1;
    const Hit &hit = c.hit;
    switch (c.kind) {
    case NK_LEAF:
        out.coef = c.leaf;
        return;
    case NK_MIRROR:
        out.coef = v3(0.0f, 0.0f, 0.0f);
        out.d0 = normalize(reflect(direction, hit.normal));
        return;
    case NK_GLASS: {
        float ior = c.m1.z;
        float n1 = hit.transition == 0 ? 1.0f : ior;
        float n2 = hit.transition == 0 ? ior : 1.0f;
        out.d0 = normalize(refract(direction, hit.normal, n1, n2));
        out.d1 = normalize(reflect(direction, hit.normal));
        float cos_theta = max_rs(0.0f, dot(hit.normal, -direction));
        out.coef = fresnel_schlick(cos_theta, v3(R, R, R));
        return;
    }
    default: {
        // Specular (:550-561) and diffuse (:562-568) bounces of one wave run as ONE instruction stream: which random
        // number feeds the angle, what goes under the two square roots and how the local vector is assembled are
        // selected per lane, everything else (sin/cos, roots, change of basis, normalisation) is shared.  Every lane
        // still performs exactly the operations of its own arm (importance_sample_ggx / importance_sample_cos).
        const bool spec = c.kind == NK_SPECULAR;
        float roughness = c.m1.x;
        Basis ts = construct_coordinate_system(hit.normal);
        float xi_x = random32(rng_draw(node_key, 0));
        float xi_y = random32(rng_draw(node_key, 1));
        float a = roughness * roughness;
        float angle = 2.0f * PI_F * (spec ? xi_x : xi_y);                                  // phi (:421) / theta (:435)
        float sin_a, cos_a;
        det::sincosf_(angle, sin_a, cos_a);
        float under1 = spec ? (1.0f - xi_y) / (xi_y * (a * a - 1.0f) + 1.0f) : xi_x;       // cos_theta^2 (:423) / r^2 (:434)
        float root1 = sqrt_ieee(under1);
        float under2 = spec ? 1.0f - root1 * root1 : max_rs(0.0f, 1.0f - xi_x);            // sin_theta^2 (:424) / y^2 (:439)
        float root2 = sqrt_ieee(under2);
        float radial = spec ? root2 : root1;                                               // sin_theta / r
        V3 local = v3(radial * cos_a, spec ? root1 : root2, radial * sin_a);               // :426-428 / :437-440
        V3 world = to_basis(ts, local);
        V3 light = spec ? reflect(direction, world) : world;                               // :553 / :563
        out.d0 = normalize(light);
        if (spec) {
            out.coef = PI_F * brdf_cook_torrance(-direction, light, hit.normal, roughness, c.f0);
        } else {
            out.coef = c.pi_lambert;  // PI * brdf_lambert(reflectivity), :334-337,567: a constant of the material
            out.coef_w = dot(light, hit.normal);
        }
        return;
    }
    }
}

// One whole level of trace_radiance for a hit (prepare + finish).
PT_HD void shade_hit(const DevScene &sc, V3 origin, V3 direction, uint32_t prim, uint32_t node_key, NodeOut &out) {
    ShadeCtx c;
    shade_prepare(sc, origin, direction, prim, 0.0f, -1, node_key, c);
    shade_finish(c, direction, node_key, out);
}

// The arm a hit will take, without rebuilding the hit (used to regroup rays by arm).
PT_HD uint32_t shade_class(const DevScene &sc, uint32_t prim, uint32_t node_key) {
    if (prim == MISS) return NK_LEAF;
    float4 m0 = sc.mat[4 * prim];
    uint32_t kind = f2u(m0.x);
    if (kind == MAT_NONE || kind == MAT_EMISSIVE) return NK_LEAF;
    if (kind == MAT_MIRROR) return NK_MIRROR;
    if (kind == MAT_TRANSLUCENT) return NK_GLASS;
    float4 m2 = sc.mat[4 * prim + 2];
    V3 f0 = v3(m2.x, m2.y, m2.z);
    float a = random32(rng_draw(node_key, 2));
    uint32_t ch = rng_draw(node_key, 3) % 3u;
    float f0c = ch == 0 ? f0.x : (ch == 1 ? f0.y : f0.z);
    return a < f0c ? NK_SPECULAR : NK_DIFFUSE;
}

// Radiance of a node from its coefficients and its children's radiance (the tail of each match arm of
// reference src/tracer.rs:501-569).
PT_HD V3 node_value(uint32_t kind, V3 coef, float coef_w, V3 child0, V3 child1) {
    switch (kind) {
    case NK_LEAF: return coef;
    case NK_MIRROR: return child0;                                  // :509
    case NK_SPECULAR: return coef * child0;                         // :560  (PI*brdf) * L
    case NK_DIFFUSE: return (coef * child0) * coef_w;               // :567  ((PI*lambert) * L) * cos
    default: return coef * child1 + (v3(1.0f, 1.0f, 1.0f) - coef) * child0;  // :532-535 k_refl*l_refl + k_refr*l_refr
    }
}

// ---- sky: reference src/tracer.rs:571-583, src/scene.rs:32-68 ------------------------------------------
PT_HD float repeat_f32(float x) {
    float t = (float)(int)x;
    if (x < 0.0f) return x - t + 1.0f;
    return x - t;
}
PT_HD V3 sky_radiance(const DevScene &sc, V3 direction) {
    if (sc.sky_kind == SKY_CONSTANT) return sc.sky_radiance;
    if (sc.sky_pixels == nullptr) return v3(1.0f, 1.0f, 1.0f);  // texture not loaded yet
    float u = det::atan2f_(direction.z, direction.x) / (2.0f * PI_F);
    float v = det::acosf_(direction.y) / PI_F;
    u = repeat_f32(u);
    v = repeat_f32(v);
    float fx = u * (float)sc.sky_w, fy = v * (float)sc.sky_h;
    uint64_t px = (fx > 0.0f) ? (uint64_t)fx : 0u;
    uint64_t py = (fy > 0.0f) ? (uint64_t)fy : 0u;
    uint64_t last = (uint64_t)sc.sky_w * sc.sky_h * 3u - 3u;
    uint64_t index = 3u * (py * sc.sky_w + px);
    if (index > last) index = last;
    return v3(sc.sky_pixels[index], sc.sky_pixels[index + 1], sc.sky_pixels[index + 2]);
}

// ---- terminal light sample: reference src/tracer.rs:451-492, src/scene.rs:116-148 -----------------------
// Returns true when a shadow ray has to be traced; `radiance` is what the node is worth if that ray reaches
// the light (hit.t > light_distance - 1e-6).  Returns false when the node is black.
PT_HD bool terminal_sample(const DevScene &sc, V3 origin, uint32_t node_key, V3 &shadow_dir, float &light_distance, V3 &radiance) {
    uint32_t n_lights = sc.n_es + sc.n_ep;
    if (n_lights == 0) return false;  // the reference panics (`% 0`); defined as black
    uint32_t r = rng_draw(node_key, 0) % n_lights;
    V3 sample_point;
    if (r < sc.n_es) {  // Sphere::sample, src/scene.rs:136-148: unit sphere at the world origin
        float4 rad = sc.lights[r];
        radiance = v3(rad.x, rad.y, rad.z);
        float r1 = random32(rng_draw(node_key, 1));
        float r2 = random32(rng_draw(node_key, 2));
        float theta = 2.0f * PI_F * r1;
        float phi = det::acosf_(2.0f * r2 - 1.0f);
        float st, ct, sp, cp;
        det::sincosf_(theta, st, ct);
        det::sincosf_(phi, sp, cp);
        sample_point = v3(st * cp, ct, st * sp);
    } else {
        // src/tracer.rs:464 indexes emissive_planes[r] without subtracting the sphere count
        if (r >= sc.n_ep) return false;  // out of range: the reference panics; defined as black
        const float4 *L = sc.lights + sc.n_es + 4 * r;
        float4 l0 = L[0], l1 = L[1], l2 = L[2], l3 = L[3];
        V3 lo = v3(l0.x, l0.y, l0.z), ln = v3(l0.w, l1.x, l1.y);
        V3 lu = v3(l1.z, l1.w, l2.x), lv = v3(l2.y, l2.z, l2.w);
        radiance = v3(l3.x, l3.y, l3.z);
        if (dot(lo - origin, ln) < 0.0f) return false;  // Plane::sample, src/scene.rs:117, before any draw
        float r1 = random32(rng_draw(node_key, 1));
        float r2 = random32(rng_draw(node_key, 2));
        sample_point = (lo + r1 * lu) + r2 * lv;
    }
    V3 light_ray = sample_point - origin;
    light_distance = length(light_ray);
    shadow_dir = normalize(light_ray);
    return true;
}

// ---- tone mapping + gamma + quantisation: reference src/tracer.rs:586-616,634-641, src/common.rs:44-57,78-82
PT_HD float clampf32(float mn, float mx, float x) { if (x > mx) return mx; else if (x < mn) return mn; else return x; }
// Pixel32::from_unit(gamma_correction(x)) for one channel: trunc(255 * x^(1/2.2)) saturating (src/tracer.rs:609-616,
// src/common.rs:78-82).  The result is a step function of x, so for 0 <= x <= 1 it is looked up by bisection in the
// table of its 255 step positions, built on the host with the SAME powf_ (PtHandle self-test checks all 2^30
// floats of [0,1] against the direct evaluation).  Anything else (NaN, negative, > 1) takes the direct route.
PT_HD uint32_t gamma_quantise_direct(float x) { return f32_as_u32(det::powf_(x, 1.0f / 2.2f) * 255.0f); }
PT_HD uint32_t gamma_quantise(const float *table, float x) {
    if (table != nullptr && x >= 0.0f && x <= 1.0f) {
        uint32_t lo = 0;
#pragma unroll
        for (uint32_t step = 128; step != 0; step >>= 1) {
            uint32_t k = lo + step;
            if (table[k] <= x) lo = k;
        }
        return lo;
    }
    return gamma_quantise_direct(x);
}

PT_HD void tonemap_gamma_quantise(const DevScene &sc, V3 hdr, uint32_t q[3]) {
    V3 ldr;
    if (sc.tone == TONE_CLAMP) ldr = v3(clampf32(0.0f, 1.0f, hdr.x), clampf32(0.0f, 1.0f, hdr.y), clampf32(0.0f, 1.0f, hdr.z));
    else if (sc.tone == TONE_REINHARD) ldr = hdr / (v3(1.0f, 1.0f, 1.0f) + hdr);
    else {
        float e = sc.exposure;
        ldr = v3(1.0f - det::expf_(-e * hdr.x), 1.0f - det::expf_(-e * hdr.y), 1.0f - det::expf_(-e * hdr.z));
    }
    q[0] = gamma_quantise(sc.gamma_table, ldr.x);
    q[1] = gamma_quantise(sc.gamma_table, ldr.y);
    q[2] = gamma_quantise(sc.gamma_table, ldr.z);
}

// ---- primary rays: reference src/tracer.rs:125-142 ------------------------------------------------------
PT_HD void pinhole_ray(const DevScene &sc, uint32_t x, uint32_t y, uint32_t sample_key, V3 &origin, V3 &direction) {
    float offset_u = (float)x;
    float offset_v = (float)y;
    if (sc.jitter) {
        uint32_t cam_key = rng_node_key(sample_key, 0u);
        offset_u += random32(rng_draw(cam_key, 0));
        offset_v += random32(rng_draw(cam_key, 1));
    }
    V3 du = offset_u * sc.cam_su;
    V3 dv = offset_v * sc.cam_sv;
    origin = (sc.cam_origin + du) + dv;
    direction = normalize(origin - sc.cam_pos);
}

// reference src/tracer.rs:144-197 (dead code there; the camera model its TODOs point to).  Two more draws of the
// camera node after the pinhole ray's.
PT_HD void camera_ray(const DevScene &sc, uint32_t x, uint32_t y, uint32_t sample_key, V3 &origin, V3 &direction) {
    pinhole_ray(sc, x, y, sample_key, origin, direction);
    if (!sc.thin_lens) return;
    uint32_t cam_key = rng_node_key(sample_key, 0u);
    V3 point_on_focus_plane = origin + sc.focus_distance * direction;
    float r1 = random32(rng_draw(cam_key, 2));
    float r2 = random32(rng_draw(cam_key, 3));
    float alpha = 2.0f * PI_F * r1;
    float r = sc.circle_of_confusion * r2;
    float sa, ca;
    det::sincosf_(alpha, sa, ca);
    float l_u = r * sa;
    float l_v = r * ca;
    V3 lens_origin = (origin + l_u * sc.cam_su) + l_v * sc.cam_sv;
    direction = normalize(point_on_focus_plane - lens_origin);
    origin = lens_origin;
}

}  // namespace pt
