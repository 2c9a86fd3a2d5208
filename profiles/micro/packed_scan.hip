// Would the linear scan be cheaper with TWO rays per lane and packed f32 arithmetic (v_pk_mul_f32 / v_pk_add_f32 /
// v_pk_fma_f32: two IEEE operations per lane and instruction, issued at the rate of one)?  The sample scene's scan - seven
// axis-aligned rectangles and three spheres, the tests of pt_scene.h operation for operation - written twice:
//   k_scan1: one ray per lane (what the fused kernel does),
//   k_scan2: two rays per lane, every mul / add / sub / fma of the two rays in one packed instruction; comparisons,
//            selects, the division's scale / reciprocal / fix-up steps and the square root's estimate have no packed form
//            and run once per ray.
// Both produce the same hits (checked).  Reported: time per ray and, from the ISA (hipcc -S), VALU instructions per ray and
// registers.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../pathtracer_amd/csrc -o packed_scan packed_scan.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pt_scene.h"

using namespace pt;
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef const f4v __attribute__((address_space(4))) *cf4;

struct Box { float4 sph[3]; float4 pln[7 * 4]; };

__device__ __forceinline__ void make_ray(uint32_t i, V3 &o, V3 &d) {
    uint32_t h = mix32(i * 2654435761u + 12345u);
    float a = (float)(h & 0xffff) / 65536.0f, b = (float)(h >> 16) / 65536.0f, c = (float)(mix32(h) & 0xffff) / 65536.0f;
    o = v3(-4.0f + 8.0f * a, 0.5f + 9.0f * b, -4.0f + 8.0f * c);
    uint32_t g = mix32(h ^ 0x9e3779b9u);
    V3 dir = v3((float)(g & 0x3ff) / 512.0f - 1.0f, (float)((g >> 10) & 0x3ff) / 512.0f - 1.0f, (float)((g >> 20) & 0x3ff) / 512.0f - 1.0f);
    d = normalize(dir + v3(1e-3f, 2e-3f, 3e-3f));
}

__global__ __launch_bounds__(256) void k_scan1(const Box *box, uint32_t rays_per_lane, uint32_t *out) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc = 0;
    for (uint32_t r = 0; r < rays_per_lane; ++r) {
        V3 o, d;
        make_ray(lane * rays_per_lane + r, o, d);
        float t;
        int tr;
        uint32_t prim = nearest_brute<false>((cf4)(uintptr_t)box->sph, 3u, (cf4)(uintptr_t)box->pln, 7u, o, d, t, tr);
        acc = acc * 31u + prim + f2u(t);
    }
    out[lane] = acc;
}

// ---- two rays per lane ------------------------------------------------------------------------------------------------
struct V3x2 { f2 x, y, z; };
__device__ __forceinline__ f2 sel(bool a, bool b, f2 yes, f2 no) { f2 r; r.x = a ? yes.x : no.x; r.y = b ? yes.y : no.y; return r; }
__device__ __forceinline__ f2 div2(f2 n, f2 d) { f2 r; r.x = n.x / d.x; r.y = n.y / d.y; return r; }   // (the expansion's fma steps may pack)
__device__ __forceinline__ f2 sqrt2(f2 v) { f2 r; r.x = sqrt_ieee(v.x); r.y = sqrt_ieee(v.y); return r; }

template <int A>
__device__ __forceinline__ f2 ax(const V3x2 &v) { return A == 0 ? v.x : (A == 1 ? v.y : v.z); }

// plane_test_axis of pt_scene.h for two rays (the rare num == 0 recomputation is left out on both sides of the comparison:
// the rays here never start in a plane)
template <int A>
__device__ __forceinline__ void plane2(float4 q0, float4 q1, float4 q2, float4 q3, const V3x2 &p, const V3x2 &d, f2 &t, bool &h0, bool &h1) {
    constexpr int A1 = (A + 1) % 3, A2 = (A + 2) % 3;
    const float sign = A == 0 ? q0.x : (A == 1 ? q0.y : q0.z);
    const float sA = A == 0 ? q0.w : (A == 1 ? q1.x : q1.y);
    const float s1 = A1 == 0 ? q0.w : (A1 == 1 ? q1.x : q1.y), s2 = A2 == 0 ? q0.w : (A2 == 1 ? q1.x : q1.y);
    const float un1 = A1 == 0 ? q1.z : (A1 == 1 ? q1.w : q2.x), vn2 = A2 == 0 ? q2.z : (A2 == 1 ? q2.w : q3.x);
    f2 denum = sign * ax<A>(d);
    bool ok0 = denum.x < 0.0f, ok1 = denum.y < 0.0f;
    f2 num = sign * (sA - ax<A>(p));
    ok0 &= !(__builtin_fabsf(denum.x) < 0.0000001f); ok1 &= !(__builtin_fabsf(denum.y) < 0.0000001f);
    f2 parameter = div2(num, denum);
    ok0 &= !(parameter.x < 0.0f); ok1 &= !(parameter.y < 0.0f);
    f2 on1 = ((ax<A1>(p) + parameter * ax<A1>(d)) - s1) * un1;
    ok0 &= !(on1.x > q2.y) & !(on1.x < 0.0f); ok1 &= !(on1.y > q2.y) & !(on1.y < 0.0f);
    f2 on2 = ((ax<A2>(p) + parameter * ax<A2>(d)) - s2) * vn2;
    ok0 &= !(on2.x > q3.y) & !(on2.x < 0.0f); ok1 &= !(on2.y > q3.y) & !(on2.y < 0.0f);
    t = parameter;
    h0 = ok0; h1 = ok1;
}

// sphere_test of pt_scene.h for two rays (no early exits: every condition a bit)
__device__ __forceinline__ void sphere2(float4 sp, const V3x2 &p, const V3x2 &d, f2 &t, bool &h0, bool &h1, int &o0, int &o1) {
    const float r = sp.w;
    f2 cx = sp.x - p.x, cy = sp.y - p.y, cz = sp.z - p.z;
    f2 e_len = (d.x * cx + d.y * cy) + d.z * cz;
    bool ok0 = !(e_len.x < -r), ok1 = !(e_len.y < -r);
    f2 vx = e_len * d.x - cx, vy = e_len * d.y - cy, vz = e_len * d.z - cz;
    f2 v2 = (vx * vx + vy * vy) + vz * vz;
    const float r2 = r * r;
    ok0 &= !(v2.x > r2 * 1.000002f); ok1 &= !(v2.y > r2 * 1.000002f);
    if (__builtin_amdgcn_ballot_w64(ok0 | ok1) == 0ull) { h0 = h1 = false; return; }   // (a wave passing at a distance skips the roots, as sphere_test's exits do)
    f2 v_len = sqrt2(v2);
    ok0 &= !(v_len.x > r); ok1 &= !(v_len.y > r);
    f2 b = sqrt2(r2 - v_len * v_len);
    f2 t1 = e_len - b, t2 = e_len + b;
    o0 = t1.x >= 0.0f ? 0 : 1; o1 = t1.y >= 0.0f ? 0 : 1;
    ok0 &= (t1.x >= 0.0f) | (t2.x >= 0.0f); ok1 &= (t1.y >= 0.0f) | (t2.y >= 0.0f);
    t = sel(t1.x >= 0.0f, t1.y >= 0.0f, t1, t2);
    h0 = ok0; h1 = ok1;
}

__global__ __launch_bounds__(256) void k_scan2(const Box *box, uint32_t rays_per_lane, uint32_t *out) {
    const uint32_t lane = blockIdx.x * 256 + threadIdx.x;
    uint32_t acc0 = 0, acc1 = 0;
    cf4 sph = (cf4)(uintptr_t)box->sph, pln = (cf4)(uintptr_t)box->pln;
    for (uint32_t r = 0; r < rays_per_lane; r += 2) {
        V3 oa, da, ob, db;
        make_ray(lane * rays_per_lane + r, oa, da);
        make_ray(lane * rays_per_lane + r + 1, ob, db);
        V3x2 p, d;
        p.x = f2{oa.x, ob.x}; p.y = f2{oa.y, ob.y}; p.z = f2{oa.z, ob.z};
        d.x = f2{da.x, db.x}; d.y = f2{da.y, db.y}; d.z = f2{da.z, db.z};
        uint32_t best0 = MISS, best1 = MISS;
        f2 best_t = f2{0.0f, 0.0f};
        for (uint32_t i = 0; i < 3; ++i) {
            f2 t; bool h0, h1; int o0 = 0, o1 = 0;
            float4 sp = make_float4(sph[i].x, sph[i].y, sph[i].z, sph[i].w);
            sphere2(sp, p, d, t, h0, h1, o0, o1);
            const bool k0 = h0 && (best0 == MISS || t.x < best_t.x), k1 = h1 && (best1 == MISS || t.y < best_t.y);
            best0 = k0 ? i : best0; best1 = k1 ? i : best1;
            best_t = sel(k0, k1, t, best_t);
        }
        for (uint32_t i = 0; i < 7; ++i) {
            float4 q0 = make_float4(pln[4 * i].x, pln[4 * i].y, pln[4 * i].z, pln[4 * i].w), q1 = make_float4(pln[4 * i + 1].x, pln[4 * i + 1].y, pln[4 * i + 1].z, pln[4 * i + 1].w);
            float4 q2 = make_float4(pln[4 * i + 2].x, pln[4 * i + 2].y, pln[4 * i + 2].z, pln[4 * i + 2].w), q3 = make_float4(pln[4 * i + 3].x, pln[4 * i + 3].y, pln[4 * i + 3].z, pln[4 * i + 3].w);
            const uint32_t code = f2u(q3.z);
            f2 t; bool h0, h1;
            if (code == 1u) plane2<0>(q0, q1, q2, q3, p, d, t, h0, h1);
            else if (code == 2u) plane2<1>(q0, q1, q2, q3, p, d, t, h0, h1);
            else plane2<2>(q0, q1, q2, q3, p, d, t, h0, h1);
            const bool k0 = h0 && (best0 == MISS || t.x < best_t.x), k1 = h1 && (best1 == MISS || t.y < best_t.y);
            best0 = k0 ? 3u + i : best0; best1 = k1 ? 3u + i : best1;
            best_t = sel(k0, k1, t, best_t);
        }
        acc0 = acc0 * 31u + best0 + f2u(best_t.x);
        acc0 = acc0 * 31u + best1 + f2u(best_t.y);
        (void)acc1;
    }
    out[lane] = acc0;
}

static void put_plane(Box &b, int k, V3 n, V3 s, V3 u, V3 v) {
    V3 un = normalize(u), vn = normalize(v);
    float ul = length(u), vl = length(v);
    bool swap_edges;
    uint32_t code = plane_axis_code(n, s, un, ul, vn, vl, swap_edges);
    if (swap_edges) { std::swap(un, vn); std::swap(ul, vl); }
    b.pln[4 * k] = make_float4(n.x, n.y, n.z, s.x);
    b.pln[4 * k + 1] = make_float4(s.y, s.z, un.x, un.y);
    b.pln[4 * k + 2] = make_float4(un.z, ul, vn.x, vn.y);
    b.pln[4 * k + 3] = make_float4(vn.z, vl, u2f(code), 0.0f);
}

int main() {
    Box h{};
    h.sph[0] = make_float4(2.0f, 2.0f, 0.0f, 2.0f); h.sph[1] = make_float4(-2.2f, 6.2f, -0.5f, 2.0f); h.sph[2] = make_float4(-1.8f, 1.5f, 3.0f, 1.5f);
    auto pl = [&](int k, V3 s, V3 u, V3 v) { put_plane(h, k, normalize(cross(u, v)), s, u, v); };
    pl(0, v3(-5, 10, -5), v3(10, 0, 0), v3(0, 0, 10)); pl(1, v3(-5, 0, 5), v3(10, 0, 0), v3(0, 0, -10)); pl(2, v3(-5, 0, -5), v3(10, 0, 0), v3(0, 10, 0));
    pl(3, v3(-5, 0, -5), v3(0, 10, 0), v3(0, 0, 10)); pl(4, v3(5, 0, 5), v3(0, 10, 0), v3(0, 0, -10)); pl(5, v3(5, 0, 5), v3(-10, 0, 0), v3(0, 10, 0));
    pl(6, v3(-2.5f, 9.9f, -2.5f), v3(5, 0, 0), v3(0, 0, 5));
    Box *d_box; uint32_t *d_out;
    const uint32_t blocks = 256 * 24, lanes = blocks * 256, rays_per_lane = 64;
    CHK(hipMalloc((void **)&d_box, sizeof h)); CHK(hipMemcpy(d_box, &h, sizeof h, hipMemcpyHostToDevice));
    CHK(hipMalloc((void **)&d_out, lanes * 4));
    std::vector<uint32_t> r1(lanes), r2(lanes);
    for (int variant = 1; variant <= 2; ++variant) {
        for (int rep = 0; rep < 3; ++rep) {
            auto t0 = std::chrono::steady_clock::now();
            if (variant == 1) k_scan1<<<blocks, 256>>>(d_box, rays_per_lane, d_out); else k_scan2<<<blocks, 256>>>(d_box, rays_per_lane, d_out);
            CHK(hipDeviceSynchronize());
            double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (rep == 2) printf("k_scan%d: %.3f ms, %.1f Grays/s\n", variant, 1e3 * s, (double)lanes * rays_per_lane / s / 1e9);
        }
        CHK(hipMemcpy(variant == 1 ? r1.data() : r2.data(), d_out, lanes * 4, hipMemcpyDeviceToHost));
    }
    // (the two accumulate the same (prim, t) pairs in the same order per lane)
    size_t bad = 0;
    for (uint32_t i = 0; i < lanes; ++i) bad += r1[i] != r2[i];
    printf("lanes whose hits differ: %zu of %u\n", bad, lanes);
    return 0;
}
