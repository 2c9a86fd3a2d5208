// pt_bvh.h -- BVH over the scene's spheres and finite planes: node layout and device traversal.
//
// The reference has no acceleration structure: find_scene_hit (reference src/scene.rs:263-297) scans every
// primitive and keeps the first one, in id order, with the smallest t.  The traversal below returns exactly
// that primitive:
//   * every leaf primitive is tested with the same sphere_test / plane_test the linear scan uses;
//   * a candidate replaces the current best when (t, id) is lexicographically smaller, which is what
//     "strictly closer wins, ties keep the earlier id" means and makes the result independent of visit order;
//   * boxes are padded on the host (see BvhBuilder) by far more than the rounding of the hit point and of the
//     slab test, a subtree is culled only when its entry distance is strictly beyond the best t, and NaN
//     slab distances are ignored, so no primitive the linear scan would accept is ever skipped.
//
// Node = 64 bytes (half a 128-byte cache line, two nodes per line), holding BOTH children's boxes so one
// fetch decides both; four 16-byte loads per lane.  Children: >= 0 inner node index, < 0 leaf
// (~c = first << 3 | count-1 into bvh_geom, ids ascending), BVH_EMPTY = no child.
//
// Leaf primitives live in their own array in leaf order (bvh_geom, 64 bytes each: the plane's four vectors or the
// sphere's one, primitive id in the last word), so a leaf test is ONE dependent fetch after the node, with the same
// shape as a node fetch.  The traversal step below loads 64 bytes from "wherever this lane is" - an inner node or
// the next primitive of its leaf - and then either descends or tests: every lane of a wave makes progress on every
// memory round trip, which is what bounds this loop (measured: the texture/L1 path is ~40 % busy, waves wait on the
// slowest lane's miss; arithmetic is nowhere near a limit).
#pragma once

#include "pt_scene.h"

namespace pt {

struct BvhNode {   // the two children's boxes interleaved, so both slab tests run as one sequence of 2-wide packed f32 operations
    float4 a;   // (lo0.x, lo1.x, lo0.y, lo1.y)
    float4 b;   // (lo0.z, lo1.z, hi0.x, hi1.x)
    float4 c;   // (hi0.y, hi1.y, hi0.z, hi1.z)
    int32_t c0, c1;
    uint32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

constexpr int32_t BVH_EMPTY = (int32_t)0x80000000;
constexpr int BVH_STACK = 40;          // entries per lane; the builder caps the tree depth at this
#ifndef PT_BVH_LDS_ENTRIES
#define PT_BVH_LDS_ENTRIES 8
#endif
constexpr int BVH_LDS_ENTRIES = PT_BVH_LDS_ENTRIES;    // of which this many live in LDS, the rest (rarely reached) in a global spill area
constexpr int BVH_STACK_STRIDE = 256;  // threads per block of every kernel that traverses
constexpr int BVH_MAX_LEAF = 4;
constexpr uint32_t BVH_TRAV_BLOCKS = 2048;  // most blocks a traversing kernel is launched with (sizes the spill area)

#if defined(__HIPCC__)
// Slab test of both children at once.  Distances are lo * inv + (-o * inv) in one fused operation per plane pair
// (v_pk_fma_f32); against (lo - o) * inv that moves a slab by at most ~|o| * 2^-23, far inside the builder's
// padding.  NaN distances (inf - inf when the ray is parallel to a slab, 0 * inf) drop out of fminf/fmaxf, which
// only removes a constraint; tfar is widened by 4 ulp for the rounding of the products.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 min2(f32x2 x, f32x2 y) { return f32x2{fminf(x.x, y.x), fminf(x.y, y.y)}; }
__device__ __forceinline__ f32x2 max2(f32x2 x, f32x2 y) { return f32x2{fmaxf(x.x, y.x), fmaxf(x.y, y.y)}; }
__device__ __forceinline__ void box_entry2(float4 a, float4 b, float4 c, V3 inv, V3 noi, f32x2 &tn, f32x2 &tf) {
    f32x2 ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    f32x2 nx = {noi.x, noi.x}, ny = {noi.y, noi.y}, nz = {noi.z, noi.z};
    f32x2 t0x = __builtin_elementwise_fma(f32x2{a.x, a.y}, ix, nx), t1x = __builtin_elementwise_fma(f32x2{b.z, b.w}, ix, nx);
    f32x2 t0y = __builtin_elementwise_fma(f32x2{a.z, a.w}, iy, ny), t1y = __builtin_elementwise_fma(f32x2{c.x, c.y}, iy, ny);
    f32x2 t0z = __builtin_elementwise_fma(f32x2{b.x, b.y}, iz, nz), t1z = __builtin_elementwise_fma(f32x2{c.z, c.w}, iz, nz);
    tn = max2(max2(min2(t0x, t1x), min2(t0y, t1y)), max2(min2(t0z, t1z), f32x2{0.0f, 0.0f}));
    tf = min2(min2(max2(t0x, t1x), max2(t0y, t1y)), max2(t0z, t1z)) * 1.0000005f;
}

// A candidate replaces the current nearest when (t, id) is lexicographically smaller.
// While nothing is hit, best == MISS (the largest id) and best_t == +inf, so the first hit always wins.
__device__ __forceinline__ void bvh_accept(bool hit, float t, uint32_t id, uint32_t &best, float &best_t) {
    if (hit && (t < best_t || (t == best_t && id < best))) { best = id; best_t = t; }
}

// Primitives outside the tree (the `always` list), by id through the scene tables.
__device__ __forceinline__ void bvh_test_prims(const DevScene &sc, const uint32_t *ids, uint32_t n, V3 o, V3 d, uint32_t &best, float &best_t) {
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t id = ids[k];
        float t;
        bool hit;
        if (id < sc.n_sph) hit = sphere_test(sc.sph[id], o, d, t);
        else {
            uint32_t p = id - sc.n_sph;
            V3 pos;
            hit = plane_test(sc.pln[4 * p], sc.pln[4 * p + 1], sc.pln[4 * p + 2], sc.pln[4 * p + 3], o, d, t, pos);
        }
        bvh_accept(hit, t, id, best, best_t);
    }
}

// Per-lane traversal stack: entries (child reference, entry distance).  The first BVH_LDS_ENTRIES sit in LDS laid
// out [entry][thread] (a wave's pushes and pops are conflict free), deeper ones in global memory [entry][thread of
// the grid]; a kernel that traverses is launched with 256-thread blocks, at most BVH_TRAV_BLOCKS of them, and
// BVH_LDS_ENTRIES * 256 * 8 bytes of dynamic LDS.
extern __shared__ uint2 pt_lds_stack[];

struct TravStack {
    uint2 *lds;     // pt_lds_stack + threadIdx.x
    uint2 *spill;   // sc.bvh_spill + global thread index
    __device__ __forceinline__ void put(int k, uint2 e) const {
        if (k < BVH_LDS_ENTRIES) lds[k * BVH_STACK_STRIDE] = e;
        else spill[(size_t)(k - BVH_LDS_ENTRIES) * (BVH_TRAV_BLOCKS * BVH_STACK_STRIDE)] = e;
    }
    __device__ __forceinline__ uint2 get(int k) const {
        if (k < BVH_LDS_ENTRIES) return lds[k * BVH_STACK_STRIDE];
        return spill[(size_t)(k - BVH_LDS_ENTRIES) * (BVH_TRAV_BLOCKS * BVH_STACK_STRIDE)];
    }
};
__device__ __forceinline__ TravStack trav_stack(const DevScene &sc) {
    TravStack s;
    s.lds = pt_lds_stack + threadIdx.x;
    s.spill = sc.bvh_spill + ((size_t)blockIdx.x * BVH_STACK_STRIDE + threadIdx.x);
    return s;
}

// Traversal state of one ray.  node >= 0: at an inner node; node < 0 (not BVH_EMPTY): inside a leaf, primitives
// leaf_cur .. leaf_end-1 of bvh_geom still to test; node == BVH_EMPTY: finished, (best, best_t) is the answer.
struct Trav {
    V3 o, d, inv, noi;   // noi = -o * inv
    uint32_t best;
    float best_t;
    int32_t node;
    int sp;
    uint32_t leaf_cur, leaf_end;
};

__device__ __forceinline__ void trav_enter(Trav &s, int32_t ref) {
    s.node = ref;
    if (ref < 0) {
        uint32_t payload = (uint32_t)~ref;
        s.leaf_cur = payload >> 3;
        s.leaf_end = s.leaf_cur + (payload & 7u) + 1u;
    }
}

// Next subtree from the stack; entries whose entry distance is strictly beyond the current best are dropped
// without touching memory (the same rule that decides whether a child is visited at all).
__device__ __forceinline__ void trav_pop(Trav &s, const TravStack &stk) {
    for (;;) {
        if (s.sp == 0) { s.node = BVH_EMPTY; return; }
        --s.sp;
        uint2 e = stk.get(s.sp);
        if (__uint_as_float(e.y) <= s.best_t) { trav_enter(s, (int32_t)e.x); return; }
    }
}

__device__ __forceinline__ void bvh_begin(const DevScene &sc, V3 o, V3 d, Trav &s) {
    s.o = o; s.d = d;
    s.best = MISS;
    s.best_t = __builtin_inff();
    bvh_test_prims(sc, sc.bvh_always, sc.n_always, o, d, s.best, s.best_t);
    s.inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    s.noi = v3(-(o.x * s.inv.x), -(o.y * s.inv.y), -(o.z * s.inv.z));
    s.node = 0;
    s.sp = 0;
    s.leaf_cur = s.leaf_end = 0;
}

// One step of an unfinished traversal (s.node != BVH_EMPTY): one 64-byte fetch, then either the two box tests of
// an inner node (near child next, far child pushed with its entry distance) or one primitive of the current leaf.
__device__ __forceinline__ void bvh_step(const DevScene &sc, Trav &s, const TravStack &stk) {
    const bool in_leaf = s.node < 0;
    const float4 *p = in_leaf ? sc.bvh_geom + 4 * (size_t)s.leaf_cur : reinterpret_cast<const float4 *>(sc.bvh_nodes + s.node);
    float4 a = p[0], b = p[1], c = p[2], e = p[3];
    if (in_leaf) {
        uint32_t id = __float_as_uint(e.w);
        float t;
        bool hit;
        if (id < sc.n_sph) hit = sphere_test(a, s.o, s.d, t);
        else { V3 pos; hit = plane_test(a, b, c, e, s.o, s.d, t, pos); }
        bvh_accept(hit, t, id, s.best, s.best_t);
        if (++s.leaf_cur == s.leaf_end) trav_pop(s, stk);
        return;
    }
    int32_t c0 = (int32_t)__float_as_uint(e.x), c1 = (int32_t)__float_as_uint(e.y);
    f32x2 tn, tf;
    box_entry2(a, b, c, s.inv, s.noi, tn, tf);
    const float tn0 = tn.x, tn1 = tn.y;
    bool h0 = c0 != BVH_EMPTY && tn0 <= tf.x && tn0 <= s.best_t;
    bool h1 = c1 != BVH_EMPTY && tn1 <= tf.y && tn1 <= s.best_t;
    if (h0 && h1) {  // depth of the stack is bounded by the builder's depth cap
        bool first0 = tn0 <= tn1;
        stk.put(s.sp, make_uint2((uint32_t)(first0 ? c1 : c0), __float_as_uint(first0 ? tn1 : tn0)));
        ++s.sp;
        trav_enter(s, first0 ? c0 : c1);
    } else if (h0) trav_enter(s, c0);
    else if (h1) trav_enter(s, c1);
    else trav_pop(s, stk);
}

__device__ __forceinline__ uint32_t bvh_nearest(const DevScene &sc, V3 o, V3 d, float &best_t) {
    Trav s;
    bvh_begin(sc, o, d, s);
    TravStack stk = trav_stack(sc);
    while (s.node != BVH_EMPTY) bvh_step(sc, s, stk);
    best_t = s.best_t;
    return s.best;
}
#endif  // __HIPCC__

}  // namespace pt
