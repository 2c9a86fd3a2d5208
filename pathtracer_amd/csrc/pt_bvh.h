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
// (~c = first << 3 | count-1 into bvh_prims, ids ascending), BVH_EMPTY = no child.
#pragma once

#include "pt_scene.h"

namespace pt {

struct BvhNode {
    float4 a;   // (lo0.x, lo0.y, lo0.z, hi0.x)
    float4 b;   // (hi0.y, hi0.z, lo1.x, lo1.y)
    float4 c;   // (lo1.z, hi1.x, hi1.y, hi1.z)
    int32_t c0, c1;
    uint32_t pad0, pad1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

constexpr int32_t BVH_EMPTY = (int32_t)0x80000000;
constexpr int BVH_STACK = 40;          // entries per lane; the builder caps the tree depth at this
constexpr int BVH_STACK_STRIDE = 256;  // threads per block of every kernel that traverses
constexpr int BVH_MAX_LEAF = 4;

#if defined(__HIPCC__)
// Slab test.  NaN distances (0 * inf on a face of the box) drop out of fminf/fmaxf, which only removes a
// constraint; tfar is widened by 4 ulp for the rounding of the products.
__device__ __forceinline__ bool box_entry(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 inv, float &tn) {
    float t0x = (lox - o.x) * inv.x, t1x = (hix - o.x) * inv.x;
    float t0y = (loy - o.y) * inv.y, t1y = (hiy - o.y) * inv.y;
    float t0z = (loz - o.z) * inv.z, t1z = (hiz - o.z) * inv.z;
    tn = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fmaxf(fminf(t0z, t1z), 0.0f));
    float tf = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
    return tn <= tf * 1.0000005f;
}

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
        if (hit && (best == MISS || t < best_t || (t == best_t && id < best))) { best = id; best_t = t; }
    }
}
__device__ __forceinline__ void bvh_leaf(const DevScene &sc, int32_t c, V3 o, V3 d, uint32_t &best, float &best_t) {
    uint32_t payload = (uint32_t)~c;
    bvh_test_prims(sc, sc.bvh_prims + (payload >> 3), (payload & 7u) + 1u, o, d, best, best_t);
}

// Per-lane traversal stack in LDS, laid out [entry][thread] so a wave's pushes and pops are conflict free.
extern __shared__ uint32_t pt_lds_stack[];

__device__ __forceinline__ uint32_t bvh_nearest(const DevScene &sc, V3 o, V3 d, float &best_t) {
    uint32_t best = MISS;
    best_t = 0.0f;
    bvh_test_prims(sc, sc.bvh_always, sc.n_always, o, d, best, best_t);
    V3 inv = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
    uint32_t *stack = pt_lds_stack + threadIdx.x;
    int sp = 0;
    int32_t node = 0;
    for (;;) {
        const BvhNode *N = sc.bvh_nodes + node;
        float4 a = N->a, b = N->b, c = N->c;
        int32_t c0 = N->c0, c1 = N->c1;
        float tn0, tn1;
        bool h0 = c0 != BVH_EMPTY && box_entry(a.x, a.y, a.z, a.w, b.x, b.y, o, inv, tn0) && (best == MISS || tn0 <= best_t);
        bool h1 = c1 != BVH_EMPTY && box_entry(b.z, b.w, c.x, c.y, c.z, c.w, o, inv, tn1) && (best == MISS || tn1 <= best_t);
        if (h0 && c0 < 0) { bvh_leaf(sc, c0, o, d, best, best_t); h0 = false; }
        if (h1 && c1 < 0) { bvh_leaf(sc, c1, o, d, best, best_t); h1 = false; }
        // a leaf hit above may have shortened best_t; re-check the inner children against it
        if (h0 && best != MISS && tn0 > best_t) h0 = false;
        if (h1 && best != MISS && tn1 > best_t) h1 = false;
        if (h0 && h1) {  // near child first, far child on the stack (depth is bounded by the builder)
            bool first0 = tn0 <= tn1;
            stack[sp * BVH_STACK_STRIDE] = (uint32_t)(first0 ? c1 : c0);
            ++sp;
            node = first0 ? c0 : c1;
            continue;
        }
        if (h0) { node = c0; continue; }
        if (h1) { node = c1; continue; }
        if (sp == 0) break;
        --sp;
        node = (int32_t)stack[sp * BVH_STACK_STRIDE];  // its children are re-tested against the current best_t
    }
    return best;
}
#endif  // __HIPCC__

}  // namespace pt
