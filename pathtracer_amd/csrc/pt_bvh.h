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
// Node = 128 bytes (one cache line): the boxes of up to FOUR children, component-wise (lo.x of all four, lo.y of
// all four, ...), so one fetch decides four subtrees and the slab tests run as 2-wide packed f32 operations.
// Children: >= 0 inner node index, < 0 leaf (~c = first << 3 | count-1 into bvh_geom, ids ascending),
// BVH_EMPTY = no child.  The tree is the binary SAH tree with every other level folded into its parent.
//
// Leaf primitives live in their own array in leaf order (bvh_geom, 64 bytes each: the plane's four vectors or the
// sphere's one, primitive id in the last word), so a leaf test is ONE dependent fetch after the node.  The
// traversal step below loads up to 128 bytes from "wherever this lane is" - an inner node or the next two
// primitives of its leaf - and then either descends or tests: every lane of a wave makes progress on every memory
// round trip, which is what bounds this loop (measured on the 2-wide version: the texture/L1 path ~40 % busy, waves
// parked on memory 56 % of the time; halving the number of dependent fetches per ray is what the 4-wide node is for).
#pragma once

#include "pt_scene.h"

namespace pt {

struct BvhNode {
    float4 lox, loy, loz, hix, hiy, hiz;   // .x = child 0 ... .w = child 3
    int32_t c[4];
    uint32_t pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be 128 bytes");

constexpr int32_t BVH_EMPTY = (int32_t)0x80000000;
constexpr int BVH_BINARY_DEPTH = 38;   // depth cap of the binary tree the builder starts from
constexpr int BVH_STACK = 64;          // entries per lane: a step pushes up to 3 per level of the 4-wide tree, whose depth the builder
                                       // keeps at or below BVH_STACK / 3 (BvhBuilder::build)
#ifndef PT_BVH_LDS_ENTRIES
#define PT_BVH_LDS_ENTRIES 8
#endif
constexpr int BVH_LDS_ENTRIES = PT_BVH_LDS_ENTRIES;    // of which this many live in LDS, the rest (rarely reached) in a global spill area
constexpr int BVH_STACK_STRIDE = 256;  // threads per block of every kernel that traverses
#ifndef PT_BVH_MAX_LEAF
#define PT_BVH_MAX_LEAF 1
#endif
constexpr int BVH_MAX_LEAF = PT_BVH_MAX_LEAF;  // primitives per leaf (at most 8: three bits of the leaf reference); measured on the
                                               // 1M-rectangle field: 1 -> 3750, 2 -> 3650, 3 -> 3560, 4 -> 3440, 8 -> 3140 Mrays/s
constexpr uint32_t BVH_TRAV_BLOCKS = 2048;  // most blocks a traversing kernel is launched with (sizes the spill area)

#if defined(__HIPCC__)
// Slab test of two children at once (called twice per node).  Distances are lo * inv + (-o * inv) in one fused
// operation per plane pair (v_pk_fma_f32); against (lo - o) * inv that moves a slab by at most ~|o| * 2^-23, far
// inside the builder's padding.  inv is finite (slab_reciprocal); a NaN distance (0 * inf cannot occur any more, but a
// NaN ray can) drops out of fminf/fmaxf; tfar is widened by 4 ulp for the rounding of the products.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 min2(f32x2 x, f32x2 y) { return f32x2{fminf(x.x, y.x), fminf(x.y, y.y)}; }
__device__ __forceinline__ f32x2 max2(f32x2 x, f32x2 y) { return f32x2{fmaxf(x.x, y.x), fmaxf(x.y, y.y)}; }
__device__ __forceinline__ void box_entry2(f32x2 lox, f32x2 loy, f32x2 loz, f32x2 hix, f32x2 hiy, f32x2 hiz, V3 inv, V3 noi, f32x2 &tn,
                                           f32x2 &tf) {
    f32x2 ix = {inv.x, inv.x}, iy = {inv.y, inv.y}, iz = {inv.z, inv.z};
    f32x2 nx = {noi.x, noi.x}, ny = {noi.y, noi.y}, nz = {noi.z, noi.z};
    f32x2 t0x = __builtin_elementwise_fma(lox, ix, nx), t1x = __builtin_elementwise_fma(hix, ix, nx);
    f32x2 t0y = __builtin_elementwise_fma(loy, iy, ny), t1y = __builtin_elementwise_fma(hiy, iy, ny);
    f32x2 t0z = __builtin_elementwise_fma(loz, iz, nz), t1z = __builtin_elementwise_fma(hiz, iz, nz);
    tn = max2(max2(min2(t0x, t1x), min2(t0y, t1y)), max2(min2(t0z, t1z), f32x2{0.0f, 0.0f}));
    tf = min2(min2(max2(t0x, t1x), max2(t0y, t1y)), max2(t0z, t1z)) * 1.0000005f;
}

// 1/d for the slab test, with its magnitude capped at 2^64.  An infinite reciprocal (a direction component that is
// zero, as in the centre column of an un-jittered image, or denormal) would make lo * inv - o * inv the difference of
// two infinities whenever lo and o have the same sign; the NaN would drop out of fminf/fmaxf and leave the slab's
// other end (an infinity of the wrong sign) as both its entry and its exit: the box, and everything in it, culled.
// With a finite power of two the products are exact, lo * inv + noi has the sign of lo - o, a ray inside the slab
// gets (-huge, +huge) and a ray beside it is culled or not according to that sign.  Capping only ever shortens a
// distance without changing its sign, which can keep a box that could have been culled, never the reverse (for
// coordinates below 2^63; the cap also covers a NaN direction, whose primitive tests all fail anyway).
__device__ __forceinline__ float slab_reciprocal(float d) {
    float r = 1.0f / d;
    return __builtin_fabsf(r) <= 0x1p64f ? r : __builtin_copysignf(0x1p64f, d);
}

// A candidate replaces the current nearest when (t, id) is lexicographically smaller.
// While nothing is hit, best == MISS (the largest id) and best_t == +inf, so the first hit always wins.
// `slot`: where the primitive's shading row is (its position in bvh_geom, or behind them for the `always` list).
__device__ __forceinline__ void bvh_accept(bool hit, float t, uint32_t id, int out, uint32_t slot, uint32_t &best, float &best_t, int &best_out,
                                           uint32_t &best_slot) {
    if (hit && (t < best_t || (t == best_t && id < best))) { best = id; best_t = t; best_out = out; best_slot = slot; }
}

// Primitives outside the tree (the `always` list), by id through the scene tables.
__device__ __forceinline__ void bvh_test_prims(const DevScene &sc, const uint32_t *ids, uint32_t n, V3 o, V3 d, uint32_t &best, float &best_t,
                                               int &best_out, uint32_t &best_slot) {
    for (uint32_t k = 0; k < n; ++k) {
        uint32_t id = ids[k];
        float t;
        bool hit;
        int out = 0;
        if (id < sc.n_sph) hit = sphere_test(sc.sph[id], o, d, t, out);
        else {
            uint32_t p = id - sc.n_sph;
            V3 pos;
            hit = plane_test(sc.pln[4 * p], sc.pln[4 * p + 1], sc.pln[4 * p + 2], sc.pln[4 * p + 3], o, d, t, pos);
        }
        bvh_accept(hit, t, id, out, sc.n_leaf_prims + k, best, best_t, best_out, best_slot);
    }
}

// Per-lane traversal stack: entries (child reference, entry distance).  The first BVH_LDS_ENTRIES sit in LDS laid
// out [entry][thread] (a wave's pushes and pops are conflict free), deeper ones in global memory [entry][thread of
// the grid]; a kernel that traverses is launched with 256-thread blocks, at most BVH_TRAV_BLOCKS of them, and
// BVH_LDS_ENTRIES * 256 * 8 bytes of dynamic LDS.
extern __shared__ uint2 pt_lds_stack[];

// An entry as one 64-bit word: child reference in the low half, entry distance (f32 bits) in the high half.  The LDS
// part is addressed through a pointer that names its address space, so that its accesses stay LDS instructions: with
// two generic pointers the compiler merges the two arms of put/get into one flat access through a selected pointer.
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
__device__ __forceinline__ unsigned long long stack_entry(int32_t ref, float dist) {
    return (unsigned long long)(uint32_t)ref | ((unsigned long long)__float_as_uint(dist) << 32);
}

struct TravStack {
    lds_u64 *lds;                // pt_lds_stack + threadIdx.x
    unsigned long long *spill;   // sc.bvh_spill + global thread index
    __device__ __forceinline__ void put(int k, unsigned long long e) const {
        if (k < BVH_LDS_ENTRIES) lds[k * BVH_STACK_STRIDE] = e;
        else {
            // the builder bounds the depth so that k < BVH_STACK always; the clamp keeps a tree that broke that promise
            // from writing outside the spill area (its image would be wrong, the process unharmed)
            k = k < BVH_STACK ? k : BVH_STACK - 1;
            spill[(size_t)(k - BVH_LDS_ENTRIES) * (BVH_TRAV_BLOCKS * BVH_STACK_STRIDE)] = e;
        }
    }
    __device__ __forceinline__ unsigned long long get(int k) const {
        if (k < BVH_LDS_ENTRIES) return lds[k * BVH_STACK_STRIDE];
        k = k < BVH_STACK ? k : BVH_STACK - 1;
        return spill[(size_t)(k - BVH_LDS_ENTRIES) * (BVH_TRAV_BLOCKS * BVH_STACK_STRIDE)];
    }
};
__device__ __forceinline__ TravStack trav_stack(const DevScene &sc) {
    TravStack s;
    s.lds = (lds_u64 *)pt_lds_stack + threadIdx.x;
    s.spill = reinterpret_cast<unsigned long long *>(sc.bvh_spill) + ((size_t)blockIdx.x * BVH_STACK_STRIDE + threadIdx.x);
    return s;
}

// Traversal state of one ray.  node >= 0: at an inner node; node < 0 (not BVH_EMPTY): inside a leaf, primitives
// leaf_cur .. leaf_end-1 of bvh_geom still to test; node == BVH_EMPTY: finished, (best, best_t) is the answer.
struct Trav {
    V3 o, d, inv, noi;   // noi = -o * inv
    uint32_t best;
    float best_t;
    int best_out;        // Transition of the best hit (1: a sphere left from inside)
    uint32_t best_slot;  // its row in sc.shade_rows
    int32_t node;
    int sp;
    uint32_t leaf_cur, leaf_end;
};

__device__ __forceinline__ void trav_enter(Trav &s, int32_t ref) {
    s.node = ref;
    if (ref < 0) {
        uint32_t payload = (uint32_t)~ref;
        s.leaf_cur = payload >> 3;
        if (BVH_MAX_LEAF > 1) s.leaf_end = s.leaf_cur + (payload & 7u) + 1u;
    }
}

// Next subtree from the stack; entries whose entry distance is strictly beyond the current best are dropped
// without touching memory (the same rule that decides whether a child is visited at all).
__device__ __forceinline__ void trav_pop(Trav &s, const TravStack &stk) {
    for (;;) {
        if (s.sp == 0) { s.node = BVH_EMPTY; return; }
        --s.sp;
        unsigned long long e = stk.get(s.sp);
        if (__uint_as_float((uint32_t)(e >> 32)) <= s.best_t) { trav_enter(s, (int32_t)(uint32_t)e); return; }
    }
}

__device__ __forceinline__ void bvh_begin(const DevScene &sc, V3 o, V3 d, Trav &s) {
    s.o = o; s.d = d;
    s.best = MISS;
    s.best_t = __builtin_inff();
    s.best_out = 0;
    s.best_slot = MISS;
    bvh_test_prims(sc, sc.bvh_always, sc.n_always, o, d, s.best, s.best_t, s.best_out, s.best_slot);
    s.inv = v3(slab_reciprocal(d.x), slab_reciprocal(d.y), slab_reciprocal(d.z));
    s.noi = v3(-(o.x * s.inv.x), -(o.y * s.inv.y), -(o.z * s.inv.z));
    s.node = 0;
    s.sp = 0;
    s.leaf_cur = s.leaf_end = 0;
}

// One primitive record of bvh_geom against the ray.
__device__ __forceinline__ void bvh_test_record(const DevScene &sc, float4 a, float4 b, float4 c, float4 e, uint32_t slot, Trav &s) {
    uint32_t id = __float_as_uint(e.w);
    float t;
    bool hit;
    int out = 0;
    if (id < sc.n_sph) hit = sphere_test(a, s.o, s.d, t, out);
    else { V3 pos; hit = plane_test(a, b, c, e, s.o, s.d, t, pos); }
    bvh_accept(hit, t, id, out, slot, s.best, s.best_t, s.best_out, s.best_slot);
}

// One step of an unfinished traversal (s.node != BVH_EMPTY): one fetch of up to 128 bytes, then either the four
// box tests of an inner node (nearest hit child next, the other hit children pushed with their entry distances) or
// up to two primitives of the current leaf.  Returns the number of 64-byte units it fetched.
__device__ __forceinline__ uint32_t bvh_step(const DevScene &sc, Trav &s, const TravStack &stk) {
    const bool in_leaf = s.node < 0;
    const float4 *p = in_leaf ? sc.bvh_geom + 4 * (size_t)s.leaf_cur : reinterpret_cast<const float4 *>(sc.bvh_nodes + s.node);
    // the second half of the line: an inner node's upper corners and references, or a leaf's second primitive
    const bool second = !in_leaf || (BVH_MAX_LEAF > 1 && s.leaf_cur + 1u < s.leaf_end);
    float4 q0 = p[0], q1 = p[1], q2 = p[2], q3 = p[3];
    float4 q4, q5, q6, q7;
    if (second) { q4 = p[4]; q5 = p[5]; q6 = p[6]; if (BVH_MAX_LEAF > 1 && in_leaf) q7 = p[7]; }
    // All of the line is requested here, in one round trip.  Without this (empty) statement that names the second half,
    // the compiler moves those loads down into the inner-node arm, behind the primitive test of the lanes that are in a
    // leaf: a second round trip per step for every lane of the wave.
    asm volatile("" : "+v"(q4.x), "+v"(q5.x), "+v"(q6.x));
    bool pop = false;    // this subtree is done: take the next one from the stack (one site for both kinds of step)
    uint32_t units = 2u;
    if (!in_leaf) {
        // q0..q5 = lox, loy, loz, hix, hiy, hiz of the four children; q6 = their references
        f32x2 tnA, tfA, tnB, tfB;
        box_entry2(f32x2{q0.x, q0.y}, f32x2{q1.x, q1.y}, f32x2{q2.x, q2.y}, f32x2{q3.x, q3.y}, f32x2{q4.x, q4.y}, f32x2{q5.x, q5.y}, s.inv,
                   s.noi, tnA, tfA);
        box_entry2(f32x2{q0.z, q0.w}, f32x2{q1.z, q1.w}, f32x2{q2.z, q2.w}, f32x2{q3.z, q3.w}, f32x2{q4.z, q4.w}, f32x2{q5.z, q5.w}, s.inv,
                   s.noi, tnB, tfB);
        const int32_t c0 = (int32_t)__float_as_uint(q6.x), c1 = (int32_t)__float_as_uint(q6.y);
        const int32_t c2 = (int32_t)__float_as_uint(q6.z), c3 = (int32_t)__float_as_uint(q6.w);
        const float INF = __builtin_inff();
        // key = entry distance of a child the ray must visit, +inf otherwise
        float k0 = (c0 != BVH_EMPTY && tnA.x <= tfA.x && tnA.x <= s.best_t) ? tnA.x : INF;
        float k1 = (c1 != BVH_EMPTY && tnA.y <= tfA.y && tnA.y <= s.best_t) ? tnA.y : INF;
        float k2 = (c2 != BVH_EMPTY && tnB.x <= tfB.x && tnB.x <= s.best_t) ? tnB.x : INF;
        float k3 = (c3 != BVH_EMPTY && tnB.y <= tfB.y && tnB.y <= s.best_t) ? tnB.y : INF;
        // (a child entered at distance +inf cannot hold a finite hit, so treating it as missed loses nothing)
        int j = 0;
        float kj = k0;
        if (k1 < kj) { kj = k1; j = 1; }
        if (k2 < kj) { kj = k2; j = 2; }
        if (k3 < kj) { kj = k3; j = 3; }
        pop = kj == INF;
        if (!pop) {
            // the others in fixed order; entries are culled by distance again when they are popped
            const bool p3 = j != 3 && k3 < INF, p2 = j != 2 && k2 < INF, p1 = j != 1 && k1 < INF, p0 = j != 0 && k0 < INF;
            const unsigned long long e3 = stack_entry(c3, k3), e2 = stack_entry(c2, k2), e1 = stack_entry(c1, k1), e0 = stack_entry(c0, k0);
            if (s.sp <= BVH_LDS_ENTRIES - 4) {
                // all four written one after the other, each over its predecessor when that one is not kept: no branches,
                // and the slots touched (sp .. sp + 3; child j is never kept) are all in LDS
                lds_u64 *w = stk.lds + s.sp * BVH_STACK_STRIDE;
                *w = e3; w += p3 ? BVH_STACK_STRIDE : 0;
                *w = e2; w += p2 ? BVH_STACK_STRIDE : 0;
                *w = e1; w += p1 ? BVH_STACK_STRIDE : 0;
                *w = e0;
                s.sp += (int)p3 + (int)p2 + (int)p1 + (int)p0;
            } else {
                if (p3) { stk.put(s.sp, e3); ++s.sp; }
                if (p2) { stk.put(s.sp, e2); ++s.sp; }
                if (p1) { stk.put(s.sp, e1); ++s.sp; }
                if (p0) { stk.put(s.sp, e0); ++s.sp; }
            }
            trav_enter(s, j == 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3)));
        }
    } else {
        bvh_test_record(sc, q0, q1, q2, q3, s.leaf_cur, s);
        if (BVH_MAX_LEAF > 1) {
            if (second) bvh_test_record(sc, q4, q5, q6, q7, s.leaf_cur + 1u, s);
            s.leaf_cur += second ? 2u : 1u;
            pop = s.leaf_cur == s.leaf_end;
            units = second ? 2u : 1u;
        } else {
            pop = true;
            units = 1u;
        }
    }
    if (pop) trav_pop(s, stk);
    return units;
}

// Is the terminal light sample at distance `light_distance` along (o, d) lit?  The reference asks for the NEAREST hit and
// then tests `hit.t > light_distance - 1e-6` (src/tracer.rs:482-488; no hit at all: black).  The nearest distance is at
// most any accepted distance, so the first accepted hit at or below that limit already decides "black" and the walk
// stops there; otherwise it runs to the end and the answer is "some primitive was hit" (all of them beyond the limit).
__device__ __forceinline__ bool bvh_light_visible(const DevScene &sc, V3 o, V3 d, float light_distance) {
    const float limit = light_distance - 0.000001f;
    Trav s;
    bvh_begin(sc, o, d, s);
    TravStack stk = trav_stack(sc);
    while (s.node != BVH_EMPTY) {
        if (s.best_t <= limit) return false;
        (void)bvh_step(sc, s, stk);
    }
    return s.best != MISS && s.best_t > limit;
}

__device__ __forceinline__ uint32_t bvh_nearest(const DevScene &sc, V3 o, V3 d, float &best_t, int &best_out) {
    Trav s;
    bvh_begin(sc, o, d, s);
    TravStack stk = trav_stack(sc);
    while (s.node != BVH_EMPTY) (void)bvh_step(sc, s, stk);
    best_t = s.best_t;
    best_out = s.best_out;
    return s.best;
}
#endif  // __HIPCC__

}  // namespace pt
