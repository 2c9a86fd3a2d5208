// pt_bvh.h -- BVH over the scene's spheres and finite planes: node layout and device traversal.
//
// The reference has no acceleration structure: find_scene_hit (reference src/scene.rs:263-297) scans every
// primitive and keeps the first one, in id order, with the smallest t.  The traversal below returns exactly
// that primitive:
//   * every leaf primitive is tested with the same sphere_test / plane_test the linear scan uses;
//   * a candidate replaces the current best when (t, id) is lexicographically smaller, which is what
//     "strictly closer wins, ties keep the earlier id" means and makes the result independent of visit order;
//   * boxes are padded on the host (see BvhBuilder) by far more than the rounding of the hit point and of the
//     slab test and are only ever rounded outwards, a subtree is culled only when its entry distance is strictly
//     beyond the best t, and NaN slab distances are ignored, so no primitive the linear scan would accept is ever
//     skipped.
//
// What bounds this loop is the number of load instructions a lane issues: measured on the 1M-rectangle field, the
// texture-address unit of every CU is 80-90 % busy with one slot per lane and load instruction, whatever the width of
// the load (two extra one-word loads per step cost 22 %).  A 16-byte load being the widest there is, the tree is laid
// out for the fewest of them:
//
// The tree lives in one array of 64-byte UNITS.  A unit is either a node (48 bytes used: three loads) or one leaf
// primitive (the plane's four vectors or the sphere's one, primitive id in the last word: four loads).  The up to four
// children of a node occupy consecutive units, so a node names them by the index of the first:
//   word  0..2   origin of the node's grid (x, y, z; f32)
//   word  3      byte 0..2: binary exponents ex, ey, ez of the grid's cell size (signed); byte 3: bit i (i < 4) set = child i
//                is a leaf primitive; its two top bits say which children are MISSING: 00 none (four children), 10 child
//                3, 11 children 2 and 3 (a node has 2, 3 or 4 children, 0 .. n-1).  The traversal never enters or pushes a
//                missing child, whatever the slab arithmetic says of its inside-out box (NaN distances, or a ray so far
//                away that 255 cells vanish in the rounding, read as "enter"): every reference it follows is one the
//                builder wrote, which is what keeps a walk inside the tree BY CONSTRUCTION.  (A tree whose root has fewer
//                than two children - one primitive or none - is not walked at all, see bvh_begin.)
//   word  4..9   the children's boxes on that grid, 8 bits a coordinate: words 4, 5 = x (child 0 and 1, child 2 and 3),
//                6, 7 = y, 8, 9 = z; a child's 16 bits are lo | hi << 8.  A child that does not exist has lo = 255,
//                hi = 0: no ray that bvh_begin lets into the tree enters it
//   word  10     unit index of child 0
// A box coordinate is origin + q * 2^e; the slab distance (origin + q * 2^e - o) / d is evaluated as
// q * (2^e / d) + (origin / d - o / d): one conversion and one fused multiply-add per plane after six operations per
// node.  Lower corners are rounded down and upper ones up on the host, so a quantised box contains the padded box.
// A reference to a unit: its index, with bit 31 set when the unit is a leaf primitive.  BVH_EMPTY (bit 31 alone: unit
// 0 is the root and never a leaf) marks a finished traversal.
#pragma once

#include "pt_scene.h"

namespace pt {

constexpr int32_t BVH_EMPTY = (int32_t)0x80000000;
constexpr uint32_t BVH_MAX_UNITS = 1u << 26;  // unit index << 6 is a 32-bit byte offset
constexpr int BVH_BINARY_DEPTH = 38;   // depth cap of the binary tree the builder starts from
constexpr int BVH_STACK = 64;          // entries per lane: a step pushes up to 3 per level of the 4-wide tree, whose depth the builder
                                       // keeps at or below BVH_STACK / 3 (BvhBuilder::build)
#ifndef PT_BVH_LDS_ENTRIES
#define PT_BVH_LDS_ENTRIES 10
#endif
constexpr int BVH_LDS_ENTRIES = PT_BVH_LDS_ENTRIES;    // of which this many live in LDS, the rest (rarely reached) in a global spill area;
                                                       // 20 KiB per block: seven blocks per CU fit (measured on the 1M-rectangle
                                                       // field: 8 entries 7 580, 9: 7 610, 10: 7 650, 11: 7 685, 12 - six blocks - 7 370 Mrays/s)
constexpr int BVH_STACK_STRIDE = 256;  // threads per block of every kernel that traverses
constexpr int BVH_MAX_LEAF = 1;        // primitives per leaf; measured on the 1M-rectangle field with 128-byte nodes:
                                       // 1 -> 3750, 2 -> 3650, 3 -> 3560, 4 -> 3440, 8 -> 3140 Mrays/s
constexpr int BVH_MAX_EXPONENT = 62;   // primitives reaching beyond 2^61 stay outside the tree: with the reciprocal direction capped at 2^64 and
                                       // cells of at most 2^55, the slab products of everything inside stay finite
constexpr int BVH_TAME_REACH_LOG2 = 4;           // tame origins lie within 2^4 scene sizes of the centre (BvhBuilder::tame_reach)
constexpr uint32_t BVH_TRAV_BLOCKS = 2048;  // most blocks a traversing kernel is launched with (sizes the spill area)

#if defined(__HIPCC__)
// Everything from here to bvh_nearest is compiled for the host as well (the same statements, with the two hardware
// instructions they use spelled out): that is how the traversal is run, on the CPU and with every unit index and stack
// slot checked, over the rays of whole renders (ptgpu_bvh_host_trace).
struct BvhHostCheck {
    const uint32_t *unit_prim = nullptr;   // per unit: primitive id of a leaf unit, MISS for a node (BvhBuilder::unit_prim)
    uint64_t bad_units = 0, bad_kinds = 0, bad_slots = 0;
    bool ungated = false;   // test hook: bvh_begin lets EVERY ray into the tree (answers may then be wrong; references must not be)
};
inline BvhHostCheck &bvh_host_check() { static thread_local BvhHostCheck c; return c; }   // (host code only)

// 1/d for the slab test, with its magnitude capped at 2^64.  An infinite reciprocal (a direction component that is
// zero, as in the centre column of an un-jittered image, or denormal) would make lo * inv - o * inv the difference of
// two infinities whenever lo and o have the same sign; the NaN would drop out of fminf/fmaxf and leave the slab's
// other end (an infinity of the wrong sign) as both its entry and its exit: the box, and everything in it, culled.
// With a finite power of two the products are exact, lo * inv + noi has the sign of lo - o, a ray inside the slab
// gets (-huge, +huge) and a ray beside it is culled or not according to that sign.  Capping only ever shortens a
// distance without changing its sign, which can keep a box that could have been culled, never the reverse (for
// coordinates below 2^63; the cap also covers a NaN direction, whose primitive tests all fail anyway).
// The reciprocal itself is the hardware's approximation (1 ulp): the slab test only has to be conservative, and an
// error of 2^-23 of a distance is far inside the builder's padding.
PT_HD float slab_reciprocal(float d) {
#if defined(__HIP_DEVICE_COMPILE__)
    float r = __builtin_amdgcn_rcpf(d);
#else
    float r = 1.0f / d;
#endif
    return __builtin_fabsf(r) <= 0x1p64f ? r : __builtin_copysignf(0x1p64f, d);
}

// A candidate replaces the current nearest when (t, id) is lexicographically smaller.
// While nothing is hit, best == MISS (the largest id) and best_t == +inf, so the first hit always wins.
// `slot`: where the primitive's shading row is (the index of its unit, or behind all units for the `always` list).
//
// That minimum is what the reference's scan returns as long as every accepted distance is a number.  A NaN distance is
// accepted by the reference's plane test (its conditions are written so that a NaN passes: a plane whose origin is NaN,
// a ray whose origin is) and then follows the scan's own rule - "replace only when strictly closer": the first hit in
// id order stays for good when its distance is NaN, a later NaN never replaces anything.  No order-free rule gives
// that, so an accepted NaN marks the ray (BVH_RESCAN in best_out) and the scan itself is repeated for it (bvh_finish,
// k_rescan).  Only the primitives OUTSIDE the tree can do that to a ray that is let into the tree (bvh_begin): inside
// it every coordinate is finite and below 2^61, the ray's origin too, its direction of ordinary size, and then a
// sphere's distance is a sum of two finite terms and a plane's a quotient of a finite numerator and a denominator of at
// least 1e-7 - so the walk itself uses the plain rule (the host build counts a NaN there as a fault).
constexpr int BVH_RESCAN = 2;
PT_HD void bvh_accept(bool hit, float t, uint32_t id, int out, uint32_t slot, uint32_t &best, float &best_t, int &best_out,
                                           uint32_t &best_slot) {
    if (hit && (t < best_t || (t == best_t && id < best))) { best = id; best_t = t; best_out = out; best_slot = slot; }
}
PT_HD void bvh_accept_outside(bool hit, float t, uint32_t id, int out, uint32_t slot, uint32_t &best, float &best_t, int &best_out,
                                                   uint32_t &best_slot) {
    if (hit) {
        if (t < best_t || (t == best_t && id < best)) { best = id; best_t = t; best_out = out | (best_out & BVH_RESCAN); best_slot = slot; }
        else if (t != t) best_out |= BVH_RESCAN;
    }
}

// Primitives outside the tree (the `always` list), by id through the scene tables.
PT_HD void bvh_test_prims(const DevScene &sc, const uint32_t *ids, uint32_t n, V3 o, V3 d, uint32_t &best, float &best_t,
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
        bvh_accept_outside(hit, t, id, out, sc.n_bvh_units + k, best, best_t, best_out, best_slot);
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
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) unsigned long long lds_u64;
#else
typedef unsigned long long lds_u64;   // (host: an ordinary array of BVH_LDS_ENTRIES * BVH_STACK_STRIDE words)
#endif
PT_HD unsigned long long stack_entry(int32_t ref, float dist) {
    return (unsigned long long)(uint32_t)ref | ((unsigned long long)f2u(dist) << 32);
}

struct TravStack {
    lds_u64 *lds;                // pt_lds_stack + threadIdx.x
    unsigned long long *spill;   // sc.bvh_spill + global thread index
    PT_HD void put(int k, unsigned long long e) const {
#if !defined(__HIP_DEVICE_COMPILE__)
        if (k < 0 || k >= BVH_STACK) ++bvh_host_check().bad_slots;
#endif
        if (k < BVH_LDS_ENTRIES) lds[k * BVH_STACK_STRIDE] = e;
        else {
            // the builder bounds the depth so that k < BVH_STACK always; the clamp keeps a tree that broke that promise
            // from writing outside the spill area (its image would be wrong, the process unharmed)
            k = k < BVH_STACK ? k : BVH_STACK - 1;
            spill[(size_t)(k - BVH_LDS_ENTRIES) * (BVH_TRAV_BLOCKS * BVH_STACK_STRIDE)] = e;
        }
    }
    PT_HD unsigned long long get(int k) const {
#if !defined(__HIP_DEVICE_COMPILE__)
        if (k < 0 || k >= BVH_STACK) ++bvh_host_check().bad_slots;
#endif
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

// Traversal state of one ray.  node >= 0: at an inner node (its unit index); node < 0 (not BVH_EMPTY): at a leaf
// primitive (unit index | bit 31); node == BVH_EMPTY: finished, (best, best_t) is the answer.
struct Trav {
    V3 o, d, inv, noi;   // noi = -o * inv
    uint32_t selx, sely, selz;  // byte selectors that bring a box word into (near, far, near, far) order for this ray's direction
    uint32_t best;
    float best_t;
    int best_out;        // Transition of the best hit (1: a sphere left from inside)
    uint32_t best_slot;  // its row in sc.shade_rows
    int32_t node;
    int sp;
};

PT_HD void trav_enter(Trav &s, int32_t ref) { s.node = ref; }

// Next subtree from the stack; entries whose entry distance is strictly beyond the current best are dropped
// without touching memory (the same rule that decides whether a child is visited at all).
PT_HD void trav_pop(Trav &s, const TravStack &stk) {
    for (;;) {
        if (s.sp == 0) { s.node = BVH_EMPTY; return; }
        --s.sp;
        unsigned long long e = stk.get(s.sp);
        if (u2f((uint32_t)(e >> 32)) <= s.best_t) { trav_enter(s, (int32_t)(uint32_t)e); return; }
    }
}

PT_HD void bvh_begin(const DevScene &sc, V3 o, V3 d, Trav &s) {
    s.o = o; s.d = d;
    s.best = MISS;
    s.best_t = __builtin_inff();
    s.best_out = 0;
    s.best_slot = MISS;
    bvh_test_prims(sc, sc.bvh_always, sc.n_always, o, d, s.best, s.best_t, s.best_out, s.best_slot);
    s.inv = v3(slab_reciprocal(d.x), slab_reciprocal(d.y), slab_reciprocal(d.z));
    s.noi = v3(-(o.x * s.inv.x), -(o.y * s.inv.y), -(o.z * s.inv.z));
    // a box word holds (lo, hi) of two children; a ray that runs towards smaller coordinates enters through hi
    const uint32_t KEEP = 0x03020100u, SWAP = 0x02030001u;
    s.selx = s.inv.x < 0.0f ? SWAP : KEEP;
    s.sely = s.inv.y < 0.0f ? SWAP : KEEP;
    s.selz = s.inv.z < 0.0f ? SWAP : KEEP;
    s.sp = 0;
    // Which rays may enter the tree.  A child that does not exist is kept out by its inside-out box alone - entry = exit
    // + 255 cells on every axis - as long as that difference survives the rounding of the slab distances.  The
    // builder's cells are at least 2^-23 of the scene's size M, so it does, with a factor 3 to spare, for a TAME ray:
    // origin within 16 M of the centre (taken as the sum of its coordinates: every ray that starts on a primitive inside
    // the tree, or at the camera) and a direction of ordinary size.  Any other ray - a NaN origin (the bounce off a
    // sphere of radius zero hit in its centre: its normal is 0/0), an origin on a primitive far outside the tree, an
    // infinite direction - computes equal or NaN distances for entry and exit, which the comparisons of node_keys read
    // as "enter", and what follows a node's last child in memory is not a child of it: on a GPU, a memory fault.  Nor
    // is the padding of the boxes, 2^-16 M, enough for what the primitive tests' own rounding accepts of a ray that
    // starts 2^8 M away or further.  Such a ray is left to the scan over all primitives (BVH_RESCAN), and so is one
    // that has already met a NaN distance among the primitives outside the tree: the walk ends before it starts.
    // (sums, not maxima: a NaN or an infinity in any coordinate must fail the comparison)
    const float ao = (__builtin_fabsf(o.x) + __builtin_fabsf(o.y)) + __builtin_fabsf(o.z), ad = (__builtin_fabsf(d.x) + __builtin_fabsf(d.y)) + __builtin_fabsf(d.z);
    bool tame = ao <= sc.bvh_reach && ad <= 2.0f;
#if !defined(__HIP_DEVICE_COMPILE__)
    if (bvh_host_check().ungated) tame = true;
#endif
    if (!tame) s.best_out |= BVH_RESCAN;
    // (a tree of fewer than three units - nothing or one primitive to bound - has a root with fewer than two children,
    // which node_keys does not expect: the scan over that one primitive is the whole answer)
    if (sc.n_bvh_units < 3u) s.best_out |= BVH_RESCAN;
    s.node = (s.best_out & BVH_RESCAN) ? BVH_EMPTY : 0;
}

// One leaf primitive (the four vectors of its unit) against the ray.
PT_HD void bvh_test_record(const DevScene &sc, float4 a, float4 b, float4 c, float4 e, uint32_t slot, Trav &s) {
    uint32_t id = f2u(e.w);
    float t;
    bool hit;
    int out = 0;
    if (id < sc.n_sph) hit = sphere_test(a, s.o, s.d, t, out);
    else { V3 pos; hit = plane_test(a, b, c, e, s.o, s.d, t, pos); }
#if !defined(__HIP_DEVICE_COMPILE__)
    if (hit && t != t) ++bvh_host_check().bad_kinds;
#endif
    bvh_accept(hit, t, id, out, slot, s.best, s.best_t, s.best_out, s.best_slot);
}

// Entry and exit distance of two children along one axis: w = their (lo, hi) bytes, A = cell size / d, B = (origin - o) / d.
PT_HD void slab_pair(uint32_t w, uint32_t sel, float A, float B, float &n0, float &f0, float &n1, float &f1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint32_t v = __builtin_amdgcn_perm(w, w, sel);   // (the byte extractions below are one v_cvt_f32_ubyteN each)
#else
    uint32_t v = 0;   // v_perm_b32 with both sources w: byte k of the result is byte (selector k mod 4) of w
    for (int k = 0; k < 4; ++k) v |= ((w >> (8 * ((sel >> (8 * k)) & 3u))) & 0xffu) << (8 * k);
#endif
    n0 = __builtin_fmaf((float)((v >> 0) & 0xffu), A, B);
    f0 = __builtin_fmaf((float)((v >> 8) & 0xffu), A, B);
    n1 = __builtin_fmaf((float)((v >> 16) & 0xffu), A, B);
    f1 = __builtin_fmaf((float)((v >> 24) & 0xffu), A, B);
}

// The four keys of a node for a ray: key = entry distance of a child the ray must visit, +inf otherwise.  A child that
// does not exist (only children 2 and 3 can be missing: the two top bits of the meta word, read as one unsigned comparison
// each) always gets +inf: its inside-out box alone keeps out only the rays whose slab distances are numbers.
PT_HD void node_keys(const Trav &s, float4 q0, float4 q1, float4 q2, float &k0, float &k1, float &k2, float &k3) {
    const uint32_t meta = f2u(q0.w);
    const float Ax = __builtin_ldexpf(s.inv.x, (int)(int8_t)(meta & 0xffu)), Bx = __builtin_fmaf(q0.x, s.inv.x, s.noi.x);
    const float Ay = __builtin_ldexpf(s.inv.y, (int)(int8_t)((meta >> 8) & 0xffu)), By = __builtin_fmaf(q0.y, s.inv.y, s.noi.y);
    const float Az = __builtin_ldexpf(s.inv.z, (int)(int8_t)((meta >> 16) & 0xffu)), Bz = __builtin_fmaf(q0.z, s.inv.z, s.noi.z);
    float nx0, fx0, nx1, fx1, nx2, fx2, nx3, fx3, ny0, fy0, ny1, fy1, ny2, fy2, ny3, fy3, nz0, fz0, nz1, fz1, nz2, fz2, nz3, fz3;
    slab_pair(f2u(q1.x), s.selx, Ax, Bx, nx0, fx0, nx1, fx1);
    slab_pair(f2u(q1.y), s.selx, Ax, Bx, nx2, fx2, nx3, fx3);
    slab_pair(f2u(q1.z), s.sely, Ay, By, ny0, fy0, ny1, fy1);
    slab_pair(f2u(q1.w), s.sely, Ay, By, ny2, fy2, ny3, fy3);
    slab_pair(f2u(q2.x), s.selz, Az, Bz, nz0, fz0, nz1, fz1);
    slab_pair(f2u(q2.y), s.selz, Az, Bz, nz2, fz2, nz3, fz3);
    // (NaN distances - a NaN direction - drop out of fminf/fmaxf; the far distance is widened for the rounding of the products)
    const float WIDEN = 1.0000005f, INF = __builtin_inff();
    const float tn0 = fmaxf(fmaxf(nx0, ny0), fmaxf(nz0, 0.0f)), tf0 = fminf(fminf(fx0, fy0), fz0) * WIDEN;
    const float tn1 = fmaxf(fmaxf(nx1, ny1), fmaxf(nz1, 0.0f)), tf1 = fminf(fminf(fx1, fy1), fz1) * WIDEN;
    const float tn2 = fmaxf(fmaxf(nx2, ny2), fmaxf(nz2, 0.0f)), tf2 = fminf(fminf(fx2, fy2), fz2) * WIDEN;
    const float tn3 = fmaxf(fmaxf(nx3, ny3), fmaxf(nz3, 0.0f)), tf3 = fminf(fminf(fx3, fy3), fz3) * WIDEN;
    k0 = tn0 <= fminf(tf0, s.best_t) ? tn0 : INF;
    k1 = tn1 <= fminf(tf1, s.best_t) ? tn1 : INF;
    k2 = (meta < 0xC0000000u && tn2 <= fminf(tf2, s.best_t)) ? tn2 : INF;
    k3 = (meta < 0x80000000u && tn3 <= fminf(tf3, s.best_t)) ? tn3 : INF;
}

// One step of an unfinished traversal (s.node != BVH_EMPTY): one fetch of the unit the lane stands on, then either the
// four box tests of a node (nearest hit child next, the other hit children pushed with their entry distances) or the
// test of a leaf primitive.  Returns the number of units it fetched (1).
// `leaf_turn` (the same for every lane of the wave): whether lanes that stand on a leaf primitive take their step now.
// The primitive test is executed by the whole wave whenever a single lane needs it; holding such lanes back until a
// turn comes gathers them, so that the test code runs in fewer steps with more lanes each.
PT_HD uint32_t bvh_step(const DevScene &sc, Trav &s, const TravStack &stk, bool leaf_turn = true) {
    const bool in_leaf = s.node < 0;
    if (in_leaf && !leaf_turn) return 0u;
#if !defined(__HIP_DEVICE_COMPILE__)
    {   // a reference to a unit that does not exist, or of the other kind: counted, and the walk ends here
        BvhHostCheck &chk = bvh_host_check();
        const uint32_t unit = (uint32_t)s.node & 0x7fffffffu;
        const bool outside = unit >= sc.n_bvh_units;
        if (outside) ++chk.bad_units;
        else if (chk.unit_prim && (chk.unit_prim[unit] != MISS) != in_leaf) ++chk.bad_kinds;
        if (outside) { s.node = BVH_EMPTY; return 0u; }
    }
#endif
    // (the shift drops bit 31; a 32-bit offset from a base all lanes share keeps the base in scalar registers).  The
    // offset is clamped to the last unit: with child counts in the nodes no reference outside the tree can arise, and
    // should one arise all the same (a damaged image) the fetch still stays inside the allocation.
    const uint32_t last_unit = (sc.n_bvh_units - 1u) << 6, at = (uint32_t)s.node << 6;
    const float4 *p = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(sc.bvh_nodes) + (size_t)(at < last_unit ? at : last_unit));
    float4 q0 = p[0], q1 = p[1], q2 = p[2];
    float4 q3;
    q3.x = 0.0f;   // (named by the statement below for every lane)
    if (in_leaf) q3 = p[3];
    // The whole unit is requested here, in one round trip.  Without this (empty) statement that names the last vector,
    // the compiler moves its load down into the leaf arm, behind the box tests: a second round trip per step.
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(q3.x));
#endif
    bool pop = false;    // this subtree is done: take the next one from the stack (one site for both kinds of step)
    if (!in_leaf) {
        const uint32_t meta = f2u(q0.w);
        float k0, k1, k2, k3;
        node_keys(s, q0, q1, q2, k0, k1, k2, k3);
        const float INF = __builtin_inff();
        // (a child entered at distance +inf cannot hold a finite hit, so treating it as missed loses nothing)
        int j = 0;
        float kj = k0;
        if (k1 < kj) { kj = k1; j = 1; }
        if (k2 < kj) { kj = k2; j = 2; }
        if (k3 < kj) { kj = k3; j = 3; }
        pop = kj == INF;
        if (!pop) {
            const uint32_t base = f2u(q2.z);
            const uint32_t leaf = meta >> 24;   // bit i (i < 4): child i is a leaf primitive (the bits above: missing children)
            const int32_t c0 = (int32_t)((base + 0u) | ((leaf & 1u) << 31)), c1 = (int32_t)((base + 1u) | ((leaf & 2u) << 30));
            const int32_t c2 = (int32_t)((base + 2u) | ((leaf & 4u) << 29)), c3 = (int32_t)((base + 3u) | ((leaf & 8u) << 28));
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
        bvh_test_record(sc, q0, q1, q2, q3, (uint32_t)s.node & 0x7fffffffu, s);
        pop = true;
    }
    if (pop) trav_pop(s, stk);
    return 1u;
}

// The end of every traversal (s.node == BVH_EMPTY): the answer stands, unless the ray was marked for the reference's own
// scan over every primitive in id order (bvh_accept, bvh_begin), which then replaces it.  Such rays come from degenerate
// data only - a NaN or infinite coordinate in a primitive or a ray - so the cost of the scan does not matter.
PT_HD void bvh_finish(const DevScene &sc, Trav &s) {
    if (__builtin_expect((s.best_out & BVH_RESCAN) != 0, 0)) {
        float t;
        int out;
        const uint32_t id = nearest_brute<false>(sc.sph, sc.n_sph, sc.pln, sc.n_pl, s.o, s.d, t, out);
        s.best = id;
        s.best_t = id == MISS ? __builtin_inff() : t;
        s.best_out = out;
        s.best_slot = id == MISS ? MISS : sc.prim_rows[id];
    }
}

// Is the terminal light sample at distance `light_distance` along (o, d) lit?  The reference asks for the NEAREST hit and
// then tests `hit.t > light_distance - 1e-6` (src/tracer.rs:482-488; no hit at all: black).  The nearest distance is at
// most any accepted distance, so the first accepted hit at or below that limit already decides "black" and the walk
// stops there (a nearest hit at distance NaN fails the test as well); otherwise it runs to the end and the answer is
// "some primitive was hit" (all of them beyond the limit) - or LIGHT_RESCAN: only the scan over all primitives can tell.
// Returns 0 (black), 1 (lit) or LIGHT_RESCAN.
constexpr int LIGHT_RESCAN = 2;
PT_HD int bvh_light_visible(const DevScene &sc, const TravStack &stk, V3 o, V3 d, float light_distance) {
    const float limit = light_distance - 0.000001f;
    Trav s;
    bvh_begin(sc, o, d, s);
    while (s.node != BVH_EMPTY) {
        if (s.best_t <= limit) return 0;
        (void)bvh_step(sc, s, stk);
    }
    if (s.best_t <= limit) return 0;
    if (s.best_out & BVH_RESCAN) return LIGHT_RESCAN;
    return s.best != MISS && s.best_t > limit ? 1 : 0;
}

PT_HD uint32_t bvh_nearest(const DevScene &sc, const TravStack &stk, V3 o, V3 d, float &best_t, int &best_out, uint32_t *best_slot = nullptr) {
    Trav s;
    bvh_begin(sc, o, d, s);
    while (s.node != BVH_EMPTY) (void)bvh_step(sc, s, stk);
    bvh_finish(sc, s);
    best_t = s.best_t;
    best_out = s.best_out;
    if (best_slot) *best_slot = s.best_slot;
    return s.best;
}
__device__ __forceinline__ int bvh_light_visible(const DevScene &sc, V3 o, V3 d, float light_distance) { return bvh_light_visible(sc, trav_stack(sc), o, d, light_distance); }
__device__ __forceinline__ uint32_t bvh_nearest(const DevScene &sc, V3 o, V3 d, float &best_t, int &best_out) { return bvh_nearest(sc, trav_stack(sc), o, d, best_t, best_out); }
#endif  // __HIPCC__

}  // namespace pt
