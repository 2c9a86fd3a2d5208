// pt_bvh_build.h -- host-side BVH construction (binned SAH, depth-capped) over the flattened primitives.
//
// The reference scans all primitives per ray (reference src/scene.rs:263-297); this structure exists only so
// that scan can skip primitives the ray cannot reach.  Boxes therefore bound the region in which the
// reference's OWN tests can accept a hit, not the ideal geometry:
//   * sphere (src/scene.rs:169-221): centre +- radius;
//   * plane  (src/scene.rs:223-261): the set {x in the plane : 0 <= (x-s).un <= |u|, 0 <= (x-s).vn <= |v|},
//     which is the u,v parallelogram only when u is perpendicular to v; its four corners are solved for here;
//   * every box is then grown by `pad` = 2^-16 of the largest coordinate magnitude in the scene (camera
//     included), orders of magnitude above the f32 rounding of num/denum, p + t*d and the slab test for rays
//     that start within a few scene diameters.
// Planes whose normal is not finite can never be hit (the In test fails on NaN) and are left out; planes with
// a finite normal but unusable axes (zero or parallel u, v) cannot be bounded and go to the `always` list,
// which the traversal tests for every ray.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <future>
#include <vector>

#include "pt_bvh.h"

namespace pt {

struct BvhBuilder {
    std::vector<float4> units;     // the tree as the device reads it: 4 vectors per 64-byte unit (pt_bvh.h)
    std::vector<uint32_t> unit_prim;  // per unit: the primitive id of a leaf unit, MISS_ID for a node
    uint64_t n_nodes = 0, n_leaf_prims = 0;
    static constexpr uint32_t MISS_ID = 0xffffffffu;
    std::vector<uint32_t> always;  // primitives tested for every ray
    int max_depth_seen = 0;        // depth of the binary tree (root = 0)
    int depth4 = 0;                // levels of inner nodes of the 4-wide tree: a traversal holds at most 3 * depth4 stack entries
    uint64_t forced_median_splits = 0;  // splits taken at the object median because SAH could have exceeded the depth cap
    bool level_fold = false;       // the 4-wide tree was folded level by level (see fold) to bound its depth
    int sah_bins = 0;              // bins of the candidate tree that was kept (see build)
    double expected_steps = 0.0;   // its estimate: steps of a random line through the root's box (see estimate_steps)

    struct Box {
        double lo[3], hi[3];
        void reset() { for (int k = 0; k < 3; ++k) { lo[k] = 1e300; hi[k] = -1e300; } }
        void grow(const Box &b) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
        void grow(const double p[3]) { for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], p[k]); hi[k] = std::max(hi[k], p[k]); } }
        double area() const {
            double d0 = hi[0] - lo[0], d1 = hi[1] - lo[1], d2 = hi[2] - lo[2];
            if (d0 < 0 || d1 < 0 || d2 < 0) return 0.0;
            return 2.0 * (d0 * d1 + d1 * d2 + d2 * d0);
        }
    };

    struct BuildNode { Box box; int32_t left = -1, right = -1; uint32_t first = 0, count = 0; };

    std::vector<Box> pbox;
    std::vector<double> pcent;  // 3 per primitive
    std::vector<uint32_t> idx;
    std::vector<BuildNode> bn;
    double pad = 0.0;
    double scene_size = 0.0;   // M: the largest |coordinate| of the camera and of every box in the tree; pad = M * 2^-16

    // Ray origins within this of the centre (the sum of their |coordinates|) are "tame" for the traversal (bvh_begin): every
    // cell of every node is at least 2 * pad / 255 = M * 2^-23 wide (write_node), a tame origin is at most 17 M from a
    // node's corner, so entry and exit of an inside-out box differ by 255 cells >= 2^-15 M against a slab distance of at
    // most 17 M: 1.8e-6 of it, where the widening of the exit (5e-7) and three roundings (6e-8 each) take 0.7e-6.
    // (Beyond 2^62 the slab products may overflow; such origins are not tame whatever the scene's size.)
    float tame_reach() const { return (float)std::min(std::ldexp(scene_size, BVH_TAME_REACH_LOG2), std::ldexp(1.0, 62)); }
    const std::vector<float4> *src_sph = nullptr, *src_pln = nullptr;

    static bool finite3(double a, double b, double c) { return std::isfinite(a) && std::isfinite(b) && std::isfinite(c); }
    // Beyond 2^BVH_MAX_EXPONENT the products of the device's slab test could overflow: such a primitive is tested for every ray.
    static bool too_far(double x, double y, double z, double reach) {
        const double lim = std::ldexp(1.0, BVH_MAX_EXPONENT - 1);
        return std::fabs(x) + reach > lim || std::fabs(y) + reach > lim || std::fabs(z) + reach > lim;
    }

    // The padded box of everything the tree holds: a ray that misses it misses every primitive - provided nothing is
    // on the `always` list.  Call after boxes().  False: no usable bound (unboundable primitives, or nothing at all).
    bool scene_bounds(float lo[3], float hi[3]) const {
        if (!always.empty() || idx.empty()) return false;
        Box b; b.reset();
        for (uint32_t i : idx) b.grow(pbox[i]);
        for (int d = 0; d < 3; ++d) { lo[d] = down(b.lo[d] - pad); hi[d] = up(b.hi[d] + pad); }
        return true;
    }

    void build(const std::vector<float4> &sph, const std::vector<float4> &pln, V3 cam_pos, V3 cam_plane_origin) {
        const auto t_start = std::chrono::steady_clock::now();
        auto ms_since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(); };
        const bool verbose = std::getenv("PTGPU_BVH_VERBOSE") != nullptr;   // diagnostic: timings and the candidates' estimates on stderr
        boxes(sph, pln, cam_pos, cam_plane_origin);
        if (verbose) std::fprintf(stderr, "bvh: boxes %.0f ms\n", ms_since());
        units.clear();
        unit_prim.clear();
        n_nodes = n_leaf_prims = 0;
        bn.clear();
        if (idx.empty()) {  // nothing boundable: a root without children
            const int32_t none[1] = {0};
            units.assign(4, make_float4(0.f, 0.f, 0.f, 0.f));
            unit_prim.assign(1, MISS_ID);
            write_node(0, none, 0, 0);
            n_nodes = 1;
            depth4 = 1;
            return;
        }
        // The greedy surface-area heuristic is a noisy guide here (one primitive per leaf, four children per node): which
        // resolution of the binning and which weight of a side's primitive count (the classic cost area * count assumes a
        // subtree costs in proportion to its primitives; with one primitive per leaf it grows more slowly) give the tree
        // with the fewest traversal steps differs from scene to scene - on the 1M-rectangle field 2 bins per axis give 6 %
        // fewer steps than 16 and 14 % fewer than 32, on rooms with 120-300 spheres and on random clutter 32 bins with
        // count^0.5 or count^0.75 beat every count^1 tree by 1-6 %.  So several binary trees are built (concurrently)
        // and the one whose 4-wide fold promises the fewest steps is kept; the estimate ranks them as the measured step
        // counts do (field: 12.9 / 13.0 / 13.5 / 14.3 estimated for 2 / 4 / 16 / 32 bins folded greedily, 16.8 / 17.0 /
        // 17.9 / 19.5 measured).  Any of them gives the same image.
#ifndef PT_CAND
#define PT_CAND {2, 3}, {16, 3}, {32, 3}, {32, 2}
#endif
        struct Cand { int bins, quarter_powers; };   // side weight = count^(quarter_powers / 4)
        static const Cand CANDIDATE_BINS[] = {PT_CAND};
        constexpr int NC = (int)(sizeof CANDIDATE_BINS / sizeof CANDIDATE_BINS[0]);
        std::vector<Tree> trees(NC);
        auto grow_tree = [&, this](int k) {
            Tree &t = trees[k];
            t.bins = CANDIDATE_BINS[k].bins; t.quarter_powers = CANDIDATE_BINS[k].quarter_powers;
            t.idx = idx;
            t.bn.reserve(2 * idx.size());
            const double t0 = ms_since();
            build_range(t, 0, (uint32_t)t.idx.size(), 0);
            const double t1 = ms_since();
            t.steps = estimate_steps(t.bn);
            if (verbose) std::fprintf(stderr, "bvh: %2d bins: started %.0f, built %.0f, estimated %.0f ms\n", t.bins, t0, t1, ms_since());
        };
        {
            std::vector<std::future<void>> jobs;
            int started = 0;
            try {
                for (; started < NC - 1; ++started) jobs.push_back(std::async(std::launch::async, grow_tree, started));
            } catch (const std::system_error &) {}   // no more threads: the rest is built here
            for (int k = started; k < NC; ++k) grow_tree(k);
            for (auto &j : jobs) j.get();
        }
        int keep = 0;
        for (int k = 1; k < NC; ++k) if (trees[k].steps < trees[keep].steps) keep = k;
        if (verbose)
            for (int k = 0; k < NC; ++k)
                std::fprintf(stderr, "bvh: %2d bins, count^%.2f: estimate %.3f steps, depth %d%s\n", trees[k].bins, trees[k].quarter_powers / 4.0, trees[k].steps, trees[k].max_depth, k == keep ? " (kept)" : "");
        if (verbose) std::fprintf(stderr, "bvh: trees %.0f ms\n", ms_since());
        bn = std::move(trees[keep].bn);
        idx = std::move(trees[keep].idx);
        max_depth_seen = trees[keep].max_depth;
        forced_median_splits = trees[keep].forced;
        sah_bins = trees[keep].bins;
        expected_steps = trees[keep].steps;
        trees.clear();
        plan = plan_fold(bn);
        // The traversal pushes up to three children per inner node on the way down, so its stack holds at most
        // 3 * depth4 entries.  Folding by box area (fastest to traverse) gives no bound better than the binary depth
        // cap; when such a tree would be too deep for the stack it is folded again level by level, which halves the
        // binary depth: at most (BVH_BINARY_DEPTH + 1) / 2 = 19 levels, 57 entries.
        level_fold = false;
        flatten();
        if (3 * depth4 > BVH_STACK) {
            level_fold = true;
            flatten();
        }
        if (verbose) std::fprintf(stderr, "bvh: flattened %.0f ms, %zu units, depth %d\n", ms_since(), unit_prim.size(), depth4);
    }

    // Per-primitive boxes (pbox, idx), the `always` list and the padding.
    void boxes(const std::vector<float4> &sph, const std::vector<float4> &pln, V3 cam_pos, V3 cam_plane_origin) {
        const uint32_t n_sph = (uint32_t)sph.size(), n_pl = (uint32_t)(pln.size() / 4);
        const uint32_t n = n_sph + n_pl;
        src_sph = &sph; src_pln = &pln;
        always.clear();
        pbox.resize(n);
        pcent.resize(3 * (size_t)n);
        idx.clear();
        idx.reserve(n);
        double M = 0.0;
        auto track = [&](double v) { if (std::isfinite(v)) M = std::max(M, std::fabs(v)); };
        track(cam_pos.x); track(cam_pos.y); track(cam_pos.z);
        track(cam_plane_origin.x); track(cam_plane_origin.y); track(cam_plane_origin.z);
        for (uint32_t i = 0; i < n_sph; ++i) {
            const float4 s = sph[i];
            Box &b = pbox[i];
            double r = std::fabs((double)s.w);
            if (!finite3(s.x, s.y, s.z) || !std::isfinite(r) || too_far(s.x, s.y, s.z, r)) { always.push_back(i); continue; }
            b.lo[0] = s.x - r; b.lo[1] = s.y - r; b.lo[2] = s.z - r;
            b.hi[0] = s.x + r; b.hi[1] = s.y + r; b.hi[2] = s.z + r;
            idx.push_back(i);
        }
        for (uint32_t k = 0; k < n_pl; ++k) {
            const float4 q0 = pln[4 * k], q1 = pln[4 * k + 1], q2 = pln[4 * k + 2], q3 = pln[4 * k + 3];
            const uint32_t id = n_sph + k;
            if (!finite3(q0.x, q0.y, q0.z)) continue;  // NaN normal: never accepted
            double s[3] = {q0.w, q1.x, q1.y};
            double un[3] = {q1.z, q1.w, q2.x}, vn[3] = {q2.z, q2.w, q3.x};
            double ul = q2.y, vl = q3.y;
            double c = un[0] * vn[0] + un[1] * vn[1] + un[2] * vn[2];
            bool ok = finite3(s[0], s[1], s[2]) && finite3(un[0], un[1], un[2]) && finite3(vn[0], vn[1], vn[2]) && std::isfinite(ul) &&
                      std::isfinite(vl) && std::fabs(c) < 0.999;
            ok = ok && !too_far(s[0], s[1], s[2], std::fabs(ul) + std::fabs(vl));
            if (!ok) { always.push_back(id); continue; }
            Box &b = pbox[id];
            b.reset();
            double inv = 1.0 / (1.0 - c * c);
            for (int ia = 0; ia < 2; ++ia)
                for (int ib = 0; ib < 2; ++ib) {
                    double a = ia ? ul : 0.0, bb = ib ? vl : 0.0;
                    double alpha = (a - bb * c) * inv, beta = (bb - a * c) * inv;
                    double p[3];
                    for (int d = 0; d < 3; ++d) p[d] = s[d] + alpha * un[d] + beta * vn[d];
                    b.grow(p);
                }
            idx.push_back(id);
        }
        for (uint32_t i : idx) {
            const Box &b = pbox[i];
            for (int d = 0; d < 3; ++d) {
                pcent[3 * (size_t)i + d] = 0.5 * (b.lo[d] + b.hi[d]);
                track(b.lo[d]); track(b.hi[d]);
            }
        }
        pad = std::ldexp(std::max(M, 1e-30), -16);
        scene_size = std::max(M, 1e-30);
        std::sort(always.begin(), always.end());
    }

    // One candidate binary tree (its own permutation of the primitives); `bins`: resolution of the binned SAH.
    struct Tree {
        std::vector<BuildNode> bn;
        std::vector<uint32_t> idx;
        int max_depth = 0, bins = 16, quarter_powers = 4;
        uint64_t forced = 0;
        double steps = 0.0;
    };
    static constexpr int MAX_BINS = 32;
    static double side_weight(uint32_t count, int quarter_powers) {   // count^(quarter_powers / 4) for 2, 3 and 4 quarters
        const double c = (double)count, r = std::sqrt(c);
        return quarter_powers >= 4 ? c : (quarter_powers == 3 ? r * std::sqrt(r) : r);
    }

    int32_t build_range(Tree &t, uint32_t first, uint32_t count, int depth) const {
        std::vector<BuildNode> &bn = t.bn;
        std::vector<uint32_t> &idx = t.idx;
        t.max_depth = std::max(t.max_depth, depth);
        const int32_t me = (int32_t)bn.size();
        bn.emplace_back();
        bn[me].box.reset();
        for (uint32_t i = first; i < first + count; ++i) bn[me].box.grow(pbox[idx[i]]);
        bn[me].first = first; bn[me].count = count;
        if (count <= (uint32_t)BVH_MAX_LEAF) return me;   // (one primitive: median splits by index always get there)
        // Children sit one level deeper; a leaf holds BVH_MAX_LEAF.  Force a median split when SAH imbalance
        // could otherwise exceed the traversal stack.
        const int remaining = BVH_BINARY_DEPTH - depth;
        bool force_median = remaining < 1 || (remaining < 40 && (uint64_t)count > ((uint64_t)BVH_MAX_LEAF << (remaining > 0 ? remaining - 1 : 0)));
        if (force_median) ++t.forced;
        Box cb; cb.reset();
        for (uint32_t i = first; i < first + count; ++i) cb.grow(&pcent[3 * (size_t)idx[i]]);
        int axis = 0;
        double ext = -1.0;
        for (int d = 0; d < 3; ++d) if (cb.hi[d] - cb.lo[d] > ext) { ext = cb.hi[d] - cb.lo[d]; axis = d; }
        uint32_t mid = first + count / 2;
        bool split_done = false;
        if (!force_median && ext > 0.0) {
            const int NB = t.bins;
            double best_cost = 1e300;
            int best_axis = -1, best_bin = -1;
            for (int d = 0; d < 3; ++d) {
                double lo = cb.lo[d], w = cb.hi[d] - cb.lo[d];
                if (!(w > 0.0)) continue;
                Box bb[MAX_BINS]; uint32_t bc[MAX_BINS];
                for (int b = 0; b < NB; ++b) { bb[b].reset(); bc[b] = 0; }
                double scale = NB / w;
                for (uint32_t i = first; i < first + count; ++i) {
                    uint32_t p = idx[i];
                    int b = (int)((pcent[3 * (size_t)p + d] - lo) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    bb[b].grow(pbox[p]); bc[b]++;
                }
                double right_area[MAX_BINS]; uint32_t right_cnt[MAX_BINS];
                Box acc; acc.reset(); uint32_t cnt = 0;
                for (int b = NB - 1; b >= 1; --b) { acc.grow(bb[b]); cnt += bc[b]; right_area[b] = acc.area(); right_cnt[b] = cnt; }
                acc.reset(); cnt = 0;
                for (int b = 0; b < NB - 1; ++b) {
                    acc.grow(bb[b]); cnt += bc[b];
                    if (cnt == 0 || right_cnt[b + 1] == 0) continue;
                    double cost = acc.area() * side_weight(cnt, t.quarter_powers) + right_area[b + 1] * side_weight(right_cnt[b + 1], t.quarter_powers);
                    if (cost < best_cost) { best_cost = cost; best_axis = d; best_bin = b; }
                }
            }
            if (best_axis >= 0) {
                double lo = cb.lo[best_axis], scale = NB / (cb.hi[best_axis] - cb.lo[best_axis]);
                auto it = std::partition(idx.begin() + first, idx.begin() + first + count, [&](uint32_t p) {
                    int b = (int)((pcent[3 * (size_t)p + best_axis] - lo) * scale);
                    b = b < 0 ? 0 : (b >= NB ? NB - 1 : b);
                    return b <= best_bin;
                });
                mid = (uint32_t)(it - idx.begin());
                split_done = mid > first && mid < first + count;
            }
        }
        if (!split_done) {  // object median along the widest centroid axis (or plain halves when all coincide)
            mid = first + count / 2;
            std::nth_element(idx.begin() + first, idx.begin() + mid, idx.begin() + first + count, [&](uint32_t a, uint32_t b) {
                double ca = pcent[3 * (size_t)a + axis], cb2 = pcent[3 * (size_t)b + axis];
                return ca < cb2 || (ca == cb2 && a < b);
            });
        }
        int32_t l = build_range(t, first, mid - first, depth + 1);
        int32_t r = build_range(t, mid, first + count - mid, depth + 1);
        bn[me].left = l; bn[me].right = r;
        return me;
    }

    static float down(double v) { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; }
    static float up(double v) { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; }

    // The unit of a leaf primitive: the plane's four vectors or the sphere's one, the primitive id in the last word.
    void write_leaf(uint32_t unit, const BuildNode &n) {
        const uint32_t id = idx[n.first];
        const uint32_t n_sph = (uint32_t)src_sph->size();
        float4 q[4] = {};
        if (id < n_sph) q[0] = (*src_sph)[id];
        else for (int k = 0; k < 4; ++k) q[k] = (*src_pln)[4 * (size_t)(id - n_sph) + k];
        std::memcpy(&q[3].w, &id, 4);
        std::copy(q, q + 4, units.begin() + 4 * (size_t)unit);
        unit_prim[unit] = id;
        ++n_leaf_prims;
    }

    // The unit of a node whose n children (binary nodes ch[0..n-1]) occupy the units first_child .. first_child + n - 1:
    // the children's padded boxes on a grid of 255 cells per axis that starts at the lowest padded corner; lower corners
    // rounded down, upper ones up, so every quantised box contains the padded one (layout: pt_bvh.h).
    void write_node(uint32_t unit, const int32_t *ch, int n, uint32_t first_child) {
        float lo[4][3], hi[4][3];
        for (int j = 0; j < n; ++j)
            for (int d = 0; d < 3; ++d) { lo[j][d] = down(bn[ch[j]].box.lo[d] - pad); hi[j][d] = up(bn[ch[j]].box.hi[d] + pad); }
        float org[3] = {0.f, 0.f, 0.f};
        int ex[3] = {0, 0, 0};
        uint8_t qlo[4][3], qhi[4][3];
        for (int j = 0; j < 4; ++j) for (int d = 0; d < 3; ++d) { qlo[j][d] = 255; qhi[j][d] = 0; }   // no such child: inside out
        for (int d = 0; d < 3; ++d) {
            // (a node without children - the root of a scene with nothing to bound - keeps the cell size every node has at least)
            double lo_min = n > 0 ? lo[0][d] : 0.0, hi_max = n > 0 ? hi[0][d] : 0.0;
            for (int j = 1; j < n; ++j) { lo_min = std::min<double>(lo_min, lo[j][d]); hi_max = std::max<double>(hi_max, hi[j][d]); }
            org[d] = (float)lo_min;
            const double ext = std::max(hi_max - lo_min, 2.0 * pad);   // (2 * pad at least anyway: every box is padded on both sides)
            int e = std::max(-100, std::ilogb(ext / 255.0));
            while (255.0 * std::ldexp(1.0, e) < ext) ++e;
            ex[d] = e;   // (at most BVH_MAX_EXPONENT - 7: boxes() keeps the extent below 2^BVH_MAX_EXPONENT)
            const double cell = std::ldexp(1.0, e);
            for (int j = 0; j < n; ++j) {
                double a = std::floor((lo[j][d] - lo_min) / cell), b = std::ceil((hi[j][d] - lo_min) / cell);
                while (lo_min + a * cell > lo[j][d]) a -= 1.0;
                while (lo_min + b * cell < hi[j][d]) b += 1.0;
                qlo[j][d] = (uint8_t)std::min(255.0, std::max(0.0, a));
                qhi[j][d] = (uint8_t)std::min(255.0, std::max(0.0, b));
            }
        }
        uint32_t leaf_bits = 0;
        for (int j = 0; j < n; ++j) if (bn[ch[j]].left < 0) leaf_bits |= 1u << j;
        uint32_t w[16] = {};
        std::memcpy(&w[0], org, 12);
        w[3] = (uint32_t)(uint8_t)(int8_t)ex[0] | (uint32_t)(uint8_t)(int8_t)ex[1] << 8 | (uint32_t)(uint8_t)(int8_t)ex[2] << 16 | leaf_bits << 24 | (n >= 4 ? 0u : (n == 3 ? 2u : 3u)) << 30;
        for (int d = 0; d < 3; ++d)
            for (int half = 0; half < 2; ++half) {
                const int j0 = 2 * half, j1 = 2 * half + 1;
                w[4 + 2 * d + half] = (uint32_t)qlo[j0][d] | (uint32_t)qhi[j0][d] << 8 | (uint32_t)qlo[j1][d] << 16 | (uint32_t)qhi[j1][d] << 24;
            }
        w[10] = first_child;
        std::memcpy(&units[4 * (size_t)unit], w, 64);
        unit_prim[unit] = MISS_ID;
        ++n_nodes;
    }

    // Which binary nodes become children of 4-wide nodes.  A binary node that is folded away costs a traversal nothing;
    // one that becomes a child costs a step whenever a ray meets its box, i.e. in proportion to its area.  The plan
    // minimises the sum of the areas of all children by dynamic programming over the binary tree: c(b, k) = cheapest way
    // to hang subtree b under a 4-wide node with at most k of its four slots = min(b itself as one child - its area plus
    // its own 4-wide node, c(left, 4 - i) + c(right, i) at best -, c(left, i) + c(right, k - i)).  (Replacing the inner
    // child with the largest box, the usual rule, is this choice made greedily; the plan promises 3-5 % fewer steps.)
    struct FoldPlan {
        std::vector<uint8_t> split;   // [4 * b + k - 1]: slots of the k that go to b's left subtree, 0 = b stays one child
        std::vector<uint8_t> root;    // [b]: slots of the four that go to b's left subtree when b is a 4-wide node
        double cost = 0.0;            // sum of the areas of all children of all 4-wide nodes
    };
    static FoldPlan plan_fold(const std::vector<BuildNode> &bn) {
        FoldPlan p;
        const size_t n = bn.size();
        p.split.assign(4 * n, 0);
        p.root.assign(n, 0);
        std::vector<double> c(4 * n, 0.0);
        for (size_t b = n; b-- > 0;) {   // (children were created after their parents)
            const double area = bn[b].box.area();
            if (bn[b].left < 0) { for (int k = 0; k < 4; ++k) c[4 * b + k] = area; continue; }
            const double *cl = &c[4 * (size_t)bn[b].left], *cr = &c[4 * (size_t)bn[b].right];
            double best = cl[0] + cr[2];
            int bi = 1;
            for (int i = 2; i <= 3; ++i) if (cl[i - 1] + cr[3 - i] < best) { best = cl[i - 1] + cr[3 - i]; bi = i; }
            p.root[b] = (uint8_t)bi;
            if (b == 0) p.cost = best;
            c[4 * b] = area + best;
            for (int k = 2; k <= 4; ++k) {
                double v = c[4 * b];
                int sp = 0;
                for (int i = 1; i < k; ++i) if (cl[i - 1] + cr[k - i - 1] < v) { v = cl[i - 1] + cr[k - i - 1]; sp = i; }
                c[4 * b + k - 1] = v;
                p.split[4 * b + k - 1] = (uint8_t)sp;
            }
        }
        return p;
    }
    FoldPlan plan;

    // Children of the 4-wide node rooted at binary node b, as planned.  With level_fold both of b's inner children are
    // replaced by theirs instead (the grandchildren), so every inner child of the 4-wide node sits exactly two binary
    // levels below it.
    int fold(int32_t b, int32_t out[4]) const {
        int n = 0;
        if (level_fold) {
            const int32_t two[2] = {bn[b].left, bn[b].right};
            for (int k = 0; k < 2; ++k) {
                if (bn[two[k]].left >= 0) { out[n++] = bn[two[k]].left; out[n++] = bn[two[k]].right; }
                else out[n++] = two[k];
            }
            return n;
        }
        place(bn[b].left, plan.root[b], out, n);
        place(bn[b].right, 4 - plan.root[b], out, n);
        return n;
    }
    void place(int32_t x, int k, int32_t out[4], int &n) const {
        const int sp = bn[x].left < 0 ? 0 : plan.split[4 * (size_t)x + k - 1];
        if (sp == 0) { out[n++] = x; return; }
        place(bn[x].left, sp, out, n);
        place(bn[x].right, k - sp, out, n);
    }

    // Self check of the device image: walks the units the way the traversal addresses them and counts (a) children whose
    // box, decoded exactly as origin + q * 2^e, does not contain the padded box of every primitive below them - such a
    // primitive could be skipped by a ray that hits it -, and (b) boundable primitives that are not reachable exactly once.
    uint64_t check() const {
        uint64_t bad = 0, seen = 0;
        struct FBox { double lo[3], hi[3]; };
        auto rec = [&](auto &&self, uint32_t unit, int depth) -> FBox {
            FBox all;
            for (int d = 0; d < 3; ++d) { all.lo[d] = 1e300; all.hi[d] = -1e300; }
            if (depth > BVH_STACK || (size_t)unit >= unit_prim.size() || unit_prim[unit] != MISS_ID) { ++bad; return all; }
            uint32_t w[16];
            std::memcpy(w, &units[4 * (size_t)unit], 64);
            // children the node says it has (its "missing" bits): a tree of at least three units has none with fewer than two
            const uint32_t missing = w[3] >> 30, n_children = missing == 0u ? 4u : (missing == 2u ? 3u : 2u);
            if (missing == 1u) ++bad;
            float org[3];
            std::memcpy(org, w, 12);
            for (int j = 0; j < 4; ++j) {
                int qlo[3], qhi[3];
                bool exists = true;
                for (int d = 0; d < 3; ++d) {
                    const uint32_t pair = (w[4 + 2 * d + j / 2] >> (16 * (j % 2))) & 0xffffu;
                    qlo[d] = (int)(pair & 0xffu); qhi[d] = (int)(pair >> 8);
                    if (qlo[d] > qhi[d]) exists = false;
                }
                // the node's own count and its inside-out boxes must agree (children 0 and 1 are never marked missing: only
                // the root of a tree of one or no primitive lacks them, and such a tree is not walked)
                if (exists != ((uint32_t)j < n_children) && (unit_prim.size() >= 3 || j >= 2)) { ++bad; continue; }
                if (!exists) continue;
                const uint32_t child = w[10] + (uint32_t)j;
                FBox sub;
                if ((w[3] >> (24 + j)) & 1u) {   // (bits 24..27: leaf children)
                    if ((size_t)child >= unit_prim.size() || unit_prim[child] == MISS_ID) { ++bad; continue; }
                    const Box &b = pbox[unit_prim[child]];
                    for (int d = 0; d < 3; ++d) { sub.lo[d] = down(b.lo[d] - pad); sub.hi[d] = up(b.hi[d] + pad); }
                    ++seen;
                } else sub = self(self, child, depth + 1);
                for (int d = 0; d < 3; ++d) {
                    const double cell = std::ldexp(1.0, (int)(int8_t)((w[3] >> (8 * d)) & 0xffu));
                    if (sub.lo[d] < (double)org[d] + qlo[d] * cell || sub.hi[d] > (double)org[d] + qhi[d] * cell) { ++bad; break; }
                }
                for (int d = 0; d < 3; ++d) { all.lo[d] = std::min(all.lo[d], sub.lo[d]); all.hi[d] = std::max(all.hi[d], sub.hi[d]); }
            }
            return all;
        };
        if (!unit_prim.empty()) rec(rec, 0u, 0);
        if (seen != idx.size() || seen != n_leaf_prims) ++bad;
        return bad;
    }

    // What the surface-area heuristic expects of a binary tree once it is folded as planned: the traversal steps of a
    // random line through the root's box = 1 + the sum over all 4-wide nodes' children of area(child) / area(root).
    static double estimate_steps(const std::vector<BuildNode> &bn) {
        if (bn.empty() || bn[0].left < 0) return 2.0;
        return 1.0 + plan_fold(bn).cost / std::max(bn[0].box.area(), 1e-300);
    }

    void flatten() {
        // Unit 0 is the root.  A node's children get a run of consecutive units when the node is written; nodes are
        // written depth first, so a subtree's units stay close together and rays that are close in space share cache
        // lines on the way down.
        units.assign(4, make_float4(0.f, 0.f, 0.f, 0.f));
        unit_prim.assign(1, MISS_ID);
        n_nodes = n_leaf_prims = 0;
        depth4 = 1;
        auto grow = [&](int n) {
            const uint32_t first = (uint32_t)unit_prim.size();
            units.resize(units.size() + 4 * (size_t)n, make_float4(0.f, 0.f, 0.f, 0.f));
            unit_prim.resize(unit_prim.size() + (size_t)n, MISS_ID);
            return first;
        };
        if (bn[0].left < 0) {  // a single leaf: wrap it in a root
            const int32_t only[1] = {0};
            const uint32_t first = grow(1);
            write_node(0, only, 1, first);
            write_leaf(first, bn[0]);
            return;
        }
        struct Todo { int32_t b; uint32_t unit; int level; };   // (binary node, where its 4-wide node goes, its level in the 4-wide tree)
        std::vector<Todo> todo{{0, 0u, 1}};
        while (!todo.empty()) {
            const Todo t = todo.back();
            todo.pop_back();
            depth4 = std::max(depth4, t.level);
            int32_t ch[4];
            const int n = fold(t.b, ch);
            const uint32_t first = grow(n);
            write_node(t.unit, ch, n, first);
            for (int k = n - 1; k >= 0; --k) {
                if (bn[ch[k]].left >= 0) todo.push_back({ch[k], first + (uint32_t)k, t.level + 1});
                else write_leaf(first + (uint32_t)k, bn[ch[k]]);
            }
        }
    }
};

}  // namespace pt
