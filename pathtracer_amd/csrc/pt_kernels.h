// pt_kernels.h -- the wavefront pipeline that replaces the reference's CPU tile pool
// (reference src/window.rs:166-196 fan-out + src/tracer.rs:618-647 render + :448-584 trace_radiance).
//
// One pipeline pass traces `S` pixel-samples.  The recursion of trace_radiance becomes a forward sweep
// over path-tree levels followed by a backward sweep:
//
//   level l holds one RECORD per trace_radiance call at recursion depth max_depth - l:
//     ray    (origin, direction, sample id, path-node id, parent record + child slot)
//     node   (kind, coefficients)            written by the shade step of that level
//     val0/1 (radiance of child 0 / child 1)  written by the children (leaves: at once, inner nodes: on the
//                                             backward sweep)
//   forward : raygen -> for l = 0..D-1: [trace, shade] (or the fused trace_shade) -> terminal (level D)
//   backward: gather for l = D-1..1 (node_value -> parent's val slot) -> resolve (level 0: tone map, gamma,
//             quantise, integer accumulate)
//
// Evaluating every node from its children's final values reproduces the reference's floating-point
// association exactly (k4*(k3*(k2*(k1*L))) rather than a running throughput product), glass branching
// included, and needs no float atomics: every val slot has exactly one writer.
//
// Queues: each level is split in 64 shards (one per lane of a wave, so a wave can scan the shard counts
// with one load).  A wave appends all its children with ONE returning atomic on its own shard's counter
// (ballot + prefix popcount); 64 counters keep that atomic far from the ~90 atomics/us a single word
// sustains.  Record index = shard * shard_cap + offset.
#pragma once

#include "pt_scene.h"
#include "pt_bvh.h"

namespace pt {

constexpr int NSHARD = 64;
constexpr int CSTRIDE = 32;  // u32 words between two shard counters: one 128-byte line each, because atomics on one
                             // line serialise (~90/us) no matter which word of it they hit
constexpr int BLOCK = 256;
#ifndef PT_SHADE_WAVES
#define PT_SHADE_WAVES 6  // waves per SIMD the shade kernel is compiled for (bounds its VGPR budget).  Measured on the sample
                          // scene, fused kernel ms per 16 steps: 5 waves (89-95 VGPRs, no spill) 100.5, 6 (79, 36-44 B of scratch)
                          // 95.3, 7 (72, 72 B) 96.4, 8 (64, 124 B) 131.8
#endif

struct Level {
    // One allocation per level, carved into SoA arrays of n = NSHARD * shard_cap records: the three 16-byte arrays
    // first, then the three 4-byte ones.  Only the base is passed to the kernels (three Levels of ten pointers each
    // cost 60 scalar registers and pushed the fused kernel into spilling them); the accessors below rebuild the
    // array addresses with a few scalar instructions.
    //
    // A record's ray is dead once the record has been shaded, and its children's radiance only arrives after that
    // (from the next level's launch, the terminal kernel or the backward sweep): the two share storage.  val0 is
    // rayA's array and val1 rayB's; with the node kind inside coef's fourth word (pack_node) a record is 60 bytes.
    char *base;
    uint32_t *count;    // [NSHARD * CSTRIDE] records per shard, shard q at count[q * CSTRIDE]
    uint32_t *overflow; // this pass's flag: set when an append found a level full (the pass is then not accumulated)
    uint32_t shard_cap;
    PT_HD size_t n() const { return (size_t)shard_cap * 64u; }  // NSHARD = 64
    PT_HD float4 *rayA() const { return reinterpret_cast<float4 *>(base); }                  // (origin.xyz, dir.x)
    PT_HD float4 *rayB() const { return reinterpret_cast<float4 *>(base + 16 * n()); }       // (dir.y, dir.z, bits(rng key), bits(path node))
    PT_HD float4 *coef() const { return reinterpret_cast<float4 *>(base + 32 * n()); }       // node coefficients (xyz) and, in w, the node kind (pack_node)
    PT_HD float4 *val0() const { return rayA(); }                                            // radiance of child 0 (after shading)
    PT_HD float4 *val1() const { return rayB(); }                                            // radiance of child 1, the glass reflection (after shading)
    PT_HD uint32_t *parent() const { return reinterpret_cast<uint32_t *>(base + 48 * n()); } // (parent record index << 1) | child slot
    PT_HD uint32_t *hitprim() const { return reinterpret_cast<uint32_t *>(base + 52 * n()); } // nearest hit: primitive id | HIT_OUT_BIT if Transition::Out, or MISS
    PT_HD float *hit_t() const { return reinterpret_cast<float *>(base + 56 * n()); }        // its distance (with hitprim: two-kernel and trace-ahead pipelines)
    static constexpr size_t RECORD_BYTES = 60;
};

struct PassParams {
    uint32_t n_pixels;        // pixels this pass renders: all P of the image, or the pixels of this handle's tiles (pixel_map)
    uint32_t sample_first;    // global sample index of local sample 0 of this pass
    uint32_t sample_stride;   // sample-index shards: shard_count; tile shards: 1
    const uint32_t *pixel_map;  // tile shards: [n_pixels] the image pixel (y * width + x) of the pass's k-th pixel - the handle's 32x32
                                // tiles one after the other, row-major inside a tile, so that a wave of 64 level-0 records is two rows
                                // of one tile; null: the identity
};

// device-side counters (one allocation)
struct Counters { uint32_t overflow; uint32_t pad; unsigned long long rays, records, shadow, bvh_steps, terminal; unsigned long long stamps[8];
                  unsigned long long level_records[25]; };

// Diagnostic builds (-DPT_STAMPS) time the phases of the fused kernel with s_memtime; never part of a shipped build.
#ifdef PT_STAMPS
#define PT_STAMP(k) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); unsigned long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp_acc[k] += t_ - stamp_last; stamp_last = t_; } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#endif

typedef float f4s_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }

// Records are written once and read once per pass, a launch or more apart, and a level is far larger than the caches:
// the node stores of the shade step and every access of the backward sweep carry the non-temporal hint, so they do not
// displace what is about to be read (measured at 1080p on the sample scene: sweep kernels -6 %, whole job +2.2 %; the
// same hint on the ray loads, the child-ray stores and the terminal kernel changed nothing or cost a little).
__device__ __forceinline__ float4 ld_stream(const float4 *p) {
    f4s_t v = __builtin_nontemporal_load(reinterpret_cast<const f4s_t *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t ld_stream(const uint32_t *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld_stream(const float *p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void st_stream(float4 *p, float4 v) {
    f4s_t w = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(w, reinterpret_cast<f4s_t *>(p));
}
__device__ __forceinline__ void st_stream(uint32_t *p, uint32_t v) { __builtin_nontemporal_store(v, p); }

// ---- shard bookkeeping ------------------------------------------------------------------------------
struct TileMap { uint32_t cnt, incl, total; };  // per lane: records and inclusive tile prefix of shard `lane`

__device__ __forceinline__ TileMap tile_map(const Level &L) {
    TileMap m;
    uint32_t lane = lane_id();
    uint32_t c = L.count[lane * CSTRIDE];
    m.cnt = c < L.shard_cap ? c : L.shard_cap;
    uint32_t incl = (m.cnt + 63u) >> 6;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t v = __shfl_up(incl, o);
        if (lane >= (uint32_t)o) incl += v;
    }
    m.incl = incl;
    m.total = __shfl(incl, 63);
    return m;
}
// Record handled by this lane in wave-tile `w` (wave-uniform); false for padding lanes.
__device__ __forceinline__ bool tile_record(const TileMap &m, const Level &L, uint32_t w, uint32_t &idx) {
    uint32_t q = (uint32_t)__popcll(__ballot(m.incl <= w));  // shards entirely before tile w
    uint32_t before = __shfl(m.incl, (int)((q + 63u) & 63u));
    if (q == 0) before = 0;
    uint32_t cq = __shfl(m.cnt, (int)q);
    uint32_t off = (w - before) * 64u + lane_id();
    idx = q * L.shard_cap + off;
    return off < cq;
}

// Reserve room in `L` for this wave's children, in two steps so the returning atomic can stay in flight while the
// wave keeps computing: wave_reserve issues it (lane 0), wave_resolve consumes the result.  Every lane of the wave
// must call both.
struct Reservation { unsigned long long m0, m1; uint32_t base; };

__device__ __forceinline__ Reservation wave_reserve(const Level &L, uint32_t shard, bool has0, bool has1) {
    Reservation r;
    r.m0 = __ballot(has0);
    r.m1 = __ballot(has1);
    r.base = 0;
    uint32_t total = (uint32_t)__popcll(r.m0) + (uint32_t)__popcll(r.m1);
    if (total != 0 && lane_id() == 0) r.base = atomicAdd(&L.count[shard * CSTRIDE], total);
    return r;
}

__device__ __forceinline__ void wave_resolve(const Level &L, uint32_t shard, const Reservation &r, bool &has0, bool &has1, uint32_t &i0,
                                             uint32_t &i1, Counters * /*ctr*/) {
    uint32_t n0 = (uint32_t)__popcll(r.m0), n1 = (uint32_t)__popcll(r.m1), total = n0 + n1;
    i0 = i1 = 0;
    if (total == 0) return;
    uint32_t base = __shfl(r.base, 0);
    if (base + total > L.shard_cap) {  // never drop silently: the pass is voided and the host re-runs it with larger levels
        if (lane_id() == 0) atomicOr(L.overflow, 1u);
        has0 = has1 = false;
        return;
    }
    unsigned long long lt = (1ull << lane_id()) - 1ull;
    i0 = shard * L.shard_cap + base + (uint32_t)__popcll(r.m0 & lt);
    i1 = shard * L.shard_cap + base + n0 + (uint32_t)__popcll(r.m1 & lt);
}

// A pass in which an append found a level full is void: nothing it computes is accumulated or counted, and the host
// renders its samples again.  The wave whose reservation did not fit wrote no children, yet the shard's counter already
// includes them, so the records behind that counter are whatever the memory held before - rays and parent links that
// nobody wrote.  No kernel may consume them: every kernel of a pass returns at once when the pass's flag is already up
// (it is raised by an EARLIER launch of the same stream, or by this one, whose stragglers then merely stop early).
__device__ __forceinline__ bool pass_voided(const Level &L) { return __builtin_nontemporal_load(L.overflow) != 0u; }
// The same decision for kernels whose blocks meet at a barrier: there it has to be ONE decision per block.  The flag may be
// raised by another block of the same launch at any moment, so waves that each read it for themselves can disagree, and a
// wave that leaves before a barrier the others wait at is undefined behaviour (and the ones that stay would read shared
// memory the leavers never filled).  Thread 0 reads the flag once and publishes it through shared memory; the caller's
// own barrier (the one that publishes the rest of its shared data) makes it visible, after which every thread of the
// block takes the same branch.
#define PT_BLOCK_VOID_DECL __shared__ uint32_t s_pass_void
#define PT_BLOCK_VOID_READ(L) do { if (threadIdx.x == 0) s_pass_void = __builtin_nontemporal_load((L).overflow); } while (0)
#define PT_BLOCK_VOIDED (s_pass_void != 0u)

__device__ __forceinline__ uint32_t my_shard() { return (blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) & (NSHARD - 1); }
__device__ __forceinline__ uint32_t wave_global() { return blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6); }
__device__ __forceinline__ uint32_t waves_in_grid() { return gridDim.x * (BLOCK / 64); }

// Image pixel of the pass's k-th pixel (see PassParams::pixel_map).
__device__ __forceinline__ uint32_t pass_pixel(const PassParams &pp, uint32_t k) { return pp.pixel_map ? pp.pixel_map[k] : k; }

// ---- nearest hit (linear scan or BVH) -----------------------------------------------------------------
// The linear scan reads the same primitive in every lane: through a constant-address-space pointer the
// compiler issues s_load (scalar cache -> SGPRs) instead of 64 identical vector loads per fetch.
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef const f4_t __attribute__((address_space(4))) *const_f4_ptr;

// `transition`: the winner's Transition (0 In, 1 Out: a sphere left from inside).  BVH is a compile-time choice:
// kernels for scenes without one carry no traversal code - with it inlined the fused kernel needs 117 instead of 89-95
// registers (4 instead of 5 waves per SIMD: measured 20.4 vs 22.1 Grays/s on the sample scene).
typedef const float4 __attribute__((address_space(3))) *lds_f4_ptr;
template <bool BVH, bool DEFER = false>
__device__ __forceinline__ uint32_t nearest_hit(const DevScene &sc, V3 o, V3 d, float &t, int &transition, lds_f4_ptr lds_sph = nullptr) {
    if (BVH) return bvh_nearest(sc, o, d, t, transition);
    if (DEFER)   // (a lane's own candidate spheres come from the block's LDS copy of the scene)
        return nearest_brute<true>((const_f4_ptr)(uintptr_t)sc.sph, sc.n_sph, (const_f4_ptr)(uintptr_t)sc.pln, sc.n_pl, o, d, t, transition, lds_sph);
    return nearest_brute<false>((const_f4_ptr)(uintptr_t)sc.sph, sc.n_sph, (const_f4_ptr)(uintptr_t)sc.pln, sc.n_pl, o, d, t, transition);
}
template <bool BVH>
__device__ __forceinline__ uint32_t nearest_hit(const DevScene &sc, V3 o, V3 d, float &t) {
    int transition;
    return nearest_hit<BVH>(sc, o, d, t, transition);
}

// Can the ray reach the box [lo, hi] at all?  Conservative on purpose: approximate reciprocals, the far distance widened,
// every comparison written so that a NaN says "maybe".  The box is the BVH builder's (the region in which the
// reference's own tests can accept a hit, padded by 2^-16 of the scene's size), so "no" means that no primitive test can
// succeed (reference src/scene.rs:263-297 would return None) and the scan over the primitives can be skipped.
__device__ __forceinline__ bool may_reach_box(V3 lo, V3 hi, V3 o, V3 d) {
    const float big = 1e30f;
    float ix = __builtin_fabsf(d.x) > 1e-30f ? __builtin_amdgcn_rcpf(d.x) : __builtin_copysignf(big, d.x);
    float iy = __builtin_fabsf(d.y) > 1e-30f ? __builtin_amdgcn_rcpf(d.y) : __builtin_copysignf(big, d.y);
    float iz = __builtin_fabsf(d.z) > 1e-30f ? __builtin_amdgcn_rcpf(d.z) : __builtin_copysignf(big, d.z);
    float ax = (lo.x - o.x) * ix, bx = (hi.x - o.x) * ix;
    float ay = (lo.y - o.y) * iy, by = (hi.y - o.y) * iy;
    float az = (lo.z - o.z) * iz, bz = (hi.z - o.z) * iz;
    float tn = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));   // (fminf/fmaxf drop a NaN operand)
    float tf = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    tf = tf * 1.0001f + 1e-4f;
    return !(tn > tf) && !(tf < 0.0f);
}

// ---- raygen: reference src/tracer.rs:125-142,628-631 -------------------------------------------------------
// A ray record carries the sample's RNG key (not the sample id), so no level has to redo the pixel/sample split.
// `sid` = local sample * n_pixels + k is the level-0 record index; the counter RNG is keyed on the IMAGE pixel and the
// GLOBAL sample index, so which handle renders a pixel-sample, and in which pass, does not change it.
__device__ __forceinline__ void primary_ray(const DevScene &sc, const PassParams &pp, uint32_t sid, V3 &o, V3 &d, uint32_t &skey) {
    const uint32_t k = sid % pp.n_pixels, s_local = sid / pp.n_pixels;
    const uint32_t pixel = pass_pixel(pp, k);
    skey = rng_sample_key(sc.seed, pixel, pp.sample_first + s_local * pp.sample_stride);
    camera_ray(sc, pixel % sc.width, pixel / sc.width, skey, o, d);
}

// Only the two-kernel (trace, shade) pipeline of scenes without a BVH stores primary rays; the others generate them in place.
__global__ __launch_bounds__(BLOCK) void k_raygen(DevScene sc, Level L0, PassParams pp, uint32_t n_records) {
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_records; i += gridDim.x * BLOCK) {
        V3 o, d;
        uint32_t skey;
        primary_ray(sc, pp, i, o, d, skey);
        L0.rayA()[i] = make_float4(o.x, o.y, o.z, d.x);
        L0.rayB()[i] = make_float4(d.y, d.z, u2f(skey), u2f(1u));
    }
}

// Primitive ids stay below 2^28, so bit 30 is free to carry the Transition of the hit next to the id.
constexpr uint32_t HIT_OUT_BIT = 1u << 30;
__device__ __forceinline__ uint32_t pack_hit(uint32_t prim, int transition) {
    return prim == MISS ? MISS : (prim | (transition == 1 ? HIT_OUT_BIT : 0u));
}

// What a kernel that traverses the tree leaves in hitprim for a ray whose answer only the reference's own scan over every
// primitive can give (BVH_RESCAN in pt_bvh.h: a NaN distance was accepted, or the ray is not finite).  The count of such
// records of a level is kept in word 2 of the level's first counter line (zeroed with the counters, every pass), and a
// second kernel, launched behind the first, repeats them with the linear scan - or returns at once when the count is
// zero, which is what it is unless a scene holds NaN or infinite coordinates or a sphere of radius zero.  Keeping the
// scan out of the traversing kernels keeps its registers out of them (inlined it cost k_trace_dyn 36 spilled registers
// and the terminal kernel two of its six waves per SIMD).
constexpr uint32_t HIT_RESCAN = 0xFFFFFFFEu;

// ---- trace: reference src/scene.rs:263-297 ------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void k_trace(DevScene sc, Level L) {
    if (pass_voided(L)) return;
    TileMap m = tile_map(L);
    for (uint32_t w = wave_global(); w < m.total; w += waves_in_grid()) {
        uint32_t i;
        if (!tile_record(m, L, w, i)) continue;
        float4 a = L.rayA()[i], b = L.rayB()[i];
        float t;
        int tr;
        uint32_t prim = nearest_hit<false>(sc, v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), t, tr);  // (BVH scenes use k_trace_dyn)
        L.hitprim()[i] = pack_hit(prim, tr);
        L.hit_t()[i] = t;
    }
}

// BVH scenes: rays of one wave need very different numbers of traversal steps (a grazing ray that misses walks
// hundreds of nodes, its neighbour hits at once); with one ray per lane the wave idles at ~13 % lane utilisation
// (measured on the 1M-rectangle field).  Here waves claim CHUNKS of records from a per-level counter (one atomic per
// wave and chunk) and a lane that finishes its ray takes the next unprocessed record of its wave's chunk (rank among
// idle lanes by ballot + popcount), so lanes stay busy until the whole level runs dry.  Results go to
// hitprim[record]; which lane traced a record is irrelevant.  The constants below were measured on that scene
// (1080p, depth 4, whole job, Mrays/s) with the others at their defaults: tiles per chunk 2: 6500, 4: 6770, 8: 6600,
// 16: 6220 (the launch ends when its slowest wave does); steps between refill checks 4: 6770, 6: 6770, 8: 6430;
// waves per SIMD 5: 6140, 6: 6240, 7: 6430, 8: 5570; leaf turns every step: 6120, every 2nd: 6230, every 4th: 6220.
#ifndef PT_DYN_TILES
#define PT_DYN_TILES 4
#endif
#ifndef PT_DYN_STEPS
#define PT_DYN_STEPS 6
#endif
constexpr uint32_t DYN_TILES = PT_DYN_TILES;   // wave-tiles (of 64 records) per chunk
static_assert(DYN_TILES >= 1 && DYN_TILES <= 64 && (DYN_TILES & (DYN_TILES - 1)) == 0, "a chunk's tiles are described by the first lanes of a wave");
constexpr int DYN_STEPS = PT_DYN_STEPS;        // traversal steps between refill checks
#ifndef PT_DYN_LEAF_EVERY
#define PT_DYN_LEAF_EVERY 2
#endif
constexpr int DYN_LEAF_EVERY = PT_DYN_LEAF_EVERY;  // lanes on a leaf primitive take their step in every n-th step of the wave
static_assert((DYN_LEAF_EVERY & (DYN_LEAF_EVERY - 1)) == 0 && DYN_STEPS % DYN_LEAF_EVERY == 0, "leaf turns must recur within a round of steps");

#ifndef PT_TRACE_WAVES
#define PT_TRACE_WAVES 7
#endif
// RAYGEN: level 0, whose rays are generated here from the record index instead of being read (the shade kernel that
// follows generates them again: cheaper than a kernel that writes them and two that read them).
template <bool RAYGEN>
__global__ __launch_bounds__(BLOCK, PT_TRACE_WAVES) void k_trace_dyn(DevScene sc, Level L, PassParams pp, Counters *ctr) {
    __shared__ uint32_t s_incl[NSHARD], s_cnt[NSHARD];
    PT_BLOCK_VOID_DECL;
    PT_BLOCK_VOID_READ(L);
    TileMap m = tile_map(L);
    if (threadIdx.x < NSHARD) { s_incl[threadIdx.x] = m.incl; s_cnt[threadIdx.x] = m.cnt; }
    __syncthreads();
    if (PT_BLOCK_VOIDED) return;
    const TravStack stk = trav_stack(sc);
#ifdef PT_TRAV_STATS
    unsigned long long n_steps = 0, n_slots = 0, n_leaf = 0, ray_steps = 0, max_steps = 0;
#endif
    const unsigned long long lt = (1ull << lane_id()) - 1ull;
    const uint32_t n_chunks = (m.total + DYN_TILES - 1) / DYN_TILES;
    // Chunk counters: the level's chunks are cut into eight ranges, each with a counter of its own (a spare word of the
    // first eight counter lines, zeroed per pass), and a wave starts with the range of the XCD it runs on: the eight L2s are
    // separate, and waves that walk neighbouring rays through the tree re-read each other's nodes (1M-rectangle field:
    // trace kernel -1.2 %).  A wave whose range is used up goes on with the next one, so the ranges only say where a wave
    // starts - whatever the register reads (a partitioned device may report one XCD for all), every chunk is claimed once.
    uint32_t *work = L.count + 1;
    const uint32_t xcc = (uint32_t)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u;   // HW_REG_XCC_ID, bits 3:0
    uint32_t ranges_done = 0;
    uint32_t end = 0, next = 0, rec = 0;
    uint32_t tile_base = 0, tile_cnt = 0;   // lane t < DYN_TILES: first record and record count of tile t of the wave's chunk
    bool exhausted = false, active = false;
    Trav st;
    st.node = BVH_EMPTY;
    auto finish_record = [&]() {
        uint32_t code = pack_hit(st.best == MISS ? MISS : st.best_slot, st.best_out & 1);   // by shading row, not by id
        if (st.best_out & BVH_RESCAN) {   // (degenerate data only: k_rescan, which follows this kernel, decides)
            code = HIT_RESCAN;
            atomicAdd(L.count + 2, 1u);
        }
        L.hitprim()[rec] = code;
        L.hit_t()[rec] = st.best_t;
    };
    uint32_t my_steps = 0;  // 64-byte units (2 per inner node, 1 per leaf primitive) this lane fetched: the kernel's algorithmic traffic
    for (;;) {
        unsigned long long idle = __ballot(!active);
        if (idle != 0 && next >= end && !exhausted) {  // chunk used up: claim the next one
            uint32_t c = n_chunks;
            while (ranges_done < 8u) {   // (one returning atomic per wave and chunk, and one more per range a wave finds used up)
                const uint32_t x = (xcc + ranges_done) & 7u;
                const uint32_t lo = (uint32_t)(((uint64_t)n_chunks * x) >> 3), hi = (uint32_t)(((uint64_t)n_chunks * (x + 1u)) >> 3);
                uint32_t k = 0;
                if (lane_id() == 0) k = atomicAdd(work + x * CSTRIDE, 1u);
                k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
                if (k < hi - lo) { c = lo + k; break; }
                ++ranges_done;
            }
            if (c < n_chunks) {
                const uint32_t first_tile = c * DYN_TILES;
                const uint32_t tiles = m.total - first_tile < DYN_TILES ? m.total - first_tile : DYN_TILES;
                end = tiles * 64u;
                next = 0;
                // where each tile of the chunk lies, found once per chunk (by the first lanes) instead of once per record
                tile_cnt = 0;
                if (lane_id() < tiles) {
                    const uint32_t w = first_tile + lane_id();
                    uint32_t lo = 0, hi = NSHARD - 1;  // first shard whose inclusive tile prefix exceeds w
                    while (lo < hi) {
                        uint32_t mid = (lo + hi) >> 1;
                        if (s_incl[mid] > w) hi = mid; else lo = mid + 1;
                    }
                    const uint32_t before = lo ? s_incl[lo - 1] : 0u;
                    const uint32_t off = (w - before) * 64u;
                    tile_base = lo * L.shard_cap + off;
                    tile_cnt = s_cnt[lo] > off ? s_cnt[lo] - off : 0u;   // (64 or more: a full tile)
                }
            } else exhausted = true;
        }
        if (idle != 0 && next < end) {  // hand the next records of the chunk to the idle lanes
            const uint32_t k = next + (uint32_t)__popcll(idle & lt);
            const uint32_t base = (uint32_t)__shfl((int)tile_base, (int)((k >> 6) & (DYN_TILES - 1))), cnt = (uint32_t)__shfl((int)tile_cnt, (int)((k >> 6) & (DYN_TILES - 1)));
            if (!active && k < end && (k & 63u) < cnt) {
                rec = base + (k & 63u);
                V3 ro, rd;
                if (RAYGEN) {
                    uint32_t skey;
                    primary_ray(sc, pp, rec, ro, rd, skey);
                } else {
                    float4 a = L.rayA()[rec], b = L.rayB()[rec];
                    ro = v3(a.x, a.y, a.z); rd = v3(a.w, b.x, b.y);
                }
                bvh_begin(sc, ro, rd, st);
                active = st.node != BVH_EMPTY;
                if (!active) finish_record();   // (a ray that is not let into the tree: left to k_rescan)
#ifdef PT_TRAV_STATS
                if (ray_steps > max_steps) max_steps = ray_steps;
                ray_steps = 0;
#endif
            }
            next += (uint32_t)__popcll(idle);
        }
        if (__ballot(active) == 0) {
            if (exhausted) break;
            continue;
        }
#pragma unroll 1
        for (int it = 0; it < DYN_STEPS; ++it) {
            if (active) {
#ifdef PT_TRAV_STATS
                ++n_steps; ++ray_steps;
                if (st.node < 0) ++n_leaf;
#endif
                my_steps += bvh_step(sc, st, stk, (it & (DYN_LEAF_EVERY - 1)) == DYN_LEAF_EVERY - 1);
                if (st.node == BVH_EMPTY) {
                    finish_record();
                    active = false;
                }
            }
#ifdef PT_TRAV_STATS
            ++n_slots;
#endif
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) my_steps += __shfl_xor(my_steps, o);
    if (lane_id() == 0 && my_steps) atomicAdd(&ctr->bvh_steps, (unsigned long long)my_steps);
#ifdef PT_TRAV_STATS
    atomicAdd(&ctr->stamps[0], n_steps); atomicAdd(&ctr->stamps[1], n_slots); atomicAdd(&ctr->stamps[2], n_leaf); atomicMax(&ctr->stamps[3], max_steps > ray_steps ? max_steps : ray_steps);
#endif
}

// The records k_trace_dyn left to the linear scan (HIT_RESCAN); same template parameter.
template <bool RAYGEN>
__global__ __launch_bounds__(BLOCK) void k_rescan(DevScene sc, Level L, PassParams pp) {
    if (pass_voided(L) || L.count[2] == 0u) return;
    TileMap m = tile_map(L);
    for (uint32_t w = wave_global(); w < m.total; w += waves_in_grid()) {
        uint32_t i;
        if (!tile_record(m, L, w, i)) continue;
        if (L.hitprim()[i] != HIT_RESCAN) continue;
        V3 o, d;
        if (RAYGEN) {
            uint32_t skey;
            primary_ray(sc, pp, i, o, d, skey);
        } else {
            float4 a = L.rayA()[i], b = L.rayB()[i];
            o = v3(a.x, a.y, a.z); d = v3(a.w, b.x, b.y);
        }
        float t;
        int tr;
        const uint32_t prim = nearest_brute<false>(sc.sph, sc.n_sph, sc.pln, sc.n_pl, o, d, t, tr);
        L.hitprim()[i] = pack_hit(prim == MISS ? MISS : sc.prim_rows[prim], tr);
        L.hit_t()[i] = t;
    }
}

// The backward sweep works on PAIRS of levels.  A node of an even level PULLS the radiance of its children out of
// their records one level up (it evaluates that odd-level node itself: leaf -> its coef, inner node -> node_value of
// its coefficients and ITS children's radiance) and hands its own radiance down: level 0 to the pixel (the resolve
// step), levels 2, 4, ... into their parent's val slot.  Odd levels therefore need no launch of their own, and the
// 16-byte write plus read per record that pushing through them would cost go away.  For that
//   * a leaf of an odd level keeps its radiance in its own coef instead of pushing it down,
//   * a record of an even level remembers where its children were queued, in the two arrays that held its hit
//     (consumed by then; unused in the fused pipeline).  NO_CHILD: the child's radiance is in this record's own val
//     slot (a black terminal light sample, a leaf child of the trace-ahead pipeline, or children on the last level,
//     which k_terminal pushes).
constexpr uint32_t NO_CHILD = 0xFFFFFFFFu;
// A node's kind travels in the fourth word of its coefficients: only a diffuse node uses that word as a number (the
// cosine of reference src/tracer.rs:567), every other kind stores a tag there - the bit pattern of a SIGNALLING NaN,
// which no arithmetic instruction can return (the cosine is the result of an addition: were it a NaN it would be a quiet
// one), so a diffuse node's cosine is never mistaken for a tag.  The word is only ever moved and compared as bits.
constexpr uint32_t KIND_TAG = 0x7FA00000u;
__device__ __forceinline__ float4 pack_node(uint32_t kind, V3 coef, float coef_w) {
    const uint32_t w = kind == NK_DIFFUSE ? f2u(coef_w) : (KIND_TAG | kind);
    return make_float4(coef.x, coef.y, coef.z, u2f(w));
}
__device__ __forceinline__ uint32_t node_kind(float4 c) {
    const uint32_t w = f2u(c.w);
    return (w & 0xFFFFFFF8u) == KIND_TAG ? (w & 7u) : (uint32_t)NK_DIFFUSE;
}
// A leaf of an even level >= 2 hands its radiance to its parent at once and leaves this in the word that holds an inner
// node's first child (hitprim), which is how the sweep tells it from a node it has to evaluate.
constexpr uint32_t LEAF_DONE = 0xFFFFFFFEu;
__device__ __forceinline__ bool pulled_level(uint32_t level) { return (level & 1u) != 0u; }
__device__ __forceinline__ void remember_children(const Level &L, uint32_t i, uint32_t kind, uint32_t last_traced, bool has0, bool has1,
                                                  uint32_t i0, uint32_t i1) {
    if (kind == NK_LEAF) return;
    const bool pull = !last_traced;  // (children on the last level are terminal samples: k_terminal pushes)
    L.hitprim()[i] = (pull && has0) ? i0 : NO_CHILD;
    if (kind == NK_GLASS) L.hit_t()[i] = u2f((pull && has1) ? i1 : NO_CHILD);
}

// Radiance of child `slot` of record i of an even level: pulled from its record one level up, or read from the
// record's own val slot when the child was never queued there.
__device__ __forceinline__ V3 child_radiance(const Level &L, const Level &Lup, uint32_t i, int slot, bool pull) {
    uint32_t c = NO_CHILD;
    if (pull) c = slot ? f2u(ld_stream(&L.hit_t()[i])) : ld_stream(&L.hitprim()[i]);
    if (c == NO_CHILD) {
        float4 v = slot ? ld_stream(&L.val1()[i]) : ld_stream(&L.val0()[i]);
        return v3(v.x, v.y, v.z);
    }
    float4 k = ld_stream(&Lup.coef()[c]);
    const uint32_t kind = node_kind(k);
    if (kind == NK_LEAF) return v3(k.x, k.y, k.z);
    float4 a = ld_stream(&Lup.val0()[c]);
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    if (kind == NK_GLASS) b = ld_stream(&Lup.val1()[c]);
    return node_value(kind, v3(k.x, k.y, k.z), k.w, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z));
}

// The same, in three steps that a caller can interleave over several records: the sweep kernels are bound by the latency
// of these dependent fetches (record -> child's coefficients -> the child's own children), not by bandwidth, and a wave
// that walks a few records' chains side by side keeps that many more requests in flight.
//   child_fetch_begin: the child's coefficients (or the record's own val slot) are requested;
//   child_fetch_next:  what the coefficients ask for - the child's val slots - is requested;
//   child_fetch_value: the child's radiance.
// `on` false: nothing is requested, the value is zero.  The arithmetic is child_radiance's.
struct ChildFetch { float4 k, a, b; bool up; };
__device__ __forceinline__ void child_fetch_begin(ChildFetch &f, const Level &L, const Level &Lup, uint32_t i, uint32_t c, bool on) {
    f.k = f.a = f.b = make_float4(0.f, 0.f, 0.f, 0.f);
    f.up = on && c != NO_CHILD;
    if (on) f.k = f.up ? ld_stream(&Lup.coef()[c]) : ld_stream(&L.val0()[i]);
}
__device__ __forceinline__ void child_fetch_next(ChildFetch &f, const Level &Lup, uint32_t c) {
    if (!f.up) return;
    const uint32_t kind = node_kind(f.k);
    if (kind != NK_LEAF) f.a = ld_stream(&Lup.val0()[c]);
    if (kind == NK_GLASS) f.b = ld_stream(&Lup.val1()[c]);
}
__device__ __forceinline__ V3 child_fetch_value(const ChildFetch &f) {
    if (!f.up) return v3(f.k.x, f.k.y, f.k.z);
    const uint32_t kind = node_kind(f.k);
    if (kind == NK_LEAF) return v3(f.k.x, f.k.y, f.k.z);
    return node_value(kind, v3(f.k.x, f.k.y, f.k.z), f.k.w, v3(f.a.x, f.a.y, f.a.z), v3(f.b.x, f.b.y, f.b.z));
}

// ---- shade: reference src/tracer.rs:494-583 ----------------------------------------------------------------------
// One record of level `level`: material switch, node coefficients, children appended to Lnext (level + 1).
// At level >= 1 a leaf stores its radiance straight into its parent's val slot; at level 0 it keeps it in its own
// coef for the resolve step.  At the last traced level (children have depth 0) a child whose terminal light sample
// needs no shadow ray (reference src/tracer.rs:472-476: Plane::sample said None, or there are no lights) is worth
// zero, which is written into this node's val slot at once instead of queueing a record for k_terminal.
// Every lane of the wave must call this (the append uses wave-wide ballots); `valid` masks padding lanes.
template <bool ROWS = false>
__device__ __forceinline__ void shade_record(const DevScene &sc, const Level &L, const Level &Lprev, const Level &Lnext, uint32_t level,
                                             uint32_t last_traced, uint32_t shard, bool valid, uint32_t i, V3 o, V3 d, uint32_t skey,
                                             uint32_t node, uint32_t prim, float hit_t, int hit_transition, Counters *ctr
#ifdef PT_STAMPS
                                             , unsigned long long *stamp_acc, unsigned long long &stamp_last
#endif
                                             ) {
    // Phase 1: what the hit is and how many children it spawns -- enough to reserve queue space.  The returning
    // atomic of wave_append is issued before the expensive arithmetic of phase 2 and is only waited for when the
    // children are written, so its latency (~1-3 us under load) hides under the shading math.
    ShadeCtx c;
    c.n_children = 0;
    const uint32_t nkey = rng_node_key(skey, node);
    if (valid) {
        shade_prepare<ROWS>(sc, o, d, prim, hit_t, hit_transition, nkey, c);
    }
    bool has0 = valid && c.n_children >= 1, has1 = valid && c.n_children >= 2;
    bool folded = false;  // every child is a black terminal sample: the node's own radiance is known here and now
    if (last_traced) {
        V3 dir, rad;
        float dist;
        const bool had0 = has0, had1 = has1;
        if (has0 && !terminal_sample(sc, c.o0, rng_node_key(skey, 2u * node), dir, dist, rad)) has0 = false;
        if (has1 && !terminal_sample(sc, c.o1, rng_node_key(skey, 2u * node + 1u), dir, dist, rad)) has1 = false;
        folded = had0 && !has0 && !has1;
        if (!folded) {  // (a glass node with one child black and the other queued)
            if (had0 && !has0) L.val0()[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (had1 && !has1) L.val1()[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    PT_STAMP(2);  // hit rebuild, material fetch, arm decision, terminal culling
    Reservation rsv = wave_reserve(Lnext, shard, has0, has1);
    PT_STAMP(3);  // the append atomic (diagnostic builds wait for it here)
    // Phase 2: directions and coefficients (no loads in here: a wait for a later load would also wait for the atomic)
    NodeOut out;
    if (valid) {
        shade_finish(c, d, nkey, out);
        if (folded) {
            // node_value with the children's exact zeros, as the backward sweep would compute it - but now, so the
            // node is handed to its parent like a leaf and the sweep only reads its kind (in the closed sample box
            // more than nine of ten nodes of the last traced level end this way)
            const V3 zero = v3(0.f, 0.f, 0.f);
            out.coef = node_value(out.kind, out.coef, out.coef_w, zero, zero);
            out.kind = NK_LEAF;
        }
        if (out.kind == NK_LEAF && level != 0 && !pulled_level(level)) {
            // (the parent link is fetched only here: in closed scenes few records are leaves)
            const uint32_t par = L.parent()[i];
            float4 v = make_float4(out.coef.x, out.coef.y, out.coef.z, 0.0f);
            if (par & 1u) Lprev.val1()[par >> 1] = v; else Lprev.val0()[par >> 1] = v;
            st_stream(&L.hitprim()[i], LEAF_DONE);
        } else {  // (a leaf of level 0 or of an odd level is read where it is: by the resolve step, or by the level below)
            st_stream(&L.coef()[i], pack_node(out.kind, out.coef, out.coef_w));
        }
    }
    PT_STAMP(4);  // sampling / BRDF arithmetic + node stores
    uint32_t i0, i1;
    wave_resolve(Lnext, shard, rsv, has0, has1, i0, i1, ctr);
    PT_STAMP(5);
    if (has0) {
        Lnext.rayA()[i0] = make_float4(out.o0.x, out.o0.y, out.o0.z, out.d0.x);
        Lnext.rayB()[i0] = make_float4(out.d0.y, out.d0.z, u2f(skey), u2f(2u * node));
        Lnext.parent()[i0] = (i << 1);
    }
    if (has1) {
        Lnext.rayA()[i1] = make_float4(out.o1.x, out.o1.y, out.o1.z, out.d1.x);
        Lnext.rayB()[i1] = make_float4(out.d1.y, out.d1.z, u2f(skey), u2f(2u * node + 1u));
        Lnext.parent()[i1] = (i << 1) | 1u;
    }
    if (!pulled_level(level) && valid) remember_children(L, i, out.kind, last_traced, has0, has1, i0, i1);
}

// Small scenes keep their primitive and material tables in LDS for the shading step: there every lane fetches the
// rows of the primitive IT hit (a divergent gather: two dependent round trips to the vector cache per record when
// read from global memory; ~64-cycle LDS reads instead).  The intersection loop keeps its scalar loads: it reads
// the same row in all lanes.
constexpr int LDS_SCENE_CAP = 1024;  // float4 rows: n_sph + 4 n_pl + 4 (n_sph + n_pl) must fit (16 KiB)
__host__ __device__ inline bool lds_scene_fits(uint32_t n_sph, uint32_t n_pl) { return 5ull * n_sph + 8ull * n_pl <= LDS_SCENE_CAP; }

// Wave-granular kernel: each wave takes 64 consecutive records.  FUSED: intersection here as well; RAYGEN: level 0
// generates its primary rays in place; LDS_SCENE: see above (the host checks lds_scene_fits).  BVH without FUSED: the
// hits were left by k_trace_dyn, which names them by shading row (DevScene::shade_rows).  DEFER: the linear scan tests
// the spheres that survive its cheap rejections lane by lane (scenes with more than DEFER_SPHERES spheres).
template <bool FUSED, bool RAYGEN, bool LDS_SCENE, bool BVH, bool DEFER = false>
__global__ __launch_bounds__(BLOCK, (BVH && FUSED) ? 4 : PT_SHADE_WAVES) void k_shade(DevScene sc, Level L, Level Lprev, Level Lnext, PassParams pp, uint32_t level,
                                                 uint32_t last_traced, Counters *ctr) {
    __shared__ float4 s_scene[LDS_SCENE ? LDS_SCENE_CAP : 1];
    PT_BLOCK_VOID_DECL;
    DevScene lsc = sc;  // what the shading step reads
    if (LDS_SCENE) {
        PT_BLOCK_VOID_READ(L);
        const uint32_t n_s = sc.n_sph, n_p = 4u * sc.n_pl, n_m = 4u * (sc.n_sph + sc.n_pl);
        for (uint32_t j = threadIdx.x; j < n_s; j += BLOCK) s_scene[j] = sc.sph[j];
        for (uint32_t j = threadIdx.x; j < n_p; j += BLOCK) s_scene[n_s + j] = sc.pln[j];
        for (uint32_t j = threadIdx.x; j < n_m; j += BLOCK) s_scene[n_s + n_p + j] = sc.mat[j];
        __syncthreads();
        if (PT_BLOCK_VOIDED) return;   // (one decision per block, after its only barrier)
        lsc.sph = s_scene;
        lsc.pln = s_scene + n_s;
        lsc.mat = s_scene + n_s + n_p;
    } else if (pass_voided(L)) return;   // (no barrier in this variant: a wave may decide for itself)
    TileMap m = tile_map(L);
    const uint32_t shard = my_shard();
#ifdef PT_STAMPS
    unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long stamp_last = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    // The rays of a wave's NEXT tile are requested before the current tile is traced and shaded, so their latency
    // (an HBM round trip: the records were written by the previous launch) is covered by a whole tile of arithmetic.
    uint32_t w = wave_global();
    uint32_t i = 0;
    bool valid = false;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    if (w < m.total) {
        valid = tile_record(m, L, w, i);
        if (!RAYGEN && valid) { ra = L.rayA()[i]; rb = L.rayB()[i]; }
    }
    while (w < m.total) {
        const uint32_t w_next = w + waves_in_grid();
        uint32_t i_next = 0;
        bool valid_next = false;
        float4 ra_next = make_float4(0.f, 0.f, 0.f, 0.f), rb_next = ra_next;
        if (w_next < m.total) {
            valid_next = tile_record(m, L, w_next, i_next);
#ifndef PT_NO_PREFETCH
            if (!RAYGEN && valid_next) { ra_next = L.rayA()[i_next]; rb_next = L.rayB()[i_next]; }
#endif
        }
        V3 o = v3(0.f, 0.f, 0.f), d = o;
        uint32_t skey = 0, node = 0, prim = MISS;
        float hit_t = 0.0f;
        int hit_tr = -1;
        if (valid) {
            if (RAYGEN) {
                primary_ray(sc, pp, i, o, d, skey);
                node = 1u;
            } else {
                o = v3(ra.x, ra.y, ra.z); d = v3(ra.w, rb.x, rb.y);
                skey = f2u(rb.z);
                node = f2u(rb.w);
            }
        }
        PT_STAMP(0);  // tile bookkeeping + ray fetch
        if (valid) {
            // primary rays of a linear-scan scene: a wave none of whose rays can reach the scene's box skips the scan
            const bool reach = !(RAYGEN && !BVH) || !sc.has_bounds || may_reach_box(sc.bounds_lo, sc.bounds_hi, o, d);
            if (FUSED) {
                if (!(RAYGEN && !BVH) || __ballot(reach) != 0ull) prim = nearest_hit<BVH, DEFER>(sc, o, d, hit_t, hit_tr, (lds_f4_ptr)s_scene);
            } else {  // the trace kernel left the winner, its Transition and its distance
                uint32_t word = L.hitprim()[i];
                hit_t = L.hit_t()[i];
                prim = word == MISS ? MISS : (word & ~HIT_OUT_BIT);
                hit_tr = (word != MISS && (word & HIT_OUT_BIT)) ? 1 : 0;
            }
        }
        PT_STAMP(1);  // intersection
#ifdef PT_STAMPS
        shade_record<(BVH && !FUSED)>(lsc, L, Lprev, Lnext, level, last_traced, shard, valid, i, o, d, skey, node, prim, hit_t, hit_tr, ctr, stamp_acc, stamp_last);
#else
        shade_record<(BVH && !FUSED)>(lsc, L, Lprev, Lnext, level, last_traced, shard, valid, i, o, d, skey, node, prim, hit_t, hit_tr, ctr);
#endif
        PT_STAMP(6);  // child stores
        w = w_next; i = i_next; valid = valid_next; ra = ra_next; rb = rb_next;
#ifdef PT_NO_PREFETCH
        if (!RAYGEN && valid) { ra = L.rayA()[i]; rb = L.rayB()[i]; }
#endif
    }
#ifdef PT_STAMPS
    if (lane_id() == 0) for (int k = 0; k < 8; ++k) if (stamp_acc[k]) atomicAdd(&ctr->stamps[k], stamp_acc[k]);
#endif
}

// ---- trace-ahead pipeline (small scenes) ---------------------------------------------------------------------------
// The same recursion with the nearest-hit query moved from the head of a level to the tail of its parent: a
// kernel shades its records (whose hit is already known), generates the child rays AND traces them at once.  A
// child that ends on the sky or on an emitter is a leaf: its radiance goes straight into this node's own val slot
// and it never becomes a record.  Measured on the sample scene about 40 % of the rays below the primary ones are
// such leaves; they used to cost a 36-byte record write, a 32-byte read, a scattered 16-byte write into the
// parent and, worst, an idle lane through the whole shading step of their wave.  Records of levels >= 1 carry
// (primitive | transition bit, distance) next to the ray, so every lane of a deeper-level wave has real work.
// The linear scan alone (the host selects this pipeline only for scenes without a BVH).
__device__ __forceinline__ uint32_t scan_nearest(const DevScene &sc, V3 o, V3 d, float &t, int &transition) {
    return nearest_brute((const_f4_ptr)(uintptr_t)sc.sph, sc.n_sph, (const_f4_ptr)(uintptr_t)sc.pln, sc.n_pl, o, d, t, transition);
}

__device__ __forceinline__ bool child_is_leaf(const DevScene &sc, const DevScene &lsc, uint32_t prim, V3 dir, V3 &value) {
    if (prim == MISS) { value = sky_radiance(sc, dir); return true; }                    // reference src/tracer.rs:571-583
    float4 m0 = lsc.mat[4 * prim];
    uint32_t mat = f2u(m0.x);
    if (mat == MAT_NONE) { value = v3(1.0f, 1.0f, 1.0f); return true; }                  // :502
    if (mat == MAT_EMISSIVE) { value = v3(m0.y, m0.z, m0.w); return true; }              // :503-505
    return false;
}

#ifndef PT_AHEAD_WAVES
#define PT_AHEAD_WAVES 3
#endif
template <bool RAYGEN, bool LDS_SCENE>
__global__ __launch_bounds__(BLOCK, PT_AHEAD_WAVES) void k_ahead(DevScene sc, Level L, Level Lnext, PassParams pp, uint32_t level,
                                                               uint32_t last_traced, uint32_t *traced_count, Counters *ctr) {
    __shared__ float4 s_scene[LDS_SCENE ? LDS_SCENE_CAP : 1];
    PT_BLOCK_VOID_DECL;
    DevScene lsc = sc;  // what the shading step reads
    if (LDS_SCENE) {
        PT_BLOCK_VOID_READ(L);
        const uint32_t n_s = sc.n_sph, n_p = 4u * sc.n_pl, n_m = 4u * (sc.n_sph + sc.n_pl);
        for (uint32_t j = threadIdx.x; j < n_s; j += BLOCK) s_scene[j] = sc.sph[j];
        for (uint32_t j = threadIdx.x; j < n_p; j += BLOCK) s_scene[n_s + j] = sc.pln[j];
        for (uint32_t j = threadIdx.x; j < n_m; j += BLOCK) s_scene[n_s + n_p + j] = sc.mat[j];
        __syncthreads();
        if (PT_BLOCK_VOIDED) return;   // (one decision per block, after its only barrier)
        lsc.sph = s_scene;
        lsc.pln = s_scene + n_s;
        lsc.mat = s_scene + n_s + n_p;
    } else if (pass_voided(L)) return;   // (no barrier in this variant: a wave may decide for itself)
    TileMap m = tile_map(L);
    const uint32_t shard = my_shard();
    uint32_t n_traced = 0;  // nearest-hit queries of this wave (wave-uniform)
    // the next tile's records are requested before the current tile is processed
    uint32_t w = wave_global();
    uint32_t i = 0;
    bool valid = false;
    float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
    uint32_t rp = MISS;
    float rt = 0.0f;
    if (w < m.total) {
        valid = tile_record(m, L, w, i);
        if (!RAYGEN && valid) { ra = L.rayA()[i]; rb = L.rayB()[i]; rp = L.hitprim()[i]; rt = L.hit_t()[i]; }
    }
    while (w < m.total) {
        const uint32_t w_next = w + waves_in_grid();
        uint32_t i_next = 0;
        bool valid_next = false;
        float4 ra_next = make_float4(0.f, 0.f, 0.f, 0.f), rb_next = ra_next;
        uint32_t rp_next = MISS;
        float rt_next = 0.0f;
        if (w_next < m.total) {
            valid_next = tile_record(m, L, w_next, i_next);
            if (!RAYGEN && valid_next) { ra_next = L.rayA()[i_next]; rb_next = L.rayB()[i_next]; rp_next = L.hitprim()[i_next]; rt_next = L.hit_t()[i_next]; }
        }
        V3 o = v3(0.f, 0.f, 0.f), d = o;
        uint32_t skey = 0, node = 0, prim = MISS;
        float hit_t = 0.0f;
        int hit_tr = 0;
        if (valid) {
            if (RAYGEN) {
                primary_ray(sc, pp, i, o, d, skey);
                node = 1u;
                prim = scan_nearest(sc, o, d, hit_t, hit_tr);
            } else {
                o = v3(ra.x, ra.y, ra.z); d = v3(ra.w, rb.x, rb.y);
                skey = f2u(rb.z);
                node = f2u(rb.w);
                prim = rp & ~HIT_OUT_BIT;  // never MISS: leaves are not queued
                hit_tr = (rp & HIT_OUT_BIT) ? 1 : 0;
                hit_t = rt;
            }
        }
        if (RAYGEN) n_traced += (uint32_t)__popcll(__ballot(valid));
        // ---- this node: reference src/tracer.rs:494-569
        const uint32_t nkey = rng_node_key(skey, node);
        ShadeCtx c;
        c.n_children = 0;
        NodeOut out;
        if (valid) {
            shade_prepare(lsc, o, d, prim, hit_t, hit_tr, nkey, c);
            shade_finish(c, d, nkey, out);
            // (only a primary ray can be a leaf here; the resolve step reads its radiance from coef)
            L.coef()[i] = pack_node(out.kind, out.coef, out.coef_w);
        }
        bool has0 = valid && c.n_children >= 1, has1 = valid && c.n_children >= 2;
        uint32_t hp0 = MISS, hp1 = MISS;
        float t0 = 0.0f, t1 = 0.0f;
        if (last_traced) {
            // children have depth 0: terminal light samples (:451-492); the ones that need no shadow ray are black
            V3 dir, rad;
            float dist;
            if (has0 && !terminal_sample(sc, c.o0, rng_node_key(skey, 2u * node), dir, dist, rad)) {
                L.val0()[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                has0 = false;
            }
            if (has1 && !terminal_sample(sc, c.o1, rng_node_key(skey, 2u * node + 1u), dir, dist, rad)) {
                L.val1()[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                has1 = false;
            }
        } else {
            // ---- the children's nearest hits, now (:448-449 of the callee); one copy of the scan serves both slots
            n_traced += (uint32_t)__popcll(__ballot(has0)) + (uint32_t)__popcll(__ballot(has1));
#pragma unroll 1
            for (int slot = 0; slot < 2; ++slot) {
                const bool has = slot == 0 ? has0 : has1;
                if (__ballot(has) == 0) continue;  // (slot 1 exists only under glass)
                if (has) {
                    V3 co = slot == 0 ? out.o0 : out.o1, cd = slot == 0 ? out.d0 : out.d1;
                    float ct;
                    int tr;
                    uint32_t p = scan_nearest(sc, co, cd, ct, tr);
                    V3 value;
                    const bool leaf = child_is_leaf(sc, lsc, p, cd, value);
                    if (leaf) {
                        float4 v = make_float4(value.x, value.y, value.z, 0.0f);
                        if (slot == 0) L.val0()[i] = v; else L.val1()[i] = v;
                    }
                    if (slot == 0) { has0 = !leaf; hp0 = pack_hit(p, tr); t0 = ct; }
                    else { has1 = !leaf; hp1 = pack_hit(p, tr); t1 = ct; }
                }
            }
        }
        Reservation rsv = wave_reserve(Lnext, shard, has0, has1);
        uint32_t i0, i1;
        wave_resolve(Lnext, shard, rsv, has0, has1, i0, i1, ctr);
        if (has0) {
            Lnext.rayA()[i0] = make_float4(out.o0.x, out.o0.y, out.o0.z, out.d0.x);
            Lnext.rayB()[i0] = make_float4(out.d0.y, out.d0.z, u2f(skey), u2f(2u * node));
            Lnext.parent()[i0] = (i << 1);
            if (!last_traced) { Lnext.hitprim()[i0] = hp0; Lnext.hit_t()[i0] = t0; }
        }
        if (has1) {
            Lnext.rayA()[i1] = make_float4(out.o1.x, out.o1.y, out.o1.z, out.d1.x);
            Lnext.rayB()[i1] = make_float4(out.d1.y, out.d1.z, u2f(skey), u2f(2u * node + 1u));
            Lnext.parent()[i1] = (i << 1) | 1u;
            if (!last_traced) { Lnext.hitprim()[i1] = hp1; Lnext.hit_t()[i1] = t1; }
        }
        if (!pulled_level(level) && valid) remember_children(L, i, out.kind, last_traced, has0, has1, i0, i1);
        w = w_next; i = i_next; valid = valid_next; ra = ra_next; rb = rb_next; rp = rp_next; rt = rt_next;
    }
    if (n_traced != 0 && lane_id() == 0) atomicAdd(&traced_count[shard * CSTRIDE], n_traced);
}

// Block-granular kernel: the 256 records a block takes per step are intersected first, then REGROUPED through LDS
// by the match arm they will take (shade_class), so that three of the block's four waves typically run one arm of
// the material switch instead of all of them (the arms differ by hundreds of instructions: GGX + Cook-Torrance vs
// cosine sampling vs refraction).  A stable counting sort over 6 classes: per-wave ballots -> 4x6 counts in LDS ->
// every lane derives its slot.  Records carry their own index, so which lane shades a record is irrelevant.
constexpr int NCLASS = 6;  // diffuse, specular, glass, mirror, leaf, padding
__device__ __forceinline__ uint32_t class_slot(uint32_t nk) {
    return nk == NK_DIFFUSE ? 0u : nk == NK_SPECULAR ? 1u : nk == NK_GLASS ? 2u : nk == NK_MIRROR ? 3u : 4u;
}

template <bool RAYGEN, bool BVH>
__global__ __launch_bounds__(BLOCK, PT_SHADE_WAVES) void k_shade_sorted(DevScene sc, Level L, Level Lprev, Level Lnext, PassParams pp,
                                                                      uint32_t level, uint32_t last_traced, Counters *ctr) {
    __shared__ float4 s_a[BLOCK], s_b[BLOCK];
    __shared__ uint2 s_c[BLOCK];
    __shared__ uint32_t s_cnt[BLOCK / 64][8];
    PT_BLOCK_VOID_DECL;
    PT_BLOCK_VOID_READ(L);
    __syncthreads();
    if (PT_BLOCK_VOIDED) return;   // (one decision per block: its steps below meet at barriers)
    TileMap m = tile_map(L);
    const uint32_t shard = my_shard();
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t n_steps = (m.total + (BLOCK / 64) - 1) / (BLOCK / 64);
    for (uint32_t step = blockIdx.x; step < n_steps; step += gridDim.x) {
        const uint32_t w = step * (BLOCK / 64) + wave;
        uint32_t i = 0;
        bool valid = false;
        if (w < m.total) valid = tile_record(m, L, w, i);
        V3 o = v3(0.f, 0.f, 0.f), d = o;
        uint32_t skey = 0, node = 0, prim = MISS, cls = NCLASS - 1;
        if (valid) {
            if (RAYGEN) {
                primary_ray(sc, pp, i, o, d, skey);
                node = 1u;
            } else {
                float4 a = L.rayA()[i], b = L.rayB()[i];
                o = v3(a.x, a.y, a.z); d = v3(a.w, b.x, b.y);
                skey = f2u(b.z);
                node = f2u(b.w);
            }
            float t;
            prim = nearest_hit<BVH>(sc, o, d, t);
            cls = class_slot(shade_class(sc, prim, rng_node_key(skey, node)));
        }
        // ---- counting sort of the block's records by class
        unsigned long long mk[NCLASS];
        uint32_t my_count = 0;
        unsigned long long my_mask = 0;
#pragma unroll
        for (int c = 0; c < NCLASS; ++c) {
            mk[c] = __ballot(cls == (uint32_t)c);
            if (lane == (uint32_t)c) my_count = (uint32_t)__popcll(mk[c]);
            if (cls == (uint32_t)c) my_mask = mk[c];
        }
        if (lane < NCLASS) s_cnt[wave][lane] = my_count;
        __syncthreads();
        uint32_t slot = (uint32_t)__popcll(my_mask & ((1ull << lane) - 1ull));
#pragma unroll
        for (int c = 0; c < NCLASS; ++c) {
            uint32_t before = 0, total = 0;
#pragma unroll
            for (int ww = 0; ww < BLOCK / 64; ++ww) {
                uint32_t n = s_cnt[ww][c];
                total += n;
                if ((uint32_t)ww < wave) before += n;
            }
            if ((uint32_t)c < cls) slot += total;
            if ((uint32_t)c == cls) slot += before;
        }
        s_a[slot] = make_float4(o.x, o.y, o.z, d.x);
        s_b[slot] = make_float4(d.y, d.z, u2f(skey), u2f(node));
        s_c[slot] = make_uint2(prim, valid ? i : 0xFFFFFFFFu);
        __syncthreads();
        float4 a = s_a[threadIdx.x], b = s_b[threadIdx.x];
        uint2 c2 = s_c[threadIdx.x];
        __syncthreads();  // everyone has its record before the next step overwrites the staging arrays
        bool v2 = c2.y != 0xFFFFFFFFu;
#ifdef PT_STAMPS
        unsigned long long stamp_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_last = 0;
        shade_record(sc, L, Lprev, Lnext, level, last_traced, shard, v2, c2.y, v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), f2u(b.z), f2u(b.w),
                     c2.x, 0.0f, -1, ctr, stamp_acc, stamp_last);
#else
        shade_record(sc, L, Lprev, Lnext, level, last_traced, shard, v2, c2.y, v3(a.x, a.y, a.z), v3(a.w, b.x, b.y), f2u(b.z), f2u(b.w),
                     c2.x, 0.0f, -1, ctr);
#endif
    }
}

// ---- terminal level (depth == 0): reference src/tracer.rs:451-492 ------------------------------------------------
// Holds only the light samples that need a shadow ray (k_shade drops the others).  `shadow_count` is a row of
// per-shard counters like Level::count (zeroed every pass, summed by k_account).
// RESCAN: the second kernel of a BVH scene (see HIT_RESCAN) - only the records the first one marked, by linear scan.
// (Six waves per SIMD for the traversing kernel: two registers over that budget would cost it the sixth.)
template <bool BVH, bool RESCAN = false>
__global__ __launch_bounds__(BLOCK, BVH ? 6 : 1) void k_terminal(DevScene sc, Level L, Level Lprev, uint32_t *shadow_count) {
    if (pass_voided(L)) return;
    if (RESCAN && L.count[2] == 0u) return;
    TileMap m = tile_map(L);
    uint32_t n_shadow = 0;
    for (uint32_t w = wave_global(); w < m.total; w += waves_in_grid()) {
        uint32_t i;
        bool valid = tile_record(m, L, w, i);
        bool shadow = false;
        if (RESCAN && valid) {   // (the level's hitprim array has no other use: terminal records are not traced)
            valid = L.hitprim()[i] == HIT_RESCAN;
            if (valid) L.hitprim()[i] = 0u;
        }
        if (valid) {
            float4 a = L.rayA()[i], b = L.rayB()[i];
            V3 o = v3(a.x, a.y, a.z);
            uint32_t nkey = rng_node_key(f2u(b.z), f2u(b.w));
            V3 value = v3(0.0f, 0.0f, 0.0f), dir, radiance;
            float dist;
            shadow = terminal_sample(sc, o, nkey, dir, dist, radiance);
            if (shadow) {
                if (BVH) {
                    const int lit = bvh_light_visible(sc, o, dir, dist);   // stops at the first occluder
                    if (lit == 1) value = radiance;
                    if (lit == LIGHT_RESCAN) {   // (degenerate data only: the RESCAN launch that follows decides)
                        L.hitprim()[i] = HIT_RESCAN;
                        atomicAdd(L.count + 2, 1u);
                    }
                } else {
                    float t;
                    uint32_t prim = nearest_hit<false>(sc, o, dir, t);
                    if (prim != MISS && t > dist - 0.000001f) value = radiance;
                }
            }
            uint32_t par = L.parent()[i];
            float4 v = make_float4(value.x, value.y, value.z, 0.0f);
            if (par & 1u) Lprev.val1()[par >> 1] = v; else Lprev.val0()[par >> 1] = v;
        }
        n_shadow += (uint32_t)__popcll(__ballot(shadow));
    }
    if (!RESCAN && n_shadow != 0 && lane_id() == 0) atomicAdd(&shadow_count[my_shard() * CSTRIDE], n_shadow);
}

// ---- backward sweep: the inner nodes of an even level l >= 2 hand their radiance to their parents -----------------
// `pull`: level l + 1 is a traced level, so the children's radiance is taken from their records there (see
// remember_children); otherwise (l is the last traced level) it is in the val slots, put there by k_terminal.
// Wave-tiles whose records a wave walks side by side.  Measured at 1080p (k_gather alone, ms per 16 steps, one pass in
// flight / whole job with the default two, Mrays/s): sample scene 1 tile 199 / 31 570, 2: 199 / 31 530, 4 (56 registers):
// 206 / 31 790, 6: 240 / 30 400, 8 (174 registers): 290 / 29 660; closed room 1: 415 / 24 450, 4: 418 / 24 060.  Its
// scattered 16-byte reads and writes keep HBM at 4 TB/s of actual traffic whatever a wave has in flight.
#ifndef PT_GATHER_TILES
#define PT_GATHER_TILES 1
#endif
constexpr int GATHER_TILES = PT_GATHER_TILES;
__global__ __launch_bounds__(BLOCK) void k_gather(Level L, Level Lup, Level Lprev, uint32_t pull) {
    if (pass_voided(L)) return;
    TileMap m = tile_map(L);
    const uint32_t stride = waves_in_grid();
    for (uint32_t w = wave_global(); w < m.total; w += GATHER_TILES * stride) {
        uint32_t i[GATHER_TILES], hp[GATHER_TILES], par[GATHER_TILES];
        float4 c[GATHER_TILES];
        bool on[GATHER_TILES];
        ChildFetch f[GATHER_TILES];
        // The record's own words (a leaf - LEAF_DONE - handed its radiance down when it was shaded).  On the last traced
        // level most records are leaves (nodes whose terminal samples were all black): there the one word that says so is
        // read first; elsewhere leaves are rare and everything is requested at once.
#pragma unroll
        for (int u = 0; u < GATHER_TILES; ++u) {
            const uint32_t wu = w + (uint32_t)u * stride;
            on[u] = wu < m.total && tile_record(m, L, wu, i[u]);
            hp[u] = LEAF_DONE; par[u] = 0u;
            c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (on[u]) hp[u] = ld_stream(&L.hitprim()[i[u]]);
        }
        if (pull != 0u) {
#pragma unroll
            for (int u = 0; u < GATHER_TILES; ++u)
                if (on[u]) { c[u] = ld_stream(&L.coef()[i[u]]); par[u] = ld_stream(&L.parent()[i[u]]); }
        } else {
#pragma unroll
            for (int u = 0; u < GATHER_TILES; ++u)
                if (on[u] && hp[u] != LEAF_DONE) { c[u] = ld_stream(&L.coef()[i[u]]); par[u] = ld_stream(&L.parent()[i[u]]); }
        }
#pragma unroll
        for (int u = 0; u < GATHER_TILES; ++u) {
            on[u] = on[u] && hp[u] != LEAF_DONE;
            child_fetch_begin(f[u], L, Lup, i[u], pull != 0u ? hp[u] : NO_CHILD, on[u]);
        }
#pragma unroll
        for (int u = 0; u < GATHER_TILES; ++u) child_fetch_next(f[u], Lup, hp[u]);
#pragma unroll
        for (int u = 0; u < GATHER_TILES; ++u) {
            if (!on[u]) continue;
            const uint32_t kind = node_kind(c[u]);
            V3 a = child_fetch_value(f[u]);
            V3 b = v3(0.f, 0.f, 0.f);
            if (kind == NK_GLASS) b = child_radiance(L, Lup, i[u], 1, pull != 0u);
            V3 value = node_value(kind, v3(c[u].x, c[u].y, c[u].z), c[u].w, a, b);
            float4 v = make_float4(value.x, value.y, value.z, 0.0f);
            if (par[u] & 1u) st_stream(&Lprev.val1()[par[u] >> 1], v); else st_stream(&Lprev.val0()[par[u] >> 1], v);
        }
    }
}

// ---- resolve: reference src/tracer.rs:634-643 (tone map, gamma, Pixel32::from_unit, add_pixel32) --------------------------
// One thread owns one pixel and folds all `spp` samples of the pass (level-0 records s * n_pixels + k) into
// its u32 sums: exact integer adds, one plain read-modify-write per pixel and pass, no atomics.
// `pixels8` (may be null): the display bytes of the pixels this pass touched, from the sums it has just written and
// `num_samples`, the samples per pixel they hold - Backbuffer::assign_pixel8_unsafe (src/tracer.rs:271-282), which the
// reference also runs per pixel right after add_pixel32 (:643-644).  The last pass of a frame does this, so presenting
// costs neither a launch nor a second read of the sums.
// `flag_snapshot` (with pixels8): block 0 first copies the handle's ring of overflow flags there.  Every pass launched
// before this one has finished by now and this pass's own flag can no longer change (appends happen before the resolve
// step), so the copy is final for all of them - and it travels to the host in the same transfer as the display bytes.
constexpr uint32_t PASS_FLAG_RING = 1024;
// Samples of a pixel whose fetch chains a wave walks side by side (ChildFetch).  Measured at 1080p on the sample scene
// (k_resolve alone, ms per 16 steps of 22 spp): a plain loop over the samples 175; 2 samples (78 registers with the
// display bytes' code) 160; 3: 170; 4 (113 registers, half the waves) 188.
#ifndef PT_RESOLVE_SAMPLES
#define PT_RESOLVE_SAMPLES 2
#endif
constexpr int RESOLVE_SAMPLES = PT_RESOLVE_SAMPLES;
__global__ __launch_bounds__(BLOCK) void k_resolve(DevScene sc, Level L0, Level L1, PassParams pp, uint32_t spp, uint32_t pull,
                                                  uint32_t *pixels32, uint8_t *pixels8, uint32_t num_samples, const uint32_t *flag_ring,
                                                  uint32_t *flag_snapshot) {
    if (flag_snapshot != nullptr && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < PASS_FLAG_RING; i += BLOCK) flag_snapshot[i] = flag_ring[i];
    // A level overflowed somewhere in this pass: it adds nothing (the host repeats it).  (No launch of the pass is still
    // running when this one starts and it appends nothing itself, so every thread reads the same value.)
    if (*L0.overflow != 0u) return;
    // the quantiser's threshold table is searched 24 times per sample with per-lane indices: keep it in LDS
    __shared__ float s_gamma[256];
    if (sc.gamma_table != nullptr) {
        s_gamma[threadIdx.x] = sc.gamma_table[threadIdx.x];
        __syncthreads();
        sc.gamma_table = s_gamma;
    }
    const float n = (float)num_samples;
    // (whole waves walk the pixels, so that the lanes of a wave can hand each other their display bytes below)
    for (uint32_t first = blockIdx.x * BLOCK + (threadIdx.x & ~63u); first < pp.n_pixels; first += gridDim.x * BLOCK) {
        const uint32_t k = first + lane_id();
        const bool valid = k < pp.n_pixels;
        uint32_t acc0 = 0, acc1 = 0, acc2 = 0;
        // RESOLVE_SAMPLES samples' fetch chains side by side (see ChildFetch)
        for (uint32_t sb = 0; sb < spp; sb += RESOLVE_SAMPLES) {
            uint32_t i[RESOLVE_SAMPLES], ch[RESOLVE_SAMPLES];
            float4 c[RESOLVE_SAMPLES];
            bool on[RESOLVE_SAMPLES];
            ChildFetch f[RESOLVE_SAMPLES];
#pragma unroll
            for (int u = 0; u < RESOLVE_SAMPLES; ++u) {
                on[u] = valid && sb + (uint32_t)u < spp;
                i[u] = (sb + (uint32_t)u) * pp.n_pixels + k;
                c[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                ch[u] = NO_CHILD;
                if (on[u]) {
                    c[u] = ld_stream(&L0.coef()[i[u]]);
                    if (pull != 0u) ch[u] = ld_stream(&L0.hitprim()[i[u]]);   // (a leaf's is not used: four bytes against a round trip)
                }
            }
#pragma unroll
            for (int u = 0; u < RESOLVE_SAMPLES; ++u) child_fetch_begin(f[u], L0, L1, i[u], ch[u], on[u] && node_kind(c[u]) != NK_LEAF);
#pragma unroll
            for (int u = 0; u < RESOLVE_SAMPLES; ++u) child_fetch_next(f[u], L1, ch[u]);
#pragma unroll
            for (int u = 0; u < RESOLVE_SAMPLES; ++u) {
                if (!on[u]) continue;
                const uint32_t kind = node_kind(c[u]);
                V3 value = v3(c[u].x, c[u].y, c[u].z);
                if (kind != NK_LEAF) {
                    V3 a = child_fetch_value(f[u]);
                    V3 b = v3(0.f, 0.f, 0.f);
                    if (kind == NK_GLASS) b = child_radiance(L0, L1, i[u], 1, pull != 0u);
                    value = node_value(kind, value, c[u].w, a, b);
                }
                uint32_t q[3];
                tonemap_gamma_quantise(sc, value, q);
                acc0 += q[0]; acc1 += q[1]; acc2 += q[2];
            }
        }
        const uint32_t pixel = valid ? pass_pixel(pp, k) : 0u;
        uint32_t s0 = 0, s1 = 0, s2 = 0;
        if (valid) {
            s0 = pixels32[3 * pixel + 0] + acc0; s1 = pixels32[3 * pixel + 1] + acc1; s2 = pixels32[3 * pixel + 2] + acc2;
            pixels32[3 * pixel + 0] = s0;
            pixels32[3 * pixel + 1] = s1;
            pixels32[3 * pixel + 2] = s2;
        }
        if (pixels8 != nullptr) {
            const uint32_t b0 = f32_as_u8(min_rs(255.0f, sc.iso_factor * (float)s0 / n)), b1 = f32_as_u8(min_rs(255.0f, sc.iso_factor * (float)s1 / n)),
                           b2 = f32_as_u8(min_rs(255.0f, sc.iso_factor * (float)s2 / n));
            // Three bytes per pixel, 96 per half wave: written as 24 whole words by the first 24 lanes of each half when the
            // half's 32 pixels lie side by side in the image and start on a word (a row of a tile, or 32 pixels of a row of
            // the image: always, except at a clipped tile's edge and at the end of the image) - byte stores otherwise.
            const uint32_t half = lane_id() & 32u, j = lane_id() & 31u;
            const uint32_t pixel_h = (uint32_t)__shfl((int)pixel, (int)half);
            const bool side_by_side = valid && pixel == pixel_h + j && (pixel_h & 3u) == 0u;
            const unsigned long long ok = __ballot(side_by_side);
            const bool words = ((ok >> half) & 0xffffffffull) == 0xffffffffull;
            const uint32_t mine = b0 | (b1 << 8) | (b2 << 16);
            // word j of the half holds bytes 4j .. 4j + 3 of its 96: the tail of pixel 4j / 3 and the head of the next
            const uint32_t p0 = (4u * j) / 3u, r = (4u * j) % 3u;
            const uint32_t w0 = (uint32_t)__shfl((int)mine, (int)(half + (p0 & 31u))), w1 = (uint32_t)__shfl((int)mine, (int)(half + ((p0 + 1u) & 31u)));
            if (words) {
                if (j < 24u) {
                    const unsigned long long both = ((unsigned long long)w0 >> (8u * r)) | ((unsigned long long)w1 << (8u * (3u - r)));
                    reinterpret_cast<uint32_t *>(pixels8 + 3 * (size_t)pixel_h)[j] = (uint32_t)both;
                }
            } else if (valid) {
                pixels8[3 * pixel + 0] = (uint8_t)b0;
                pixels8[3 * pixel + 1] = (uint8_t)b1;
                pixels8[3 * pixel + 2] = (uint8_t)b2;
            }
        }
    }
}

// ---- bookkeeping -----------------------------------------------------------------------------------------------
// Adds this pass's record/ray counts to the running totals: rays = records of the traced levels 0..D-1;
// row n_levels of `counts` holds the shadow rays counted by k_terminal.
// Row n_levels + 1 holds the nearest-hit queries counted by the trace-ahead kernels (`ahead`: there a ray is no
// longer the same thing as a record); `terminal` = records of the last level (terminal light samples queued).
__global__ void k_account(const uint32_t *counts, const uint32_t *shard_caps, int n_levels, int ahead, const uint32_t *overflow, Counters *ctr) {
    if (*overflow != 0u) return;  // a voided pass is not counted either
    // one wave: lane q sums shard q over the levels, then a shuffle reduction
    const int q = threadIdx.x & 63;
    unsigned long long rec = 0, rays = 0, terminal = 0;
    for (int l = 0; l < n_levels; ++l) {
        uint32_t c = counts[(l * NSHARD + q) * CSTRIDE];
        uint32_t cap = shard_caps[l];
        unsigned long long n = c < cap ? c : cap;
        rec += n;
        if (l < n_levels - 1) rays += n; else terminal = n;
        unsigned long long level_total = n;   // records of this level over all shards
        for (int o = 32; o > 0; o >>= 1) level_total += __shfl_down(level_total, o);
        if (q == 0 && l < 25) ctr->level_records[l] += level_total;
    }
    if (ahead) rays = counts[((n_levels + 1) * NSHARD + q) * CSTRIDE];
    unsigned long long shadow = counts[(n_levels * NSHARD + q) * CSTRIDE];
    for (int o = 32; o > 0; o >>= 1) {
        rec += __shfl_down(rec, o);
        rays += __shfl_down(rays, o);
        shadow += __shfl_down(shadow, o);
        terminal += __shfl_down(terminal, o);
    }
    if (q == 0) {
        ctr->records += rec;
        ctr->rays += rays;
        ctr->shadow += shadow;
        ctr->terminal += terminal;
    }
}

// Backbuffer.pixels8: reference src/tracer.rs:271-282
__global__ __launch_bounds__(BLOCK) void k_pixels8(const uint32_t *pixels32, uint8_t *pixels8, uint32_t n_words, float iso_factor,
                                                  uint32_t num_samples) {
    float n = (float)num_samples;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < n_words; i += gridDim.x * BLOCK) {
        float v = num_samples ? min_rs(255.0f, iso_factor * (float)pixels32[i] / n) : 0.0f;
        pixels8[i] = (uint8_t)f32_as_u8(v);
    }
}

// Parity hook: primary-hit primitive ids for one sample index.
template <bool BVH>
__global__ __launch_bounds__(BLOCK) void k_primary_ids(DevScene sc, uint32_t sample_index, uint32_t *ids) {
    uint32_t n = sc.width * sc.height;
    for (uint32_t pixel = blockIdx.x * BLOCK + threadIdx.x; pixel < n; pixel += gridDim.x * BLOCK) {
        uint32_t skey = rng_sample_key(sc.seed, pixel, sample_index);
        V3 o, d;
        camera_ray(sc, pixel % sc.width, pixel / sc.width, skey, o, d);
        float t;
        ids[pixel] = nearest_hit<BVH>(sc, o, d, t);
    }
}

// Self-test: table lookup vs direct evaluation of the gamma quantiser for every float bit pattern in [first, last].
__global__ void k_check_gamma(const float *table, uint32_t first, uint32_t last, unsigned long long *mismatches) {
    unsigned long long bad = 0;
    for (uint64_t b = (uint64_t)first + blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x) {
        float x = u2f((uint32_t)b);
        if (gamma_quantise(table, x) != gamma_quantise_direct(x)) ++bad;
    }
    if (bad) atomicAdd(mismatches, bad);
}

// Test hook: the deterministic math functions as the kernels evaluate them.
__global__ void k_eval_math(int fn, const float *a, const float *b, float *out, uint64_t n) {
    for (uint64_t i = blockIdx.x * (uint64_t)blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        float x = a[i], y = b ? b[i] : 0.0f, r;
        switch (fn) {
        case 0: r = det::sinf_(x); break;
        case 1: r = det::cosf_(x); break;
        case 2: r = det::powf_(x, y); break;
        case 3: r = det::expf_(x); break;
        case 4: r = det::acosf_(x); break;
        case 6: r = sqrt_ieee(x); break;     // the two basic operations the compiler expands into instruction sequences
        case 7: r = x / y; break;
        case 8: r = (float)f2u(x) / 4294967296.0f; break;   // random32 of the argument's bit pattern (common.rs:38-41)
        // normalize as the kernels evaluate it (three divisions that may share one refined reciprocal): a component each
        case 9: r = normalize(v3(x, y, x * 0.37f + y)).x; break;
        case 10: r = normalize(v3(x, y, x * 0.37f + y)).y; break;
        case 11: r = normalize(v3(x, y, x * 0.37f + y)).z; break;
        default: r = det::atan2f_(x, y); break;
        }
        out[i] = r;
    }
}

}  // namespace pt
