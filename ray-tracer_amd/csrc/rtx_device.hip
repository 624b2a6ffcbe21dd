// rtx_device.hip — the hot path `Camera.render -> World.colorAt` as a wavefront pipeline of
// hand-written HIP kernels for gfx950 (MI355X), behind the C ABI of include/rtx.h.
//
// Pipeline per batch of pixels (all launches asynchronous on one stream, no host syncs):
//
//   k_trace_shade  level 0..depth   one launch per ray-tree level.  Level 0 generates the
//                                    camera rays (rayForPixel); deeper levels read the
//                                    compacted ray queue the previous level wrote.  Per ray:
//                                    closest hit over the flattened shape list (group gates
//                                    included), prepareComputations, shadow any-hit, Phong,
//                                    Schlick, and the <=2 child rays pushed to the next
//                                    level's queue with wave-aggregated atomics.
//   k_combine      level depth-1..0  bottom-up: interior ray-tree nodes evaluate
//                                    surface + reflected*R + refracted*(1-R) in the
//                                    reference's order and hand the result to their parent
//                                    slot (or the canvas at the root).
//
// Every pixel's colour is produced by exactly the reference's f32 operations in the
// reference's order (no float atomics, no reassociation), so the canvas is bit-reproducible
// and independent of batch size, launch geometry and GPU count.
//
// Compile with -ffp-contract=off (explicit fmaf only where the reference uses @mulAdd), no
// fast-math, correctly rounded f32 divide/sqrt (hipcc default).
//
// Reference citations are file:line into /root/reference.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_scene.h"
#include "../../include/rtx.h"
#include "../host/rt_libm.hpp"
#include "rtx_bvh.hpp"
#include "rtx_lbvh.hpp"

namespace {

// ------------------------------------------------------------------------- errors

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess) return fail(RTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// ------------------------------------------------------------------- device records

constexpr float kEpsilon = 0.00001f;  // src/tuple.zig:10
constexpr uint32_t kNone = 0xFFFFFFFFu;
// A link names whoever waits for a colour: a canvas pixel (rays of level 0, nodes of level 0) or a
// ray-tree node (index) plus which of its two child slots the colour is for.
constexpr uint32_t kLinkIndexMask = 0x0FFFFFFFu;
constexpr uint32_t kLinkOwnSingleBit = 1u << 28;  // in a node's link: the node itself has one child (k_combine skips it)
constexpr uint32_t kLinkSingleBit = 1u << 29;  // the parent node named by this link has no other child (see deliver)
constexpr uint32_t kLinkSlotBit = 1u << 30;    // 0: reflected slot, 1: refracted slot
constexpr uint32_t kLinkSchlickBit = 1u << 31; // in a node's link: the node combines with Schlick reflectance
constexpr int kMaxDepth = 5;                   // src/world.zig:20
constexpr int kLevels = kMaxDepth + 1;

struct V4 {
    float x, y, z, w;
};

// The traversal code of the triangle BVH (and the tuple math under it) also compiles for the
// host: rtx_bvh_selftest runs it on the CPU against brute force, so the culling logic is
// covered by the CPU test suite.  Kernels and everything that touches LDS stay device-only.
#define RTX_HD __host__ __device__ __forceinline__
RTX_HD uint32_t bits_of(float f) { return __builtin_bit_cast(uint32_t, f); }

// Scene records are read-only for the lifetime of a launch.  Reading the wave-uniform ones
// (the shape / node being tested by the whole wave) through the constant address space makes
// hipcc emit scalar loads (s_load) into SGPRs instead of 64 identical vector loads.
#define RT_CONST __attribute__((address_space(4)))
template <class T>
__device__ __forceinline__ const RT_CONST T* as_const(const T* p) {
    return (const RT_CONST T*)(uintptr_t)p;
}

// The ray queues and node arrays are split into kShards independent shards so that slot
// allocation (one returning atomic per wave and iteration) never funnels the whole chip
// through one word: wave w works for shard w % kShards at every level, reads that shard's
// rays and appends its children to the same shard.  A single counter word saturates near
// 88 returning atomics per microsecond on MI355X (MI355X_MICROARCH.md, row "dequeue"), which
// made the unsharded version 93 % SQ_WAIT_ANY (profiles/r01_*).
#ifndef RTX_SHARDS
#define RTX_SHARDS 64
#endif
constexpr uint32_t kShards = RTX_SHARDS;  // a multiple of 64, at most 256 (level 0's first block clears one line per thread)
static_assert(kShards % 64u == 0u && kShards <= 256u, "see k_trace_shade and k_accumulate");

struct ShardCounters {  // one 128-byte line per shard
    // per level d: low 32 bits = reflected rays queued for level d + 1, high 32 bits = interior nodes at d
    unsigned long long alloc[kLevels];
    unsigned long long hit_kinds;  // low 32: hits on patterned materials, high 32: on transparent ones
    uint32_t hits[kLevels];        // hits at level d == isShadowed calls
    uint32_t refr[kLevels];        // per level d: refracted rays queued for level d + 1 (their own region)
    uint32_t pad[32 - 2 * kLevels - 2 - kLevels - kLevels];
};
static_assert(sizeof(ShardCounters) == 128, "one cache line per shard");

// Chunk claims live on a line of their own: every wave of a shard hits them once per iteration, and on
// the allocation line they would queue behind (and ahead of) the slot atomics.
struct ClaimLine {
    uint32_t claim[kLevels];  // per level d: 64-ray chunks of this shard's level-d work handed out so far
    uint32_t pad[32 - kLevels];
};
static_assert(sizeof(ClaimLine) == 128, "one cache line per shard");

// Event counts of a statistics frame (rtx_stats: what the roofline's algorithmic flops are priced from);
// written only by the STATS kernels.
struct StatLine {
    unsigned long long bvh_box;   // box tests of the triangle-BVH walks (closest hit, predecessor, shadow)
    unsigned long long bvh_tri;   // triangle tests inside BVH leaves
    unsigned long long gate;      // group-gate box tests (bnd.intersect, src/group.zig:58)
    unsigned long long shape[6];  // linear shape tests by RTX_* type, lanes that actually test
    unsigned long long pad[16 - 9];
};
static_assert(sizeof(StatLine) == 128, "one cache line per shard");

struct Counters {
    ShardCounters shard[kShards];
    ClaimLine claims[kShards];
    StatLine stats[kShards];
};

// What shading a hit needs before (and for planes and mesh triangles: instead of) the 160-byte shape
// record: 16 bytes per shape, so that the table of a mesh of ~2000 triangles sits in LDS and a hit on
// a larger one costs one cache line instead of three.
struct HitRec {
    float n[3];     // finished world-space normal of a plane / triangle (kHitWorldNormal), else unused
    uint32_t bits;  // material [0, 20) | RTX_* type [20, 23) | kHit* flags
};
constexpr uint32_t kHitMaterialMask = (1u << 20) - 1u;
constexpr uint32_t kHitTypeShift = 20;
constexpr uint32_t kHitIdentity = 1u << 23;     // the shape's inverse is exactly the identity
constexpr uint32_t kHitDiagonal = 1u << 24;     // ... is diag(a, b, c) plus a translation
constexpr uint32_t kHitWorldNormal = 1u << 25;  // `n` is the hit's normal whatever the point
static_assert(sizeof(HitRec) == 16, "one 16-byte load per hit");

struct Totals {
    unsigned long long level_rays[kLevels];
    unsigned long long level_hits[kLevels];
    unsigned long long nodes;
    unsigned long long hits_patterned;
    unsigned long long hits_transparent;
    unsigned long long bvh_box, bvh_tri, gate, shape[6];
};

// Event tally of one wave of a STATS kernel; the other kernels carry the empty form and pay nothing.
// bvh_* are per lane (the BVH walk is per lane); gate / shape are wave totals kept by every lane alike.
template <bool ON>
struct Tally {
    uint32_t bvh_box = 0, bvh_tri = 0, gate = 0, shape[6] = {0, 0, 0, 0, 0, 0};
    RTX_HD void box() { ++bvh_box; }
    RTX_HD void tri() { ++bvh_tri; }
    __device__ __forceinline__ void gates(bool lane_tests) { gate += __popcll(__ballot(lane_tests)); }
    __device__ __forceinline__ void shapes(uint32_t type, bool lane_tests) { shape[type] += __popcll(__ballot(lane_tests)); }
};
template <>
struct Tally<false> {
    RTX_HD void box() {}
    RTX_HD void tri() {}
    __device__ __forceinline__ void gates(bool) {}
    __device__ __forceinline__ void shapes(uint32_t, bool) {}
};

struct Params {
    const rtx_shape* shapes;
    const HitRec* hitrecs;  // one per shape
    const rtx_node* nodes;
    const rtx_material* materials;
    const rtx_pattern* patterns;
    const rtx_bvh::Node* bvh_nodes;  // triangle BVHs of all gate nodes, threaded pre-order
    const rtx_bvh::Tri* bvh_tris;
    uint32_t n_nodes;
    uint32_t n_shapes;
    uint32_t n_materials;
    uint32_t n_patterns;
    uint32_t lds_bytes;  // > 0: scene tables are staged in this much dynamic LDS (which ones: k_trace_shade's TAB)
    uint32_t tab_mode;   // TAB of the kernels this frame runs
    uint32_t track_n1;  // scene has a transparent material: track the n1 predecessor
    uint32_t canon;     // this frame runs the canonical-w kernels (ray records are 32 bytes)
    uint32_t has_bvh;   // some gate node of `nodes` carries a triangle BVH
    uint32_t finite_schlick;  // every refractive index is finite and > 0 (schlick() is finite)
    uint32_t book;            // rtx_opts.containers == RTX_CONTAINERS_BOOK (and the scene tracks n1)
    uint32_t depth;
    float light_position[4];
    float light_intensity[3];
    float over_eps;  // tup.epsilon * 100.0 as an f32 product (src/world.zig:55)

    // camera (src/camera.zig:17-23 with the inverse hoisted, :77-78)
    float cam_inv[16];
    float half_width, half_height, pixel_size;
    uint32_t width;

    // this batch: local rows [row0, row0 + n_rows) of the rows this call renders
    uint32_t row0, n_rows, tiles_x;
    uint32_t strip_rows, strip_count, strip_index, compact_rows;

    // ray queues (ping-pong by level parity) and ray-tree nodes
    float4* q_org[2];
    float4* q_dir[2];
    uint2* q_link[2];
    float4* node_a;  // surface.rgb, reflectance
    float4* node_b;  // reflected slot: rgb (w unused), written whole by the reflected child
    float4* node_c;  // refracted slot: rgb (w unused), written whole by the refracted child
    uint2* node_d;   // link bits towards the parent, weight towards the parent
    uint32_t node_off[kLevels];
    uint32_t shard_cap0;   // level-0 items per shard; a shard holds shard_cap0 << (d * level_shift) rays / nodes at level d
    uint32_t level_shift;  // 1: some material spawns two children per hit (levels may double); 0: at most one

    Counters* counters;
    Counters* counters_next;  // the set the NEXT batch will use: level 0 of this batch clears it
    float* canvas;        // root results land here: 3 floats per root index
    uint32_t root_count;  // level-0 work items when rays come from the queue (rtx_color_at)
};

// ------------------------------------------------------------------------ f32 math

RTX_HD V4 v4(float x, float y, float z, float w) { return V4{x, y, z, w}; }
RTX_HD V4 add(V4 a, V4 b) { return v4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RTX_HD V4 sub(V4 a, V4 b) { return v4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
RTX_HD V4 muls(V4 a, float s) { return v4(a.x * s, a.y * s, a.z * s, a.w * s); }
RTX_HD V4 neg(V4 a) { return v4(-a.x, -a.y, -a.z, -a.w); }

// The template flag C ("canonical") marks kernels for scenes in which every point carries
// w == 1 and every vector w == 0 exactly: all shape inverses, the camera inverse and the light
// end in (0, 0, 0, 1) and no cube is translated (the reference normalises a cube normal with
// its w lane, src/cube.zig:28-44, so a translated cube leaks a non-zero w into its rays).
// There the w lanes of the reference's four-lane arithmetic only ever add an exact zero
// (u.w * v.w with one factor 0) or multiply by an exact one, so they are left out: the results
// are the same f32 values (a zero may differ in sign, which no consumer here can observe).
// The host picks C per frame (scene_is_canonical, camera_is_canonical); C == false keeps every lane.

// tup.dot — four lanes, left to right (src/tuple.zig:38-40)
template <bool C>
RTX_HD float dot(V4 u, V4 v) {
    const float s = u.x * v.x + u.y * v.y + u.z * v.z;
    return C ? s : s + u.w * v.w;
}
template <bool C>
RTX_HD float magnitude(V4 v) { return sqrtf(dot<C>(v, v)); }  // tuple.zig:50-52
template <bool C>
RTX_HD V4 normalize(V4 v) {                                   // tuple.zig:54-56
    const float m = magnitude<C>(v);
    return v4(v.x / m, v.y / m, v.z / m, C ? 0.0f : v.w / m);
}
template <bool C>
RTX_HD V4 reflect(V4 incident, V4 normal) {                   // tuple.zig:58-77
    return sub(incident, muls(muls(normal, dot<C>(incident, normal)), 2.0f));
}

// mat.mulTuple (src/matrix.zig:62-64) with a row-major 16-float matrix.
RTX_HD V4 mul_mat(const float* m, V4 b) {
    return v4(m[0] * b.x + m[1] * b.y + m[2] * b.z + m[3] * b.w, m[4] * b.x + m[5] * b.y + m[6] * b.z + m[7] * b.w,
              m[8] * b.x + m[9] * b.y + m[10] * b.z + m[11] * b.w,
              m[12] * b.x + m[13] * b.y + m[14] * b.z + m[15] * b.w);
}
// The same for a point (w == 1 when C) and for a vector (w == 0 when C).
template <bool C>
RTX_HD V4 mul_point(const float* m, V4 b) {
    if (!C) return mul_mat(m, b);
    return v4(m[0] * b.x + m[1] * b.y + m[2] * b.z + m[3], m[4] * b.x + m[5] * b.y + m[6] * b.z + m[7],
              m[8] * b.x + m[9] * b.y + m[10] * b.z + m[11], 1.0f);
}
template <bool C>
RTX_HD V4 mul_dir(const float* m, V4 b) {
    if (!C) return mul_mat(m, b);
    return v4(m[0] * b.x + m[1] * b.y + m[2] * b.z, m[4] * b.x + m[5] * b.y + m[6] * b.z,
              m[8] * b.x + m[9] * b.y + m[10] * b.z, 0.0f);
}
// mat.mul(mat.transpose(m), b) (src/matrix.zig:74-87 then :62-64); b is a normal (w == 0 when C).
template <bool C>
RTX_HD V4 mul_mat_t(const float* m, V4 b) {
    if (C) return v4(m[0] * b.x + m[4] * b.y + m[8] * b.z, m[1] * b.x + m[5] * b.y + m[9] * b.z,
                     m[2] * b.x + m[6] * b.y + m[10] * b.z, 0.0f);
    return v4(m[0] * b.x + m[4] * b.y + m[8] * b.z + m[12] * b.w, m[1] * b.x + m[5] * b.y + m[9] * b.z + m[13] * b.w,
              m[2] * b.x + m[6] * b.y + m[10] * b.z + m[14] * b.w,
              m[3] * b.x + m[7] * b.y + m[11] * b.z + m[15] * b.w);
}

// The same three products for an inverse whose upper-left 3x3 block is diagonal (scalings and
// translations only — most analytic shapes): the dropped terms are exact zeros (0 * finite), and
// adding an exact zero changes nothing but, possibly, the sign of a zero result, which no consumer
// can observe (every division on the path is guarded against a zero divisor, comparisons and
// floor/sqrt treat both zeros alike).  Canonical-w kernels only.
RTX_HD V4 mul_point_diag(const float* m, V4 b) { return v4(m[0] * b.x + m[3], m[5] * b.y + m[7], m[10] * b.z + m[11], 1.0f); }
RTX_HD V4 mul_dir_diag(const float* m, V4 b) { return v4(m[0] * b.x, m[5] * b.y, m[10] * b.z, 0.0f); }

// std.math.approxEqAbs(f32, x, 0, eps)
__device__ __forceinline__ bool approx_zero(float x) {
    if (x == 0.0f) return true;
    if (x != x) return false;
    return fabsf(x) <= kEpsilon;
}

// ------------------------------------------------------------------- intersections

// Closest hit and, when the scene has a transparent material, the n1 predecessor.
//
// The reference appends every root to one list, stable-sorts it by t and takes the first
// t >= 0 (src/world.zig:38-47, src/intersection.zig:14-28): that is the minimum t >= 0
// with ties going to the earlier insertion — a strict `<` while visiting shapes in
// insertion order.  Its n1/n2 container walk compares whole intersections including t
// (std.meta.eql, src/intersection.zig:104,128), so an exit never removes its entry: n1 is
// the refractive index of the entry just before the hit in the sorted list, i.e. the
// largest t < 0 with ties going to the LATER insertion (`>=`), or 1.0 if none; n2 is the
// hit shape's own index.  Exactly equal (t, shape) entries toggle each other out of the
// walk, hence the same-shape duplicate cancelling below and RTX_SHAPE_N1_INVISIBLE.
struct Closest {
    float t;
    uint32_t idx;
    float pred_t;
    uint32_t pred_idx;
};

// BOOK (rtx_opts.containers = RTX_CONTAINERS_BOOK): the container walk compares shapes only, as the
// book the reference follows does and as the reference's shipped refractive images were rendered
// (SURVEY Q7).  Every entry before the hit in the sorted list has t < 0 (the hit is the first t >= 0),
// and an entry toggles its shape in and out of the container list, so at the hit the list holds the
// shapes with an ODD number of negative roots, ordered by their largest negative root (ties: later
// insertion last).  n1 is the index of the last of them (else 1); the hit then toggles its own shape:
// if it was absent it is appended (n2 = its own index), if it was present it is removed (n2 = the new
// last: the runner-up if the hit shape was last, n1 otherwise).  So BOOK tracks the best TWO odd-parity
// shapes (pred_*, pred2_*) and whether the hit shape itself is one (hit_contained).
template <bool TRACK, bool BOOK = false>
struct ClosestVisitor {
    Closest c;
    float pred2_t;
    uint32_t pred2_idx;
    bool hit_contained;
    RTX_HD void init() {
        c.t = INFINITY;  // "none yet" carries the range end the BVH walk culls against (t_hi / t_lo)
        c.idx = kNone;
        c.pred_t = -INFINITY;
        c.pred_idx = kNone;
        pred2_t = -INFINITY;
        pred2_idx = kNone;
        hit_contained = false;
    }
    // Ties go to the lowest insertion index for the hit and to the highest for the predecessor.
    // The world/group walk offers shapes in insertion order (ORDERED): a strict compare is
    // enough there.  The BVH offers triangles in tree order, so it compares the index too.
    // Written as selects, not branches: inside the divergent shape tests hipcc turned the `if`
    // form into exec-mask regions whose phi copies kept three register copies of (t, idx) alive
    // and cost ~25 v_mov per sphere test (a VALU instruction is 4 SIMD cycles whatever it does).
    template <bool ORDERED>
    RTX_HD void hit(float t, uint32_t s) {
        const bool better = (t >= 0.0f) & ((c.idx == kNone) | (t < c.t) | (!ORDERED & (t == c.t) & (s < c.idx)));
        c.t = better ? t : c.t;
        c.idx = better ? s : c.idx;
        if (BOOK) hit_contained = better ? false : hit_contained;  // book_shape() of the same shape follows and decides
    }
    template <bool ORDERED>
    RTX_HD void pred(float t, uint32_t s) {
        if (!TRACK) return;
        const bool later = ORDERED ? (t >= c.pred_t) : ((t > c.pred_t) | ((t == c.pred_t) & (s > c.pred_idx)));
        const bool better = (t < 0.0f) & ((c.pred_idx == kNone) | later);
        c.pred_t = better ? t : c.pred_t;
        c.pred_idx = better ? s : c.pred_idx;
    }
    // BOOK: shape s has been offered whole; `odd`: an odd number of its roots is negative, `tm` the
    // largest of them.  Shapes arrive in insertion order, so a tie in tm goes to the newcomer.
    RTX_HD void book_shape(uint32_t s, bool odd, float tm) {
        const bool first = odd & ((c.pred_idx == kNone) | (tm >= c.pred_t));
        const bool second = odd & !first & ((pred2_idx == kNone) | (tm >= pred2_t));
        pred2_t = first ? c.pred_t : (second ? tm : pred2_t);
        pred2_idx = first ? c.pred_idx : (second ? s : pred2_idx);
        c.pred_t = first ? tm : c.pred_t;
        c.pred_idx = first ? s : c.pred_idx;
        hit_contained = (c.idx == s) ? odd : hit_contained;
    }
    static constexpr bool kTrack = TRACK;
    static constexpr bool kBook = BOOK;
    RTX_HD void visited_node() {}     // statistics hooks of the BVH walk (rtx_bvh_selftest counts with them)
    RTX_HD void tested_triangle() {}
    // The BVH walk of the closest-hit pass looks ahead of the origin only: a predecessor among
    // the mesh triangles is needed by few rays (those that end on a transparent material) and
    // is collected for exactly those by a second walk (PredVisitor, traverse_pred).
    static constexpr bool kBvhPred = false;
    RTX_HD bool done() const { return false; }
    // Range of t a BVH subtree must overlap to matter: [0, current best].
    RTX_HD float t_lo() const { return 0.0f; }
    RTX_HD float t_hi() const { return c.t; }
};

// The n1 predecessor among BVH triangles: the largest t < 0, ties to the highest insertion
// index, merged into what the insertion-ordered walk found among the other shapes.
// BOOK: a triangle has one root, so a negative root is an odd count: the walk keeps the best two
// (see ClosestVisitor) and culls against the runner-up.
template <bool BOOK = false>
struct PredVisitorT {
    float pred_t;
    uint32_t pred_idx;
    float pred2_t = -INFINITY;
    uint32_t pred2_idx = kNone;
    template <bool ORDERED>
    RTX_HD void hit(float, uint32_t) {}
    template <bool ORDERED>
    RTX_HD void pred(float t, uint32_t s) {
        const bool better = (t < 0.0f) & ((pred_idx == kNone) | (t > pred_t) | ((t == pred_t) & (s > pred_idx)));
        if (BOOK) {
            const bool second = (t < 0.0f) & !better & ((pred2_idx == kNone) | (t > pred2_t) | ((t == pred2_t) & (s > pred2_idx)));
            pred2_t = better ? pred_t : (second ? t : pred2_t);
            pred2_idx = better ? pred_idx : (second ? s : pred2_idx);
        }
        pred_t = better ? t : pred_t;
        pred_idx = better ? s : pred_idx;
    }
    static constexpr bool kBvhPred = true;
    RTX_HD void visited_node() {}
    RTX_HD void tested_triangle() {}
    RTX_HD bool done() const { return false; }
    RTX_HD float t_lo() const { return BOOK ? pred2_t : pred_t; }  // -inf while there is none (ClosestVisitor::init)
    RTX_HD float t_hi() const { return 0.0f; }
};
using PredVisitor = PredVisitorT<false>;

// isShadowed (src/world.zig:97-120): the nearest t >= 0 is < distance  <=>  some root lies
// in [0, distance).
struct ShadowVisitor {
    float distance;
    bool found;
    template <bool ORDERED>
    __device__ __forceinline__ void hit(float t, uint32_t) { found = found | ((t >= 0.0f) & (t < distance)); }
    template <bool ORDERED>
    __device__ __forceinline__ void pred(float, uint32_t) {}
    static constexpr bool kTrack = false;
    static constexpr bool kBook = false;
    static constexpr bool kBvhPred = false;
    __device__ __forceinline__ void visited_node() {}
    __device__ __forceinline__ void tested_triangle() {}
    __device__ __forceinline__ bool done() const { return found; }
    __device__ __forceinline__ float t_lo() const { return 0.0f; }
    __device__ __forceinline__ float t_hi() const { return distance; }
};

struct TMinMax {
    float tmin, tmax;
};

// checkAxis (src/cube.zig:52-67 with min/max = -1/1, src/bounds.zig:160-175)
__device__ __forceinline__ TMinMax check_axis(float mn, float mx, float origin, float direction) {
    const float tmin_numerator = mn - origin;
    const float tmax_numerator = mx - origin;
    float tmin, tmax;
    if (fabsf(direction) >= kEpsilon) {
        tmin = tmin_numerator / direction;
        tmax = tmax_numerator / direction;
    } else {
        tmin = tmin_numerator * INFINITY;
        tmax = tmax_numerator * INFINITY;
    }
    if (tmin > tmax) return TMinMax{tmax, tmin};
    return TMinMax{tmin, tmax};
}

// bnd.intersect (src/bounds.zig:149-158): infinite-line slab test, no t range.
__device__ __forceinline__ bool bounds_hit(const RT_CONST rtx_node& n, V4 o, V4 d) {
    const TMinMax x = check_axis(n.bounds_min[0], n.bounds_max[0], o.x, d.x);
    const TMinMax y = check_axis(n.bounds_min[1], n.bounds_max[1], o.y, d.y);
    const TMinMax z = check_axis(n.bounds_min[2], n.bounds_max[2], o.z, d.z);
    const float tmin = fmaxf(x.tmin, fmaxf(y.tmin, z.tmin));
    const float tmax = fminf(x.tmax, fminf(y.tmax, z.tmax));
    return tmin <= tmax;
}

// Offer the roots of one shape (insertion-ordered walk).  For the predecessor, roots of the same
// shape with exactly equal t cancel pairwise (they are equal Intersections in the container walk).
template <class V>
__device__ __forceinline__ void offer1(V& v, uint32_t s, bool n1_visible, float t) {
    v.template hit<true>(t, s);
    if constexpr (V::kBook) {
        v.book_shape(s, n1_visible & (t < 0.0f), t);
        return;
    }
    if (V::kTrack && n1_visible) v.template pred<true>(t, s);
}
template <class V>
__device__ __forceinline__ void offer2(V& v, uint32_t s, bool n1_visible, float t0, float t1) {
    v.template hit<true>(t0, s);
    v.template hit<true>(t1, s);
    if constexpr (V::kBook) {  // exactly one negative root: that one is the largest
        const bool n0 = t0 < 0.0f, n1 = t1 < 0.0f;
        v.book_shape(s, n1_visible & (n0 != n1), n0 ? t0 : t1);
        return;
    }
    if (V::kTrack && n1_visible) {  // wave-uniform flag; equal roots cancel each other: offer NaN (never a predecessor) then
        const bool distinct = t0 != t1;
        v.template pred<true>(distinct ? t0 : NAN, s);
        v.template pred<true>(distinct ? t1 : NAN, s);
    }
}
// Up to four roots (cylinder / cone: two walls, two caps); a slot without a root holds NaN, which
// is never a hit, never a predecessor and never equal to anything.
template <class V>
__device__ __forceinline__ void offer_roots(V& v, uint32_t s, bool n1_visible, const float* t) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v.template hit<true>(t[i], s);
    if constexpr (V::kBook) {
        uint32_t negatives = 0;
        float tm = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool neg_root = t[i] < 0.0f;  // false for an empty slot (NaN)
            negatives += neg_root ? 1u : 0u;
            tm = (neg_root & (t[i] > tm)) ? t[i] : tm;
        }
        v.book_shape(s, n1_visible & ((negatives & 1u) != 0u), tm);
        return;
    }
    if (V::kTrack && n1_visible) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int same = 0;  // roots of this shape with exactly this t (itself included): pairs cancel
#pragma unroll
            for (int j = 0; j < 4; ++j) same += (t[j] == t[i]) ? 1 : 0;
            v.template pred<true>((same & 1) ? t[i] : NAN, s);
        }
    }
}

constexpr uint32_t kShapeIdentity = 1u << 8;  // library-internal: the inverse is exactly the identity
constexpr uint32_t kShapeDiagonal = 1u << 9;  // library-internal: the inverse is diag(a, b, c) plus a translation, last row (0, 0, 0, 1)
constexpr uint32_t kShapeWorldNormal = 1u << 10;  // library-internal: `normal` holds the finished world-space normal (planes, triangles)

// Triangle.intersect (src/triangle.zig:32-50, Moller-Trumbore) on the object-space ray.
// cross() has w = 0, so the w lanes of e0 / e1 / p0_to_origin only ever add an exact zero to
// the dots; they are left out.
RTX_HD bool triangle_root(V4 ro, V4 rd, const float* p0, const float* e0, const float* e1, float& t) {
    const float cx = rd.y * e1[2] - rd.z * e1[1], cy = rd.z * e1[0] - rd.x * e1[2], cz = rd.x * e1[1] - rd.y * e1[0];
    const float det = e0[0] * cx + e0[1] * cy + e0[2] * cz;
    if (fabsf(det) < kEpsilon) return false;
    const float f = 1.0f / det;
    const float px = ro.x - p0[0], py = ro.y - p0[1], pz = ro.z - p0[2];
    const float u = f * (px * cx + py * cy + pz * cz);
    if (u < 0.0f || u > 1.0f) return false;
    const float ox = py * e0[2] - pz * e0[1], oy = pz * e0[0] - px * e0[2], oz = px * e0[1] - py * e0[0];
    const float vv = f * (rd.x * ox + rd.y * oy + rd.z * oz);
    if (vv < 0.0f || (u + vv) > 1.0f) return false;
    t = f * (e1[0] * ox + e1[1] * oy + e1[2] * oz);
    return true;
}

// Shape.intersect for every shape type; `sh` is wave-uniform (scalar loads), the ray is
// per lane.  o/d are the world-space ray; the object-space ray is r.transform(inverse)
// (src/ray.zig:13-18) with all four lanes kept.
// Shape of the code: the roots are computed under the divergent conditions (`active` — this lane
// tests this shape at all — folded in) into temporaries that default to NaN, and the visitor is
// then updated unconditionally with selects.  A NaN root changes nothing (not >= 0, not < 0), so
// the visitor's registers are never written inside a divergent region and hipcc has no phi copies
// to make at its end (they were 11 v_mov per sphere test).
template <bool C, class V>
__device__ __forceinline__ void intersect_shape(V& v, const RT_CONST rtx_shape& sh, uint32_t s, V4 o, V4 d, bool active) {
    const bool n1_visible = !(sh.flags & RTX_SHAPE_N1_INVISIBLE);
    // r.transform(inverse), src/ray.zig:13-18 (all branches wave-uniform)
    auto to_object_space = [&](V4& ro, V4& rd) {
        ro = o;
        rd = d;
        if (C && (sh.flags & kShapeDiagonal)) {  // 9 instead of 33 operations
            const float inv[16] = {sh.inverse[0], 0.0f, 0.0f, sh.inverse[3], 0.0f, sh.inverse[5], 0.0f, sh.inverse[7],
                                   0.0f, 0.0f, sh.inverse[10], sh.inverse[11], 0.0f, 0.0f, 0.0f, 1.0f};
            ro = mul_point_diag(inv, o);
            rd = mul_dir_diag(inv, d);
        } else if (!(sh.flags & kShapeIdentity)) {  // x * identity is x (up to the sign of a zero): skip it
            float inv[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) inv[k] = sh.inverse[k];
            ro = mul_point<C>(inv, o);
            rd = mul_dir<C>(inv, d);
        }
    };
    if (sh.type == RTX_PLANE) {  // src/plane.zig:13-28 — reads only the y row of the object-space ray,
        V4 ro, rd;               // so the transform sits inside the branch and its x / z rows fall away
        to_object_space(ro, rd);
        float t = NAN;
        if (active && !(fabsf(rd.y) < kEpsilon)) t = -ro.y / rd.y;
        offer1(v, s, n1_visible, t);
        return;
    }
    V4 ro, rd;
    to_object_space(ro, rd);

    switch (sh.type) {
    case RTX_SPHERE: {  // src/sphere.zig:11-54
        const V4 sphere_to_ray = sub(ro, v4(0.0f, 0.0f, 0.0f, 1.0f));
        const float a = dot<C>(rd, rd);
        const float b = 2.0f * dot<C>(rd, sphere_to_ray);
        const float c = dot<C>(sphere_to_ray, sphere_to_ray) - 1.0f;
        const float discriminant = b * b - 4.0f * a * c;
        float t0 = NAN, t1 = NAN;
        if (active && discriminant >= 0.0f) {
            const float sq = sqrtf(discriminant);
            t0 = (-b - sq) / (2.0f * a);
            t1 = (-b + sq) / (2.0f * a);
        }
        offer2(v, s, n1_visible, t0, t1);
        return;
    }
    case RTX_CUBE: {  // src/cube.zig:11-26
        const TMinMax x = check_axis(-1.0f, 1.0f, ro.x, rd.x);
        const TMinMax y = check_axis(-1.0f, 1.0f, ro.y, rd.y);
        const TMinMax z = check_axis(-1.0f, 1.0f, ro.z, rd.z);
        const float tmin = fmaxf(x.tmin, fmaxf(y.tmin, z.tmin));
        const float tmax = fminf(x.tmax, fminf(y.tmax, z.tmax));
        const bool crosses = active && !(tmin > tmax);
        offer2(v, s, n1_visible, crosses ? tmin : NAN, crosses ? tmax : NAN);
        return;
    }
    case RTX_CYLINDER:  // src/cylinder.zig:14-71
    case RTX_CONE: {    // src/cone.zig:14-82
        const bool cone = sh.type == RTX_CONE;
        float t[4] = {NAN, NAN, NAN, NAN};
        float a, b, c;
        if (cone) {
            a = rd.x * rd.x - rd.y * rd.y + rd.z * rd.z;
            b = 2.0f * ro.x * rd.x - 2.0f * ro.y * rd.y + 2.0f * ro.z * rd.z;
            c = ro.x * ro.x - ro.y * ro.y + ro.z * ro.z;
        } else {
            a = rd.x * rd.x + rd.z * rd.z;
            b = 2.0f * ro.x * rd.x + 2.0f * ro.z * rd.z;
            c = ro.x * ro.x + ro.z * ro.z - 1.0f;
        }
        bool walls = active;
        if (approx_zero(a)) {
            walls = false;
            if (active && cone && !approx_zero(b)) t[0] = -c / (2.0f * b);  // src/cone.zig:35-39: the single root
        }
        if (walls) {
            const float discriminant = b * b - 4.0f * a * c;
            if (!(discriminant < 0.0f)) {
                const float sq = sqrtf(discriminant);
                float t0 = (-b - sq) / (2.0f * a);
                float t1 = (-b + sq) / (2.0f * a);
                if (t0 > t1) {
                    const float tmp = t0;
                    t0 = t1;
                    t1 = tmp;
                }
                const float y0 = fmaf(t0, rd.y, ro.y);  // @mulAdd, cylinder.zig:40 / cone.zig:53
                if (sh.minimum < y0 && y0 < sh.maximum) t[0] = t0;
                const float y1 = fmaf(t1, rd.y, ro.y);  // @mulAdd, cylinder.zig:45 / cone.zig:58
                if (sh.minimum < y1 && y1 < sh.maximum) t[1] = t1;
            }
        }
        // intersectCaps (src/cylinder.zig:53-71, src/cone.zig:66-82)
        if (active && (sh.flags & RTX_SHAPE_CLOSED) && !approx_zero(rd.y)) {
            const float t_lower = (sh.minimum - ro.y) / rd.y;
            {
                const float x = fmaf(t_lower, rd.x, ro.x);  // checkCap, cylinder.zig:98-99 / cone.zig:111-112
                const float z = fmaf(t_lower, rd.z, ro.z);
                const float limit = cone ? fabsf(sh.minimum * sh.minimum) : 1.0f;
                if ((x * x + z * z) <= limit) t[2] = t_lower;
            }
            const float t_upper = (sh.maximum - ro.y) / rd.y;
            {
                const float x = fmaf(t_upper, rd.x, ro.x);
                const float z = fmaf(t_upper, rd.z, ro.z);
                const float limit = cone ? fabsf(sh.maximum * sh.maximum) : 1.0f;
                if ((x * x + z * z) <= limit) t[3] = t_upper;
            }
        }
        offer_roots(v, s, n1_visible, t);
        return;
    }
    default: {  // RTX_TRIANGLE
        float p0[3], e0[3], e1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            p0[k] = sh.p0[k];
            e0[k] = sh.e0[k];
            e1[k] = sh.e1[k];
        }
        float root, tt = NAN;
        if (active && triangle_root(ro, rd, p0, e0, e1, root)) tt = root;
        offer1(v, s, n1_visible, tt);
        return;
    }
    }
}

// Per-lane walk of one triangle BVH (threaded pre-order, no stack).  Only culling happens
// here; every surviving triangle goes through the reference's test and the visitor.
//
// Two alternating phases ("while-while"): every lane first walks boxes until it holds a leaf
// (or runs out of nodes), then the lanes test their leaf triangles together.  With a single
// loop the lanes that sit on a leaf would run the long triangle test while all the others,
// still walking boxes, idle through it — and the other way round.
template <bool C, class V, class T>
RTX_HD void traverse_bvh(V& v, T& tally, const Params& p, uint32_t root, V4 o, V4 d) {
    const float ix = 1.0f / d.x, iy = 1.0f / d.y, iz = 1.0f / d.z;
    const rtx_bvh::Node* nodes = p.bvh_nodes;
    // The tree comes in rtx_bvh::kLayouts threaded layouts of equal size, one per octant of the
    // ray direction, each visiting the nearer child of every split first (the predecessor walk
    // looks backwards and takes the opposite octant).
    const uint32_t stride = nodes[root].skip - root;
    uint32_t octant = (d.x < 0.0f ? 1u : 0u) | (d.y < 0.0f ? 2u : 0u) | (d.z < 0.0f ? 4u : 0u);
    if (V::kBvhPred) octant ^= 7u;
    uint32_t n = root + octant * stride;
    const uint32_t end = n + stride;
    uint32_t tri = 0, tri_end = 0;  // pending leaf: triangles [tri, tri_end)
    for (;;) {
        while (tri == tri_end && n < end && !v.done()) {
            const float4 a = *reinterpret_cast<const float4*>(&nodes[n].bmin[0]);
            const float4 b = *reinterpret_cast<const float4*>(&nodes[n].bmax[0]);
            const uint32_t skip = bits_of(a.w), leaf = bits_of(b.w);
            v.visited_node();
            tally.box();
            // slab test; a NaN (0 * inf) drops out of fmin/fmax like in the reference's checkAxis.
            // Kept as (bound - o) * (1 / d): the contracted form fma(bound, 1/d, -o/d) loses the
            // small difference bound - o against a large o / d and would cull wrongly.
            const float x0 = (a.x - o.x) * ix, x1 = (b.x - o.x) * ix;
            const float y0 = (a.y - o.y) * iy, y1 = (b.y - o.y) * iy;
            const float z0 = (a.z - o.z) * iz, z1 = (b.z - o.z) * iz;
            const float tmin = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fminf(z0, z1));
            const float tmax = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fmaxf(z0, z1));
            // Slack.  Each of x0 .. z1 is fl(fl(bound - o) * fl(1 / d)): three roundings, so it is within
            // 3u (1 + O(u)) of the exact slab parameter, relatively (u = 2^-24) — whatever the size of
            // o: a far origin makes bound - o and the parameter large together.  min / max are exact, so if
            // the exact slabs overlap (tmin <= tmax) the computed ones satisfy
            //     tmin - tmax <= 3.1 u (|tmin| + |tmax|),
            // i.e. a relative slack of 1.6 u on each side already makes the box test conservative for the
            // arithmetic of this loop.  kSlabRel = 1e-5 = 168 u leaves a factor 100 for the other inexact
            // party: the f32 triangle test (src/triangle.zig:32-50) accepts points up to a few ulps of
            // |o| + |p0| outside the exact triangle, more for rays within ~1e-3 rad of its plane, where
            // 1 / det amplifies the rounding of u and v; the leaf boxes are padded for that in the build
            // (rtx_bvh.hpp bound_triangle: relative to the box's extent and coordinates), kSlabAbs covers
            // parameters that round to zero.  Both terms are relative, so the walk culls the same way at
            // any scale or distance (tests: test_bvh_host.py, test_gpu_parity.py "large group transforms").
            constexpr float kSlabRel = 1e-5f, kSlabAbs = 1e-6f;
            const float slack = kSlabRel * (fabsf(tmin) + fabsf(tmax)) + kSlabAbs;
            const bool overlap = (tmin - slack <= tmax + slack) && (tmax + slack >= v.t_lo()) && (tmin - slack <= v.t_hi());
            if (overlap && leaf == rtx_bvh::kNoLeaf) {
                n = n + 1;  // first child
            } else {
                if (overlap) {
                    tri = leaf >> 3;
                    tri_end = tri + (leaf & 7u);
                }
                n = skip;
            }
        }
        if (tri == tri_end) break;  // no leaf pending: this lane has run out of nodes (or is done)
        while (tri < tri_end) {
            const rtx_bvh::Tri* tr = &p.bvh_tris[tri];
            const float4 q0 = *reinterpret_cast<const float4*>(&tr->p0[0]);
            const float4 q1 = *reinterpret_cast<const float4*>(&tr->e0[0]);
            const float4 q2 = *reinterpret_cast<const float4*>(&tr->e1[0]);
            const uint32_t shape = bits_of(q0.w), flags = bits_of(q1.w);
            v.tested_triangle();
            tally.tri();
            V4 ro = o, rd = d;
            if (!(flags & kShapeIdentity)) {
                const float* inv = p.shapes[shape].inverse;
                ro = mul_point<C>(inv, o);
                rd = mul_dir<C>(inv, d);
            }
            const float p0[3] = {q0.x, q0.y, q0.z}, e0[3] = {q1.x, q1.y, q1.z}, e1[3] = {q2.x, q2.y, q2.z};
            float t;
            if (triangle_root(ro, rd, p0, e0, e1, t)) {
                v.template hit<false>(t, shape);
                if (V::kBvhPred && !(flags & RTX_SHAPE_N1_INVISIBLE)) v.template pred<false>(t, shape);
            }
            ++tri;
        }
    }
}

// World.intersect + Group.intersect (src/world.zig:38-47, src/group.zig:55-87) over the
// flattened node/shape arrays.  The node and shape loops are wave-uniform (records come in
// through scalar loads); a lane whose ray failed a group's box is parked until the loop
// leaves that group's subtree.  `lane_on` is false for padding lanes.  A node whose triangle
// run has a BVH tests only its `n_linear` leading shapes linearly.
template <bool C, bool BVH, class V, class T>
__device__ __forceinline__ void traverse(V& v, T& tally, const Params& p, V4 o, V4 d, bool lane_on) {
    uint32_t parked_until = lane_on ? 0u : kNone;  // node index at which this lane wakes up
    const RT_CONST rtx_node* nodes = as_const(p.nodes);
    const RT_CONST rtx_shape* shapes = as_const(p.shapes);
    for (uint32_t n = 0; n < p.n_nodes; ++n) {
        const RT_CONST rtx_node& node = nodes[n];
        bool active = n >= parked_until && !v.done();
        tally.gates(active && node.has_bounds);
        if (active && node.has_bounds && !bounds_hit(node, o, d)) {
            parked_until = node.skip;
            active = false;
        }
        if (__ballot(active) == 0ull) continue;
        const uint32_t bvh_root = BVH ? node.reserved[0] : 0u;  // root + 1, 0 = no BVH
        const uint32_t first = node.first_shape;
        const uint32_t last = first + (BVH ? node.reserved[1] : node.n_shapes);  // the rest sits in a tree (of this or a later node)
        for (uint32_t s = first; s < last; ++s) {
            tally.shapes(shapes[s].type, active);
            intersect_shape<C>(v, shapes[s], s, o, d, active);
        }
        if (BVH && bvh_root && active) traverse_bvh<C>(v, tally, p, bvh_root - 1u, o, d);
    }
}

// The gates again, for the BVH triangles only: the predecessor walk of the lanes in `lane_on`.
template <bool C, class PV, class T>
__device__ __forceinline__ void traverse_pred(PV& v, T& tally, const Params& p, V4 o, V4 d, bool lane_on) {
    uint32_t parked_until = lane_on ? 0u : kNone;
    const RT_CONST rtx_node* nodes = as_const(p.nodes);
    for (uint32_t n = 0; n < p.n_nodes; ++n) {
        const RT_CONST rtx_node& node = nodes[n];
        bool active = n >= parked_until;
        tally.gates(active && node.has_bounds);
        if (active && node.has_bounds && !bounds_hit(node, o, d)) {
            parked_until = node.skip;
            active = false;
        }
        const uint32_t bvh_root = node.reserved[0];
        if (bvh_root && active) traverse_bvh<C>(v, tally, p, bvh_root - 1u, o, d);
    }
}

// ------------------------------------------------------------------------ shading

// Shape.normalAt for the hit shape (per-lane record).
// world -> object point for the per-lane hit shape; the diagonal form when the shape allows it
// (a per-lane branch: waves whose hits all sit on scaled / translated shapes skip the general one).
// `bits` are the hit record's flags: an identity shape's full record is never read.
template <bool C>
__device__ __forceinline__ V4 to_object(uint32_t bits, const rtx_shape& sh, V4 world_point) {
    if (bits & kHitIdentity) return world_point;
    if (C && (bits & kHitDiagonal)) return mul_point_diag(sh.inverse, world_point);
    return mul_point<C>(sh.inverse, world_point);
}
template <bool C>
__device__ __forceinline__ V4 normal_to_world(uint32_t bits, const rtx_shape& sh, V4 n) {
    if (bits & kHitIdentity) return n;
    if (C && (bits & kHitDiagonal)) return mul_dir_diag(sh.inverse, n);  // a diagonal block is its own transpose
    return mul_mat_t<C>(sh.inverse, n);
}

// Planes and triangles have one normal (src/plane.zig:30-46, src/triangle.zig:52-61): the scene
// upload evaluates normalize(transpose(inverse) * n, w = 0) once per shape with these same f32
// operations on the host (world_normal_of) and every hit reads the result.
__host__ inline void world_normal_of(const rtx_shape& sh, float out[4]) {
    V4 n = sh.type == RTX_PLANE ? v4(0.0f, 1.0f, 0.0f, 0.0f) : v4(sh.normal[0], sh.normal[1], sh.normal[2], sh.normal[3]);
    V4 w = mul_mat_t<false>(sh.inverse, n);
    w.w = 0.0f;
    w = normalize<false>(w);
    out[0] = w.x;
    out[1] = w.y;
    out[2] = w.z;
    out[3] = w.w;
}

// Shapes whose normal depends on the point (spheres, cubes, cylinders, cones); planes and triangles
// are answered from the hit record by the caller.
template <bool C>
__device__ __forceinline__ V4 normal_at(uint32_t bits, const rtx_shape& sh, V4 world_point) {
    V4 object_normal;
    switch ((bits >> kHitTypeShift) & 7u) {
    case RTX_SPHERE: {  // src/sphere.zig:56-82
        const V4 object_point = to_object<C>(bits, sh, world_point);
        object_normal = sub(object_point, v4(0.0f, 0.0f, 0.0f, 1.0f));
        break;
    }
    case RTX_PLANE:  // src/plane.zig:30-46
        object_normal = v4(0.0f, 1.0f, 0.0f, 0.0f);
        break;
    case RTX_CUBE: {  // src/cube.zig:28-44 — w is NOT reset before normalising
        const V4 object_point = to_object<C>(bits, sh, world_point);
        const float x = fabsf(object_point.x), y = fabsf(object_point.y), z = fabsf(object_point.z);
        const float maxc = fmaxf(x, fmaxf(y, z));
        if (maxc == x) object_normal = v4(object_point.x, 0.0f, 0.0f, 0.0f);
        else if (maxc == y) object_normal = v4(0.0f, object_point.y, 0.0f, 0.0f);
        else object_normal = v4(0.0f, 0.0f, object_point.z, 0.0f);
        return normalize<C>(normal_to_world<C>(bits, sh, object_normal));
    }
    case RTX_CYLINDER: {  // src/cylinder.zig:73-92
        const V4 object_point = to_object<C>(bits, sh, world_point);
        const float distance = object_point.x * object_point.x + object_point.z * object_point.z;
        if (distance < 1.0f && object_point.y >= sh.maximum - kEpsilon) object_normal = v4(0.0f, 1.0f, 0.0f, 0.0f);
        else if (distance < 1.0f && object_point.y <= sh.minimum + kEpsilon) object_normal = v4(0.0f, -1.0f, 0.0f, 0.0f);
        else object_normal = v4(object_point.x, 0.0f, object_point.z, 0.0f);
        break;
    }
    case RTX_CONE: {  // src/cone.zig:84-105
        const V4 object_point = to_object<C>(bits, sh, world_point);
        const float distance = object_point.x * object_point.x + object_point.z * object_point.z;
        if (distance < sh.maximum * sh.maximum && object_point.y >= sh.maximum - kEpsilon) {
            object_normal = v4(0.0f, 1.0f, 0.0f, 0.0f);
        } else if (distance < sh.minimum * sh.minimum && object_point.y <= sh.minimum + kEpsilon) {
            object_normal = v4(0.0f, -1.0f, 0.0f, 0.0f);
        } else {
            const float y = (object_point.y > 0.0f) ? -sqrtf(distance) : sqrtf(distance);
            object_normal = v4(object_point.x, y, object_point.z, 0.0f);
        }
        break;
    }
    default:  // src/triangle.zig:52-61
        object_normal = v4(sh.normal[0], sh.normal[1], sh.normal[2], sh.normal[3]);
        break;
    }
    V4 world_normal = normal_to_world<C>(bits, sh, object_normal);
    world_normal.w = 0.0f;
    return normalize<C>(world_normal);
}

// @mod(x, 2) == 0 for the integer-valued x the patterns pass (a floor, or a sum of floors): x is even
// <=> x / 2 is an integer; halving is exact, every float from 2^24 up is an even integer, NaN compares
// false and an infinity (whose @mod is NaN) is excluded.  4 instructions; fmodf(x, 2) == 0 was ~30.
__device__ __forceinline__ bool mod2_is_zero(float x) {
    const float h = x * 0.5f;
    return h == truncf(h) && fabsf(x) < INFINITY;
}

struct Color {
    float r, g, b;
};

// Pattern.atShape (src/pattern.zig:23-51)
template <bool C>
__device__ __forceinline__ Color pattern_at_shape(const rtx_pattern& pat, uint32_t bits, const rtx_shape& sh, V4 world_point) {
    const V4 object_point = to_object<C>(bits, sh, world_point);
    const V4 pt = mul_point<C>(pat.inverse, object_point);  // only x, y, z are read
    const Color a{pat.a[0], pat.a[1], pat.a[2]}, b{pat.b[0], pat.b[1], pat.b[2]};
    switch (pat.color_map) {
    case RT_PATTERN_STRIPE:
        return mod2_is_zero(floorf(pt.x)) ? a : b;
    case RT_PATTERN_GRADIENT: {
        const float fraction = pt.x - floorf(pt.x);
        return Color{a.r + (b.r - a.r) * fraction, a.g + (b.g - a.g) * fraction, a.b + (b.b - a.b) * fraction};
    }
    case RT_PATTERN_RING:
        return mod2_is_zero(floorf(sqrtf(pt.x * pt.x + pt.z * pt.z))) ? a : b;
    default:
        return mod2_is_zero(floorf(pt.x) + floorf(pt.y) + floorf(pt.z)) ? a : b;
    }
}

// std.math.pow(f32, x, y) for the case every sample hits — y a small positive integer (shininess
// 200 / 300, Schlick's 5), the same for every active lane, x a positive normal number below 2 — as
// rt::zmath::pow_f32 evaluates it (squaring loop on the frexp'd significand, src/material.zig:96,
// src/intersection.zig:179), with the loop control on the scalar unit and the renormalisation
// `if (x1 < 0.5) { x1 += x1; xe -= 1; }` as a frexp of the square (x1 * x1 is in [0.25, 1): same
// value, same exponent step).  pow_f32's overflow exit (|xe| > 512: ae += xe; break) can only fire
// with xe < -512 here, which sends the result to ldexp(a1 <= 1, < -512) = +0: lanes that reach it
// are marked and return 0, the others never see the exit.  Any lane outside that domain sends the
// whole wave to the general routine.  ~55 VALU instructions for y = 200 instead of ~130.
__device__ __forceinline__ float pow_lanes(float x, float y) {
    const int yi = static_cast<int>(y);
    const bool simple = (y >= 2.0f) & (y < 1048576.0f) & (static_cast<float>(yi) == y) & (x >= 1.17549435e-38f) & (x < 2.0f);
    const int y0 = __builtin_amdgcn_readfirstlane(yi);
    if (__ballot(simple && yi == y0) != __ballot(true)) return rt::zmath::pow_f32(x, y);
    float x1 = __builtin_amdgcn_frexp_mantf(x);
    uint32_t xe = static_cast<uint32_t>(__builtin_amdgcn_frexp_expf(x));  // two's complement, wraps harmlessly in dead lanes
    float a1 = 1.0f;
    uint32_t ae = 0u;
    bool dead = false;
    for (uint32_t i = static_cast<uint32_t>(y0); i != 0u; i >>= 1) {  // wave-uniform
        dead = dead | (xe + 512u > 1024u);  // xe < -512 || 512 < xe
        if (i & 1u) {
            a1 *= x1;
            ae += xe;
        }
        const float sq = x1 * x1;
        x1 = __builtin_amdgcn_frexp_mantf(sq);
        xe = (xe << 1) + static_cast<uint32_t>(__builtin_amdgcn_frexp_expf(sq));
    }
    return dead ? 0.0f : ldexpf(a1, static_cast<int>(ae));
}

// Material.lighting (src/material.zig:40-102)
template <bool C>
__device__ __forceinline__ Color lighting(const Params& p, const rtx_pattern* patterns, const rtx_material& m, uint32_t bits,
                                          const rtx_shape& sh, V4 point, V4 light_direction, V4 eye, V4 normal, bool in_shadow) {
    Color color;
    if (m.pattern >= 0) color = pattern_at_shape<C>(patterns[m.pattern], bits, sh, point);
    else color = Color{m.color[0], m.color[1], m.color[2]};

    const Color li{p.light_intensity[0], p.light_intensity[1], p.light_intensity[2]};
    const Color effective{color.r * li.r, color.g * li.g, color.b * li.b};
    // light_direction = normalize(light.position - point) (src/material.zig:58) comes from the caller:
    // the shadow ray's direction is the same expression on the same point (src/world.zig:100-106).
    const Color ambient{effective.r * m.ambient, effective.g * m.ambient, effective.b * m.ambient};
    if (in_shadow) return ambient;

    Color diffuse{0.0f, 0.0f, 0.0f}, specular{0.0f, 0.0f, 0.0f};
    const float cos_light_normal = dot<C>(light_direction, normal);
    if (!(cos_light_normal < 0.0f)) {
        diffuse = Color{effective.r * m.diffuse * cos_light_normal, effective.g * m.diffuse * cos_light_normal,
                        effective.b * m.diffuse * cos_light_normal};
        // reflect(-L, N) = -L - N * dot(-L, N) * 2 (src/tuple.zig:58-77); dot(-L, N) is exactly -dot(L, N)
        const V4 refl = sub(neg(light_direction), muls(muls(normal, -cos_light_normal), 2.0f));
        const float cos_reflect_eye = dot<C>(refl, eye);
        if (!(cos_reflect_eye <= 0.0f)) {
            const float factor = pow_lanes(cos_reflect_eye, m.shininess);
            specular = Color{li.r * m.specular * factor, li.g * m.specular * factor, li.b * m.specular * factor};
        }
    }
    return Color{ambient.r + diffuse.r + specular.r, ambient.g + diffuse.g + specular.g, ambient.b + diffuse.b + specular.b};
}

// shadeHitInternal's final expression (src/world.zig:71-78)
__device__ __forceinline__ Color combine(Color surface, Color reflected, Color refracted, bool use_schlick, float reflectance) {
    if (use_schlick) {
        const float k = 1.0f - reflectance;
        return Color{surface.r + reflected.r * reflectance + refracted.r * k,
                     surface.g + reflected.g * reflectance + refracted.g * k,
                     surface.b + reflected.b * reflectance + refracted.b * k};
    }
    return Color{surface.r + reflected.r + refracted.r, surface.g + reflected.g + refracted.g,
                 surface.b + reflected.b + refracted.b};
}

// ------------------------------------------------------------- queues and delivery

// Slots for this wave's interior nodes and child rays with ONE returning atomic: the shard's
// per-level word packs (nodes << 32) | rays.  Returns the shard-local bases.
struct WaveAlloc {
    uint32_t refl_base, refr_base, node_base;
};
// Reflected and refracted children go to separate regions of the shard's queue (reflected from
// the bottom up, refracted from the top down): a wave of the next level then reads rays of one
// kind only — rays that left a surface and rays that entered it meet different shapes and
// materials, and mixed in one wave they would serialise each other's shading branches.
__device__ __forceinline__ WaveAlloc wave_alloc(ShardCounters* sc, uint32_t level, uint32_t n_refl, uint32_t n_refr, uint32_t n_nodes) {
    WaveAlloc a{0u, 0u, 0u};
    if (n_refl == 0u && n_refr == 0u && n_nodes == 0u) return a;  // wave-uniform
    unsigned long long old = 0ull;
    uint32_t old_refr = 0u;
    if (__lane_id() == 0u) {
        if (n_refl != 0u || n_nodes != 0u) old = atomicAdd(&sc->alloc[level], (static_cast<unsigned long long>(n_nodes) << 32) | n_refl);
        if (n_refr != 0u) old_refr = atomicAdd(&sc->refr[level], n_refr);
    }
    a.refl_base = __shfl(static_cast<uint32_t>(old), 0);
    a.node_base = __shfl(static_cast<uint32_t>(old >> 32), 0);
    a.refr_base = __shfl(old_refr, 0);
    return a;
}

// Hand a finished colour to whoever asked for it: the canvas at the root
// (canvas.writePixel, src/canvas.zig:34-36), otherwise the parent, scaled by the parent's
// reflective / transparency (src/world.zig:142,185).
// A parent with two children keeps the colour in its reflected / refracted slot until k_combine
// has both.  A parent whose only child this is needs no slot and no combine pass: its other
// colour is the black of src/world.zig:131,151, so shadeHit's sum (src/world.zig:71-78) can be
// evaluated right here — the same f32 operations on the same values as k_combine's — and the
// result handed on to ITS parent, and so on up a chain of single-child nodes (most interior
// nodes of a scene with mirrors are such nodes: a reflective floor has one child per hit).
__device__ __forceinline__ void deliver(const Params& p, bool root, uint32_t link, float weight, Color c) {
    const Color black{0.0f, 0.0f, 0.0f};
    while (!root) {
        const uint32_t parent = link & kLinkIndexMask;
        const Color cw{c.r * weight, c.g * weight, c.b * weight};
        if (!(link & kLinkSingleBit)) {
            // one 16-byte store per child: three scattered dword stores cost a 32-byte partial write each
            float4* slot = (link & kLinkSlotBit) ? &p.node_c[parent] : &p.node_b[parent];
            *slot = make_float4(cw.r, cw.g, cw.b, 0.0f);
            return;
        }
        const float4 a = p.node_a[parent];
        const uint2 dd = p.node_d[parent];
        const bool refracted = (link & kLinkSlotBit) != 0u;
        c = combine(Color{a.x, a.y, a.z}, refracted ? black : cw, refracted ? cw : black, (dd.x & kLinkSchlickBit) != 0u, a.w);
        root = parent < p.node_off[1];  // a node of level 0 answers to the canvas
        link = dd.x;
        weight = __uint_as_float(dd.y);
    }
    float* px = p.canvas + 3ull * (link & kLinkIndexMask);
    px[0] = c.r;
    px[1] = c.g;
    px[2] = c.b;
}

// ------------------------------------------------------------------------ kernels

// Diagnostic build only (-DRTX_DIAG, `make diag`): per-phase cycle shares of k_trace_shade,
// every phase fenced with a full s_waitcnt so a wait is charged to the phase that issued the
// memory operation.  Never compiled into the product library; timings of this build are
// meaningless, only the shares are read (tools/diag_phases.py).
#ifdef RTX_DIAG
__device__ unsigned long long g_diag[16];
#define DIAG_DECL unsigned long long diag_t = 0, diag_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define DIAG_START()                                                     \
    do {                                                                 \
        __builtin_amdgcn_s_waitcnt(0);                                   \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
        diag_t = __builtin_readcyclecounter();                           \
    } while (0)
#define DIAG_PHASE(k)                                                    \
    do {                                                                 \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      \
        const unsigned long long now_ = __builtin_readcyclecounter();    \
        diag_acc[k] += now_ - diag_t;                                    \
        diag_t = now_;                                                   \
    } while (0)
#define DIAG_FLUSH()                                                     \
    do {                                                                 \
        if ((threadIdx.x & 63u) == 0u)                                   \
            for (int k_ = 0; k_ < 8; ++k_) atomicAdd(&g_diag[k_], diag_acc[k_]); \
    } while (0)
#else
#define DIAG_DECL
#define DIAG_START()
#define DIAG_PHASE(k)
#define DIAG_FLUSH()
#endif

// Scene tables staged in LDS (CDNA4: 160 KB per CU).  The hit shape, its material and its
// pattern are per-lane lookups (normalAt, lighting, n1/n2); from global memory each is a
// dependent L2 round trip per lane and the diagnostic build showed 65 % of the kernel waiting
// on them, so scenes whose tables fit kLdsSceneBytes are copied into LDS once per block.
constexpr uint32_t kLdsSceneBytes = 32768;  // 5 blocks per CU (the mesh kernels' occupancy) fill the 160 KB exactly

struct LdsScene {
    const rtx_shape* shapes;
    const HitRec* hit;
    const rtx_material* materials;
    const rtx_pattern* patterns;
};
// Which per-lane tables a kernel stages in LDS (template parameter TAB):
constexpr int kTabGlobal = 0;  // none: every per-lane record comes from global memory (meshes beyond ~2000 triangles)
constexpr int kTabAll = 1;     // hit records, shape records, materials, patterns (analytic scenes, meshes up to ~180 triangles)
constexpr int kTabHit = 2;     // hit records, materials, patterns; the 160-byte shape records stay in global memory and
                               // are read only for hits whose normal or pattern needs the inverse (meshes up to ~2000 triangles)

// One ray-tree level.  GEN: level-0 rays come from the camera (rayForPixel,
// src/camera.zig:70-82) instead of the queue.  TRACK: keep the n1 predecessor.  TAB: which scene
// tables are staged in LDS (kTab*, p.lds_bytes of dynamic shared memory).  C: canonical w lanes (see
// dot<C>); rays then travel as two float4 (origin.xyz + link, direction.xyz + weight = 32 B)
// instead of origin, direction and a link/weight pair (40 B).  BVH: some gate carries a triangle
// BVH (scenes without one get kernels without the traversal code and its registers).
// Register budget: 5 waves per SIMD (<= 96 VGPRs) for kernels that carry the BVH walk; measured
// on configs[1] when every kernel carried it (with -fno-slp-vectorize, see the Makefile): 4 waves
// 1.47 ms/step, 5 waves 1.39, 6 waves (spills) 1.41, 8 waves (spills) 1.62.  Kernels without
// the BVH code need 75-80 VGPRs when they also leave the w lanes out (C) and get
// RTX_WAVES_PER_EU_NOBVH; the four-lane ones would spill there and stay at 5.
#ifndef RTX_WAVES_PER_EU
#define RTX_WAVES_PER_EU 5
#endif
#ifndef RTX_WAVES_PER_EU_NOBVH
#define RTX_WAVES_PER_EU_NOBVH 6
#endif
template <int TAB, bool GEN, bool TRACK, bool C, bool BVH, bool STATS, bool BOOK = false>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu((BVH || !C) ? RTX_WAVES_PER_EU : RTX_WAVES_PER_EU_NOBVH,
                                   (BVH || !C) ? RTX_WAVES_PER_EU : RTX_WAVES_PER_EU_NOBVH))) void k_trace_shade(const Params p, const uint32_t level) {
    extern __shared__ uint4 lds_raw[];
    LdsScene tab{p.shapes, p.hitrecs, p.materials, p.patterns};
    if (TAB != kTabGlobal) {
        // [hit records][shapes, if TAB == kTabAll][materials][patterns], all record sizes are multiples of 16 bytes.
        const uint32_t n_hr = p.n_shapes * (uint32_t)(sizeof(HitRec) / 16);
        const uint32_t n_sh = TAB == kTabAll ? p.n_shapes * (uint32_t)(sizeof(rtx_shape) / 16) : 0u;
        const uint32_t n_ma = p.n_materials * (uint32_t)(sizeof(rtx_material) / 16), n_pa = p.n_patterns * (uint32_t)(sizeof(rtx_pattern) / 16);
        const uint4* g_hr = reinterpret_cast<const uint4*>(p.hitrecs);
        const uint4* g_sh = reinterpret_cast<const uint4*>(p.shapes);
        const uint4* g_ma = reinterpret_cast<const uint4*>(p.materials);
        const uint4* g_pa = reinterpret_cast<const uint4*>(p.patterns);
        for (uint32_t k = threadIdx.x; k < n_hr; k += blockDim.x) lds_raw[k] = g_hr[k];
        for (uint32_t k = threadIdx.x; k < n_sh; k += blockDim.x) lds_raw[n_hr + k] = g_sh[k];
        for (uint32_t k = threadIdx.x; k < n_ma; k += blockDim.x) lds_raw[n_hr + n_sh + k] = g_ma[k];
        for (uint32_t k = threadIdx.x; k < n_pa; k += blockDim.x) lds_raw[n_hr + n_sh + n_ma + k] = g_pa[k];
        __syncthreads();
        tab.hit = reinterpret_cast<const HitRec*>(lds_raw);
        if (TAB == kTabAll) tab.shapes = reinterpret_cast<const rtx_shape*>(lds_raw + n_hr);
        tab.materials = reinterpret_cast<const rtx_material*>(lds_raw + n_hr + n_sh);
        tab.patterns = reinterpret_cast<const rtx_pattern*>(lds_raw + n_hr + n_sh + n_ma);
    }
    const bool root = level == 0;
    // Two counter sets alternate between batches: while this batch allocates from one, its first
    // launch clears the other for the batch after it (everything that used that set finished
    // before this kernel started), so frames that do not ask for statistics need neither a fill
    // nor k_accumulate between batches.
    static_assert(sizeof(Counters) == 3u * kShards * 128u && 3u * kShards <= 16u * 256u, "one thread of the first blocks per line");
    if (root && blockIdx.x * 256u + threadIdx.x < 3u * kShards) {
        // lines [0, kShards) are the shard counters, then the claim lines, then the statistics lines (struct Counters)
        uint4* line = reinterpret_cast<uint4*>(p.counters_next) + (size_t)(blockIdx.x * 256u + threadIdx.x) * (sizeof(ShardCounters) / sizeof(uint4));
#pragma unroll
        for (int k = 0; k < (int)(sizeof(ShardCounters) / sizeof(uint4)); ++k) line[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    const uint32_t remaining = p.depth - level;  // `remaining` of colorAtInternal (src/world.zig:85)
    const uint32_t qi = level & 1u, qo = qi ^ 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t shard = wave % kShards;
    ShardCounters* sc = &p.counters->shard[shard];

    // Level 0 covers all items in chunks of 64, chunk c belonging to shard c % kShards; deeper
    // levels cover this wave's shard: its region holds reflected rays [0, n_refl_in) from its
    // bottom and refracted rays from its top down, and chunks of 64 cover first the one, then
    // the other.  `count` is the end of the item range the waves of this shard claim chunks from.
    uint32_t count, in_base, n_refl_in = 0u, n_refr_in = 0u, refl_span = 0u, in_top = 0u;
    if (root) {
        count = GEN ? p.tiles_x * ((p.n_rows + 7u) / 8u) * 64u : p.root_count;
        in_base = 0u;
    } else {
        n_refl_in = static_cast<uint32_t>(sc->alloc[level - 1]);
        n_refr_in = sc->refr[level - 1];
        refl_span = (n_refl_in + 63u) & ~63u;
        count = refl_span + ((n_refr_in + 63u) & ~63u);
        const uint32_t cap_in = p.shard_cap0 << (level * p.level_shift);
        in_base = shard * cap_in;
        in_top = in_base + cap_in - 1u;
    }
    const uint32_t cap_out = p.shard_cap0 << ((level + 1u) * p.level_shift);
    const uint32_t out_base = shard * cap_out;
    const uint32_t out_top = out_base + cap_out - 1u;
    const uint32_t node_base = p.node_off[level] + shard * (p.shard_cap0 << (level * p.level_shift));

    uint32_t local_hits = 0, local_patterned = 0, local_transparent = 0;
    Tally<STATS> tally;
    DIAG_DECL
    // Chunks of 64 rays are handed out on demand — one returning atomic on the shard's line per
    // iteration — instead of dealt in advance (wave w: chunks w, w + n_waves, ...): with the static
    // deal a wave whose tiles cover the expensive part of the image finishes long after the others
    // and the launch ends with most CUs idle.
    auto next_base = [&]() -> uint32_t {
        uint32_t k = 0u;
        if (lane == 0u) k = atomicAdd(&p.counters->claims[shard].claim[level], 1u);
        k = __shfl(k, 0);
        return root ? (k * kShards + shard) * 64u : k * 64u;  // level 0: chunk c belongs to shard c % kShards
    };
    for (uint32_t base = next_base(); base < count; base = next_base()) {
        DIAG_START();
        const uint32_t i = base + lane;
        bool on = i < count;
        V4 o, d;
        uint32_t link = 0;
        float weight = 1.0f;
        if (GEN) {
            // 8x8 pixel tile per wave over this batch's local rows.
            const uint32_t tile = i >> 6;
            const uint32_t tx = tile % p.tiles_x, ty = tile / p.tiles_x;
            const uint32_t px = tx * 8u + (lane & 7u);
            const uint32_t lrow = p.row0 + ty * 8u + (lane >> 3);  // local row among the rows this call renders
            on = on && px < p.width && lrow < p.row0 + p.n_rows;
            uint32_t py = lrow;
            if (p.strip_count > 1u) {
                const uint32_t ls = lrow / p.strip_rows;
                py = (ls * p.strip_count + p.strip_index) * p.strip_rows + (lrow - ls * p.strip_rows);
            }
            const float x_offset = (static_cast<float>(px) + 0.5f) * p.pixel_size;
            const float y_offset = (static_cast<float>(py) + 0.5f) * p.pixel_size;
            const float world_x = p.half_width - x_offset;
            const float world_y = p.half_height - y_offset;
            const V4 pixel = mul_mat(p.cam_inv, v4(world_x, world_y, -1.0f, 1.0f));
            o = mul_mat(p.cam_inv, v4(0.0f, 0.0f, 0.0f, 1.0f));
            d = normalize<C>(sub(pixel, o));
            link = (p.compact_rows ? lrow : py) * p.width + px;
        } else {
            uint32_t at = i;  // level 0 of rtx_color_at: one region from slot 0
            if (!root) {
                const bool in_refl = base < refl_span;  // wave-uniform
                on = in_refl ? (i < n_refl_in) : (i - refl_span < n_refr_in);
                at = in_refl ? in_base + i : in_top - (i - refl_span);
            }
            if (on) {
                const float4 qo4 = p.q_org[qi][at];
                const float4 qd4 = p.q_dir[qi][at];
                if (C) {
                    o = v4(qo4.x, qo4.y, qo4.z, 1.0f);
                    d = v4(qd4.x, qd4.y, qd4.z, 0.0f);
                    link = __float_as_uint(qo4.w);
                    weight = qd4.w;
                } else {
                    const uint2 ql = p.q_link[qi][at];
                    o = v4(qo4.x, qo4.y, qo4.z, qo4.w);
                    d = v4(qd4.x, qd4.y, qd4.z, qd4.w);
                    link = ql.x;
                    weight = __uint_as_float(ql.y);
                }
            } else {
                o = v4(0.0f, 0.0f, 0.0f, 1.0f);
                d = v4(0.0f, 0.0f, 1.0f, 0.0f);
            }
        }

        DIAG_PHASE(0);  // ray fetch / generation
        // colorAtInternal: world.intersect + hit (src/world.zig:85-95)
        static_assert(!BOOK || TRACK, "the book container walk is a form of n1 tracking");
        ClosestVisitor<TRACK, BOOK> cv;
        cv.init();
        traverse<C, BVH>(cv, tally, p, o, d, on);
        const bool is_hit = on && cv.c.idx != kNone;
        if (TRACK && BVH) {
            // n1 matters only where the hit material is transparent (src/world.zig:151,
            // src/intersection.zig:165): only those lanes look for a predecessor among the meshes.
            const bool need = is_hit && tab.materials[tab.hit[cv.c.idx].bits & kHitMaterialMask].transparency != 0.0f;
            if (__ballot(need) != 0ull) {
                PredVisitorT<BOOK> pv{cv.c.pred_t, cv.c.pred_idx, cv.pred2_t, cv.pred2_idx};
                traverse_pred<C>(pv, tally, p, o, d, need);
                if (need) {
                    cv.c.pred_t = pv.pred_t;
                    cv.c.pred_idx = pv.pred_idx;
                    if (BOOK) {
                        cv.pred2_t = pv.pred2_t;
                        cv.pred2_idx = pv.pred2_idx;
                    }
                }
            }
        }
        DIAG_PHASE(1);  // closest-hit traversal

        // prepareComputations (src/intersection.zig:43-90) and the shadow ray
        // (src/world.zig:55,97-106), per lane.
        V4 point = o, normal = d, over_point = o, shadow_dir = d;
        float n1 = 1.0f, n2 = 1.0f, shadow_distance = 0.0f, cos_i = 0.0f;
        HitRec hr{{0.0f, 0.0f, 0.0f}, 0u};
        if (is_hit) {
            ++local_hits;
            hr = tab.hit[cv.c.idx];
            point = add(o, muls(d, cv.c.t));  // ray.position, src/ray.zig:9-11
            // Planes and triangles have one normal (finished at upload, world_normal_of); the others
            // evaluate normalAt on their full record.
            if (hr.bits & kHitWorldNormal) normal = v4(hr.n[0], hr.n[1], hr.n[2], 0.0f);
            else normal = normal_at<C>(hr.bits, tab.shapes[cv.c.idx], point);
            // dot(normal, eye) decides the flip (src/intersection.zig:79); the dots the shading takes
            // later with the final normal — dot(eye, normal), dot(ray.direction, normal) — are this
            // value or its exact negation (products commute, negation commutes with rounding).
            const float dn = dot<C>(normal, neg(d));
            cos_i = dn;
            if (dn < 0.0f) {
                normal = neg(normal);
                cos_i = -dn;
            }
            // n1 / n2 (src/intersection.zig:92-159, see Closest above)
            if (TRACK && cv.c.pred_idx != kNone) n1 = tab.materials[tab.hit[cv.c.pred_idx].bits & kHitMaterialMask].refractive_index;
            n2 = tab.materials[hr.bits & kHitMaterialMask].refractive_index;
            if (BOOK && cv.hit_contained) {
                // the hit leaves its own shape: what remains is the runner-up if the shape was the last
                // container, the unchanged last one otherwise, 1.0 if nothing is left
                n2 = n1;
                if (cv.c.pred_idx == cv.c.idx)
                    n2 = cv.pred2_idx != kNone ? tab.materials[tab.hit[cv.pred2_idx].bits & kHitMaterialMask].refractive_index : 1.0f;
            }
            over_point = add(point, muls(normal, p.over_eps));
            const V4 lp = v4(p.light_position[0], p.light_position[1], p.light_position[2], p.light_position[3]);
            const V4 to_light = sub(lp, over_point);
            shadow_distance = magnitude<C>(to_light);
            shadow_dir = normalize<C>(to_light);
        }

        DIAG_PHASE(2);  // prepareComputations + shadow-ray set-up
        // isShadowed (src/world.zig:97-120).  The traversal loop is wave-uniform, so it runs
        // outside the divergent branch; lanes without a hit are parked.
        ShadowVisitor sv;
        sv.distance = shadow_distance;
        sv.found = false;
        traverse<C, BVH>(sv, tally, p, over_point, shadow_dir, is_hit);
        DIAG_PHASE(3);  // shadow traversal

        bool spawn_reflect = false, spawn_refract = false, use_schlick = false;
        float reflectance = 0.0f, reflective = 0.0f, transparency = 0.0f;
        Color surface{0.0f, 0.0f, 0.0f};
        V4 reflectv = d, under_point = o, refract_dir = d;
        if (is_hit) {
            const rtx_shape& sh = tab.shapes[cv.c.idx];  // read only if a pattern needs the shape's inverse
            const rtx_material& m = tab.materials[hr.bits & kHitMaterialMask];
            const V4 eye = neg(d);
            // reflect(ray.direction, normal) (src/intersection.zig:88): dot(direction, normal) == -cos_i
            reflectv = sub(d, muls(muls(normal, -cos_i), 2.0f));
            reflective = m.reflective;
            transparency = m.transparency;
            local_patterned += m.pattern >= 0 ? 1u : 0u;
            local_transparent += (m.transparency != 0.0f && remaining != 0u) ? 1u : 0u;  // refraction set-up evaluated
            // shadeHitInternal (src/world.zig:54-79): Phong is evaluated at over_point.
            surface = lighting<C>(p, tab.patterns, m, hr.bits, sh, over_point, shadow_dir, eye, normal, sv.found);
            // reflectedColorInternal (src/world.zig:126-143)
            spawn_reflect = remaining != 0u && m.reflective != 0.0f;
            // refractedColorInternal (src/world.zig:149-186) and schlick (src/intersection.zig:161-180)
            // take the same n1 / n2, sin2_t and sqrt(1 - sin2_t): evaluated once for both.
            const bool refracts = remaining != 0u && m.transparency != 0.0f;
            use_schlick = m.reflective > 0.0f && m.transparency > 0.0f;
            // At the depth limit the reflectance only weighs two black children: 0 * R and
            // 0 * (1 - R) are 0 for any finite R, so a scene whose Schlick values are all
            // finite (p.finite_schlick) need not evaluate it there.
            const bool eval_schlick = use_schlick && (remaining != 0u || !p.finite_schlick);
            const bool denser = n1 > n2;
            float n_ratio = 1.0f, cos_t = 0.0f;
            bool total_internal = false;
            if (refracts || (eval_schlick && denser)) {
                n_ratio = n1 / n2;
                const float sin2_t = (n_ratio * n_ratio) * (1.0f - cos_i * cos_i);
                total_internal = sin2_t > 1.0f;
                if (!total_internal) cos_t = sqrtf(1.0f - sin2_t);
            }
            if (refracts && !total_internal) {
                spawn_refract = true;
                under_point = sub(point, muls(normal, p.over_eps));
                refract_dir = sub(muls(normal, n_ratio * cos_i - cos_t), muls(eye, n_ratio));
            }
            if (eval_schlick) {
                if (denser && total_internal) {
                    reflectance = 1.0f;
                } else {
                    float r0 = (n1 - n2) / (n1 + n2);
                    r0 *= r0;
                    const float cosv = denser ? cos_t : cos_i;
                    reflectance = r0 + (1.0f - r0) * pow_lanes(1.0f - cosv, 5.0f);
                }
            }
        }

        DIAG_PHASE(4);  // lighting, refraction set-up, Schlick
        // Leaves finish here; interior nodes park their surface colour for k_combine.
        const bool interior = spawn_reflect || spawn_refract;
        if (on && !interior) {
            // A miss is black (src/world.zig:94); a hit without children adds two blacks.
            const Color black{0.0f, 0.0f, 0.0f};
            const Color out = is_hit ? combine(surface, black, black, use_schlick, reflectance) : black;
            deliver(p, root, link, weight, out);
        }
        // One allocation per wave: reflected children first, refracted ones after them.
        const unsigned long long m_node = __ballot(interior), m_refl = __ballot(spawn_reflect), m_refr = __ballot(spawn_refract);
        const unsigned long long below = (1ull << lane) - 1ull;
        const uint32_t n_refl = __popcll(m_refl);
        DIAG_PHASE(5);  // leaf delivery
        const WaveAlloc wa = wave_alloc(sc, level, n_refl, __popcll(m_refr), __popcll(m_node));
        DIAG_PHASE(6);  // slot allocation (the atomic)
        uint32_t node = 0;
        if (interior) {
            node = node_base + wa.node_base + __popcll(m_node & below);
            p.node_a[node] = make_float4(surface.r, surface.g, surface.b, reflectance);
            // the node's own link upwards: index, slot and whether ITS parent is a single-child node
            const uint32_t bits = (link & (kLinkIndexMask | kLinkSlotBit | kLinkSingleBit)) | (use_schlick ? kLinkSchlickBit : 0u) |
                                  ((spawn_reflect != spawn_refract) ? kLinkOwnSingleBit : 0u);
            p.node_d[node] = make_uint2(bits, __float_as_uint(weight));
        }
        // Children of a node with one child say so in their link: that node is folded where its child's
        // colour is delivered (deliver) and k_combine skips it.
        const uint32_t single = (spawn_reflect != spawn_refract) ? kLinkSingleBit : 0u;
        if (spawn_reflect) {
            const uint32_t slot = out_base + wa.refl_base + __popcll(m_refl & below);
            p.q_org[qo][slot] = make_float4(over_point.x, over_point.y, over_point.z, C ? __uint_as_float(node | single) : over_point.w);
            p.q_dir[qo][slot] = make_float4(reflectv.x, reflectv.y, reflectv.z, C ? reflective : reflectv.w);
            if (!C) p.q_link[qo][slot] = make_uint2(node | single, __float_as_uint(reflective));
        }
        if (spawn_refract) {
            const uint32_t slot = out_top - (wa.refr_base + __popcll(m_refr & below));
            p.q_org[qo][slot] =
                make_float4(under_point.x, under_point.y, under_point.z, C ? __uint_as_float(node | kLinkSlotBit | single) : under_point.w);
            p.q_dir[qo][slot] = make_float4(refract_dir.x, refract_dir.y, refract_dir.z, C ? transparency : refract_dir.w);
            if (!C) p.q_link[qo][slot] = make_uint2(node | kLinkSlotBit | single, __float_as_uint(transparency));
        }
        DIAG_PHASE(7);  // node + child-ray stores
    }
    DIAG_FLUSH();
    // one atomic per wave for the hit (= shadow ray) count
    for (int off = 32; off > 0; off >>= 1) {
        local_hits += __shfl_down(local_hits, off);
        local_patterned += __shfl_down(local_patterned, off);
        local_transparent += __shfl_down(local_transparent, off);
    }
    // Statistics go to the wave's shard line too: a chip-wide counter word would serialise
    // every wave of the launch at the memory side (measured: ~11 ns per same-address atomic).
    if (lane == 0 && local_hits) {
        atomicAdd(&sc->hits[level], local_hits);
        if (local_patterned | local_transparent)
            atomicAdd(&sc->hit_kinds, (static_cast<unsigned long long>(local_transparent) << 32) | local_patterned);
    }
    if constexpr (STATS) {
        uint32_t box = tally.bvh_box, tri = tally.bvh_tri;
        for (int off = 32; off > 0; off >>= 1) {
            box += __shfl_down(box, off);
            tri += __shfl_down(tri, off);
        }
        if (lane == 0) {
            StatLine* sl = &p.counters->stats[shard];
            if (box) atomicAdd(&sl->bvh_box, (unsigned long long)box);
            if (tri) atomicAdd(&sl->bvh_tri, (unsigned long long)tri);
            if (tally.gate) atomicAdd(&sl->gate, (unsigned long long)tally.gate);
#pragma unroll
            for (int k = 0; k < 6; ++k)
                if (tally.shape[k]) atomicAdd(&sl->shape[k], (unsigned long long)tally.shape[k]);
        }
    }
}

// Interior nodes of one level, after every deeper level has delivered into their slots.
__global__ __launch_bounds__(256) void k_combine(const Params p, const uint32_t level) {
    const bool root = level == 0;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint32_t n_waves = (gridDim.x * blockDim.x) >> 6;
    const uint32_t shard = wave % kShards;
    const uint32_t count = static_cast<uint32_t>(p.counters->shard[shard].alloc[level] >> 32);
    const uint32_t node_base = p.node_off[level] + shard * (p.shard_cap0 << (level * p.level_shift));
    for (uint32_t i = (wave / kShards) * 64u + lane; i < count; i += (n_waves / kShards) * 64u) {
        const uint32_t node = node_base + i;
        const uint2 dd = p.node_d[node];
        const uint32_t bits = dd.x;
        if (bits & kLinkOwnSingleBit) continue;  // folded where its only child's colour was delivered
        const float4 a = p.node_a[node], b = p.node_b[node], c = p.node_c[node];
        const Color out = combine(Color{a.x, a.y, a.z}, Color{b.x, b.y, b.z}, Color{c.x, c.y, c.z},
                                  (bits & kLinkSchlickBit) != 0u, a.w);
        deliver(p, root, bits, __uint_as_float(dd.y), out);
    }
}

// Fold this batch's counters into the running totals and clear them for the next batch:
// one wave, lane k owns shard k.  `reset` starts a new frame's totals.
__global__ void k_accumulate(Counters* c, Totals* t, uint32_t level0_rays, uint32_t depth, uint32_t reset) {
    const uint32_t k = threadIdx.x;  // one wave; lane k owns shards k, k + 64, ...
    unsigned long long rays[kLevels] = {}, hits[kLevels] = {}, nodes = 0, patterned = 0, transparent = 0;
    unsigned long long ev[9] = {};  // StatLine's nine counts
    for (uint32_t sh = k; sh < kShards; sh += 64u) {
        ShardCounters* sc = &c->shard[sh];
        unsigned long long* sl = reinterpret_cast<unsigned long long*>(&c->stats[sh]);
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            ev[e] += sl[e];
            sl[e] = 0ull;
        }
#pragma unroll
        for (int l = 0; l < kLevels; ++l) {
            const unsigned long long a = sc->alloc[l];
            rays[l] += (static_cast<uint32_t>(l) <= depth) ? static_cast<unsigned long long>(static_cast<uint32_t>(a)) + sc->refr[l] : 0ull;
            nodes += (static_cast<uint32_t>(l) <= depth) ? (a >> 32) : 0ull;
            hits[l] += sc->hits[l];
            sc->alloc[l] = 0ull;
            sc->hits[l] = 0u;
            sc->refr[l] = 0u;
            c->claims[sh].claim[l] = 0u;
        }
        patterned += static_cast<uint32_t>(sc->hit_kinds);
        transparent += sc->hit_kinds >> 32;
        sc->hit_kinds = 0ull;
    }
    for (int off = 32; off > 0; off >>= 1) {
#pragma unroll
        for (int l = 0; l < kLevels; ++l) {
            rays[l] += __shfl_down(rays[l], off);
            hits[l] += __shfl_down(hits[l], off);
        }
        nodes += __shfl_down(nodes, off);
        patterned += __shfl_down(patterned, off);
        transparent += __shfl_down(transparent, off);
#pragma unroll
        for (int e = 0; e < 9; ++e) ev[e] += __shfl_down(ev[e], off);
    }
    if (k == 0) {
        if (reset) *t = Totals{};
        t->bvh_box += ev[0];
        t->bvh_tri += ev[1];
        t->gate += ev[2];
        for (int e = 0; e < 6; ++e) t->shape[e] += ev[3 + e];
        t->level_rays[0] += level0_rays;
        for (uint32_t l = 0; l < depth; ++l) t->level_rays[l + 1] += rays[l];
        for (uint32_t l = 0; l <= depth; ++l) t->level_hits[l] += hits[l];
        t->nodes += nodes;
        t->hits_patterned += patterned;
        t->hits_transparent += transparent;
    }
}

// canvas.zig:45-47: round(clamp(c * 255, 0, 255)), round half away from zero.
__device__ __forceinline__ uint32_t quantise(float c) {
    float v = c * 255.0f;
    v = fmaxf(0.0f, fminf(v, 255.0f));
    return static_cast<uint32_t>(roundf(v));
}
// Four values per thread (one 16-byte load, one 4-byte store) where the buffers are aligned for it
// (`n4` groups, 0 if they are not), one value per thread for the rest.
__global__ void k_quantise(const float* rgb, uint8_t* out, size_t n, size_t n4) {
    const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    const float4* rgb4 = reinterpret_cast<const float4*>(rgb);
    uint32_t* out4 = reinterpret_cast<uint32_t*>(out);
    for (size_t i = tid; i < n4; i += stride) {
        const float4 v = rgb4[i];
        out4[i] = quantise(v.x) | (quantise(v.y) << 8) | (quantise(v.z) << 16) | (quantise(v.w) << 24);
    }
    for (size_t i = 4 * n4 + tid; i < n; i += stride) out[i] = static_cast<uint8_t>(quantise(rgb[i]));
}

}  // namespace

// ---------------------------------------------------------------------- host side

constexpr int kLanes = 2;

// One set of work buffers, sized for `cap_pixels` pixels per batch at depth kMaxDepth.
struct WorkSet {
    uint32_t cap_pixels = 0;
    uint32_t shard_cap0 = 0;  // level-0 items per shard the buffers were sized for
    float4* q_org[2] = {nullptr, nullptr};
    float4* q_dir[2] = {nullptr, nullptr};
    uint2* q_link[2] = {nullptr, nullptr};
    float4* node_a = nullptr;
    float4* node_b = nullptr;
    float4* node_c = nullptr;
    uint2* node_d = nullptr;
    Counters* d_counters = nullptr;  // two sets, used alternately by successive batches (see k_trace_shade)
    uint32_t counter_set = 0;        // the set the next batch uses
    Totals* d_totals = nullptr;
    bool counters_dirty = true;    // d_counters are not known to be zero (see run_levels)
    hipStream_t stream = nullptr;  // lane stream (lane 0 of a single-lane frame runs on the caller's)
    hipEvent_t done = nullptr;
};

// ---- work buffers that are never allocated: rtx_render_device_static --------------------------
//
// Callers that cannot call hipMalloc (HIP graph capture, tight serving loops, harnesses that forbid
// it) render through one work set carved out of this file-scope device array.  It is sized for
// batches of kStaticPixels pixels in the worst case (every hit spawns two children): frames up to
// kStaticPixels / 8 pixels wide render in 8-row bands of that size.  Each GPU the library runs on
// holds its own copy of the array, at its own address, from module load until the process ends.
// In host code `scratch` is only a stand-in for the symbol: the address kernels and memsets use
// comes from hipGetSymbolAddress, once per device (static_scratch_base).
constexpr uint32_t kStaticPixels = 8192;
constexpr uint32_t kStaticMaxWidth = kStaticPixels / 8u;
constexpr size_t align256(size_t n) { return (n + 255u) & ~(size_t)255u; }
constexpr size_t kStaticPerLevel0 = kStaticPixels;  // kShards * 64 * ceil(ceil(px / 64) / kShards) for px a multiple of 64 * kShards
static_assert(kStaticPixels % (64u * kShards) == 0u, "kStaticPerLevel0 assumes whole chunk rounds");
constexpr size_t kStaticQCap[2] = {kStaticPerLevel0 << (kMaxDepth - 1), kStaticPerLevel0 << kMaxDepth};
constexpr size_t kStaticNodeCap = kStaticPerLevel0 * ((1u << kMaxDepth) - 1u);
// layout: [counters x2][q_org0][q_org1][q_dir0][q_dir1][q_link0][q_link1][node_a][node_b][node_c][node_d]
constexpr size_t kStaticOffCounters = 0;
constexpr size_t kStaticOffQOrg0 = kStaticOffCounters + align256(2 * sizeof(Counters));
constexpr size_t kStaticOffQOrg1 = kStaticOffQOrg0 + align256(kStaticQCap[0] * sizeof(float4));
constexpr size_t kStaticOffQDir0 = kStaticOffQOrg1 + align256(kStaticQCap[1] * sizeof(float4));
constexpr size_t kStaticOffQDir1 = kStaticOffQDir0 + align256(kStaticQCap[0] * sizeof(float4));
constexpr size_t kStaticOffQLink0 = kStaticOffQDir1 + align256(kStaticQCap[1] * sizeof(float4));
constexpr size_t kStaticOffQLink1 = kStaticOffQLink0 + align256(kStaticQCap[0] * sizeof(uint2));
constexpr size_t kStaticOffNodeA = kStaticOffQLink1 + align256(kStaticQCap[1] * sizeof(uint2));
constexpr size_t kStaticOffNodeB = kStaticOffNodeA + align256(kStaticNodeCap * sizeof(float4));
constexpr size_t kStaticOffNodeC = kStaticOffNodeB + align256(kStaticNodeCap * sizeof(float4));
constexpr size_t kStaticOffNodeD = kStaticOffNodeC + align256(kStaticNodeCap * sizeof(float4));
constexpr size_t kStaticScratchBytes = kStaticOffNodeD + align256(kStaticNodeCap * sizeof(uint2));
static_assert(kStaticScratchBytes <= (32u << 20), "the static work set is documented as at most 32 MiB per GPU");

static __device__ __align__(256) unsigned char scratch[kStaticScratchBytes];

struct rtx_scene {
    int device = 0;
    hipStream_t stream = nullptr;
    rtx_shape* d_shapes = nullptr;
    HitRec* d_hitrecs = nullptr;         // one per shape (build_hitrecs)
    rtx_node* d_nodes = nullptr;         // gates with BVH links (reserved[0] = root + 1, reserved[1] = linear shapes)
    rtx_node* d_nodes_linear = nullptr;  // the same gates, brute force only (rtx_opts.no_bvh)
    rtx_bvh::Node* d_bvh_nodes = nullptr;
    rtx_bvh::Tri* d_bvh_tris = nullptr;
    uint32_t n_bvh_nodes = 0, n_bvh_tris = 0;
    struct DeviceTree {  // a tree built on the device (RTX_SCENE_BVH_DEVICE), kept for rtx_scene_refit
        rtx_lbvh::Tree* tree;
        uint32_t node_base, tri_base;
    };
    std::vector<DeviceTree> device_trees;
    std::vector<rtx_node> h_nodes;  // the gates as uploaded (BVH links included): refit re-uploads them with new boxes
    bool host_trees = false;        // some tree came from the host builder (such a scene cannot be refitted)
    rtx_material* d_materials = nullptr;
    rtx_pattern* d_patterns = nullptr;
    uint32_t n_shapes = 0, n_nodes = 0, n_materials = 0, n_patterns = 0;
    bool track_n1 = false;   // some material is transparent: n1/n2 matter
    bool can_spawn = false;  // some material is reflective or transparent: child rays exist
    bool canon = false;      // the scene keeps w == 1 / w == 0 exact (see dot<C>)
    bool two_children = false;  // some material is reflective AND transparent: a hit may spawn two rays
    bool finite_schlick = true;  // every refractive index is finite and > 0: schlick() cannot return inf / NaN
    float light_position[4];
    float light_intensity[3];

    // Work buffers come in kLanes independent sets: the batches of one frame alternate between
    // them on two streams, so the kernels of one batch fill the tails (and the empty deep
    // levels) of the other's.
    WorkSet ws[kLanes];
    WorkSet ws_static;  // pointers into this device's `scratch` (rtx_render_device_static); owns nothing
    hipEvent_t ev_fork = nullptr;
    float* d_canvas = nullptr;  // staging canvas for the host-output entry points
    size_t canvas_floats = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    int num_cus = 256;
    int resident_blocks[4][2] = {};  // k_trace_shade blocks per CU: [canonical + 2 * bvh][0 queue-fed, 1 ray-generating]
    // profile = 1: pooled events bracketing each launch, pending until read back
    std::vector<hipEvent_t> prof_events;
    std::vector<int> prof_kind;  // per pair: 0 trace_shade, 1 combine
    size_t prof_used = 0;
};

namespace {

// Device address of `scratch` on `device`, which must be the current device: looked up with
// hipGetSymbolAddress the first time the library is set up on that device (rtx_scene_create) and
// kept — every GPU holds its own copy of the array at its own address.
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[kMaxDevices] = {};

int static_scratch_base(int device, unsigned char** out) {
    if (device < 0 || device >= kMaxDevices) return fail(RTX_E_INVALID, "device %d: the static work set supports devices 0..%d", device, kMaxDevices - 1);
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device]) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(RTX_E_HIP, "hipGetSymbolAddress(scratch) returned a null address on device %d", device);
        g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    }
    *out = g_scratch_ptr[device];
    return RTX_OK;
}

// Point a work set at the static array of `device` (current).  Nothing is allocated or owned.
int bind_static_work(int device, WorkSet* w) {
    unsigned char* base = nullptr;
    const int rc = static_scratch_base(device, &base);
    if (rc) return rc;
    w->d_counters = reinterpret_cast<Counters*>(base + kStaticOffCounters);
    w->q_org[0] = reinterpret_cast<float4*>(base + kStaticOffQOrg0);
    w->q_org[1] = reinterpret_cast<float4*>(base + kStaticOffQOrg1);
    w->q_dir[0] = reinterpret_cast<float4*>(base + kStaticOffQDir0);
    w->q_dir[1] = reinterpret_cast<float4*>(base + kStaticOffQDir1);
    w->q_link[0] = reinterpret_cast<uint2*>(base + kStaticOffQLink0);
    w->q_link[1] = reinterpret_cast<uint2*>(base + kStaticOffQLink1);
    w->node_a = reinterpret_cast<float4*>(base + kStaticOffNodeA);
    w->node_b = reinterpret_cast<float4*>(base + kStaticOffNodeB);
    w->node_c = reinterpret_cast<float4*>(base + kStaticOffNodeC);
    w->node_d = reinterpret_cast<uint2*>(base + kStaticOffNodeD);
    w->cap_pixels = kStaticPixels;
    w->shard_cap0 = (uint32_t)(kStaticPerLevel0 / kShards);
    w->counter_set = 0;
    w->counters_dirty = true;
    return RTX_OK;
}

void free_work(WorkSet* w) {
    for (int i = 0; i < 2; ++i) {
        if (w->q_org[i]) (void)hipFree(w->q_org[i]);
        if (w->q_dir[i]) (void)hipFree(w->q_dir[i]);
        if (w->q_link[i]) (void)hipFree(w->q_link[i]);
        w->q_org[i] = w->q_dir[i] = nullptr;
        w->q_link[i] = nullptr;
    }
    if (w->node_a) (void)hipFree(w->node_a);
    if (w->node_b) (void)hipFree(w->node_b);
    if (w->node_c) (void)hipFree(w->node_c);
    if (w->node_d) (void)hipFree(w->node_d);
    w->node_a = w->node_b = w->node_c = nullptr;
    w->node_d = nullptr;
    w->cap_pixels = 0;
}

// Worst-case sizing, so no overflow path is needed: a hit spawns a reflected child if its
// material is reflective and a refracted one if it is transparent (src/world.zig:131,151).
// If some material is both, level d holds at most 2^d * pixels rays and as many interior
// nodes; otherwise a level never outgrows the pixels.  Queue 0 carries the even levels and
// queue 1 the odd ones.  `need_link`: the general-w ray format keeps link/weight in a third array.
// Items per shard at level 0 for a batch of `pixels`, and the ray-tree nodes / queue slots a batch of
// that size can need in the worst case (64-bit: the caller checks them against the 30-bit link index).
struct WorkSizes {
    uint32_t shard_cap0;
    uint64_t per_level0, q_cap[2], node_cap;
};
WorkSizes work_sizes(uint64_t pixels, bool two_children) {
    WorkSizes z;
    // Level-0 chunks of 64 items are dealt round-robin to the shards, so a shard sees at most
    // 64 * ceil(chunks / kShards) items at level 0 and (1 << shift * d) times that at level d.
    const uint32_t shift = two_children ? 1u : 0u;
    const uint64_t chunks = (pixels + 63u) / 64u;
    z.shard_cap0 = (uint32_t)(64u * ((chunks + kShards - 1u) / kShards));
    z.per_level0 = (uint64_t)kShards * z.shard_cap0;
    z.q_cap[0] = z.per_level0 << (shift * (kMaxDepth - 1));  // levels 0, 2, 4
    z.q_cap[1] = z.per_level0 << (shift * kMaxDepth);        // levels 1, 3, 5
    z.node_cap = 0;
    for (int l = 0; l < kMaxDepth; ++l) z.node_cap += z.per_level0 << (shift * l);
    return z;
}

int ensure_work(const rtx_scene* s, WorkSet* w, uint32_t pixels, bool need_link) {
    if (pixels <= w->cap_pixels && (!need_link || w->q_link[0])) return RTX_OK;
    if (pixels < w->cap_pixels) pixels = w->cap_pixels;
    need_link = need_link || w->q_link[0];
    free_work(w);
    const WorkSizes z = work_sizes(pixels, s->two_children);
    if (z.node_cap > kLinkIndexMask || z.q_cap[1] > 0xFFFFFFFFull)
        return fail(RTX_E_INVALID, "a batch of %u pixels needs %llu ray-tree nodes: beyond the 30-bit node index", pixels,
                    (unsigned long long)z.node_cap);
    const uint32_t shard_cap0 = z.shard_cap0;
    const size_t q_cap[2] = {(size_t)z.q_cap[0], (size_t)z.q_cap[1]};
    const size_t node_cap = (size_t)z.node_cap;
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipMalloc(&w->q_org[i], q_cap[i] * sizeof(float4)));
        HIP_TRY(hipMalloc(&w->q_dir[i], q_cap[i] * sizeof(float4)));
        if (need_link) HIP_TRY(hipMalloc(&w->q_link[i], q_cap[i] * sizeof(uint2)));
    }
    HIP_TRY(hipMalloc(&w->node_a, node_cap * sizeof(float4)));
    HIP_TRY(hipMalloc(&w->node_b, node_cap * sizeof(float4)));
    HIP_TRY(hipMalloc(&w->node_c, node_cap * sizeof(float4)));
    HIP_TRY(hipMalloc(&w->node_d, node_cap * sizeof(uint2)));
    w->cap_pixels = pixels;
    w->shard_cap0 = shard_cap0;
    return RTX_OK;
}

// Point a batch's parameters at a work set.
void bind_work(Params& p, const WorkSet& w) {
    for (int i = 0; i < 2; ++i) {
        p.q_org[i] = w.q_org[i];
        p.q_dir[i] = w.q_dir[i];
        p.q_link[i] = w.q_link[i];
    }
    p.node_a = w.node_a;
    p.node_b = w.node_b;
    p.node_c = w.node_c;
    p.node_d = w.node_d;
    p.counters = w.d_counters + w.counter_set;
    p.counters_next = w.d_counters + (w.counter_set ^ 1u);
}

int ensure_canvas(rtx_scene* s, size_t floats) {
    if (floats <= s->canvas_floats) return RTX_OK;
    if (s->d_canvas) (void)hipFree(s->d_canvas);
    s->d_canvas = nullptr;
    s->canvas_floats = 0;
    HIP_TRY(hipMalloc(&s->d_canvas, floats * sizeof(float)));
    s->canvas_floats = floats;
    return RTX_OK;
}

struct RowPlan {
    uint32_t rows_total;   // rows of the frame that are rendered at all (rows quirk applied)
    uint32_t rows_local;   // rows this call renders
    uint32_t strip_rows, strip_count, strip_index;
};

RowPlan plan_rows(const rtx_camera* cam, const rtx_opts* o) {
    RowPlan r;
    const uint32_t h = cam->vertical_size;
    r.rows_total = (o && o->rows_quirk) ? 8u * (h / 8u) : h;  // src/camera.zig:87-88
    r.strip_count = (o && o->strip_count > 1) ? o->strip_count : 1;
    r.strip_rows = (o && o->strip_rows) ? o->strip_rows : 8;
    r.strip_index = (o && r.strip_count > 1) ? o->strip_index : 0;
    if (r.strip_count == 1) {
        r.rows_local = r.rows_total;
    } else {
        // strips s = strip_index, strip_index + strip_count, ... of strip_rows rows each
        uint32_t rows = 0;
        for (uint32_t s = r.strip_index;; s += r.strip_count) {
            const uint64_t y0 = (uint64_t)s * r.strip_rows;
            if (y0 >= r.rows_total) break;
            const uint64_t y1 = y0 + r.strip_rows;
            rows += (uint32_t)((y1 > r.rows_total ? r.rows_total : y1) - y0);
        }
        r.rows_local = rows;
    }
    return r;
}

// The k_trace_shade instantiation a frame uses, picked from run-time flags.
using TraceKernel = void (*)(const Params, const uint32_t);
// Run-time flags -> template flags, one at a time, in k_trace_shade's parameter order.
template <int TAB, bool... Flags>
TraceKernel pick_kernel() {
    return k_trace_shade<TAB, Flags...>;
}
template <int TAB, bool... Flags, class... Rest>
TraceKernel pick_kernel(bool f, Rest... rest) {
    return f ? pick_kernel<TAB, Flags..., true>(rest...) : pick_kernel<TAB, Flags..., false>(rest...);
}
// Book-mode kernels exist in the plainest form only (tables in global memory, four-lane arithmetic, no
// event tallies): the mode reports against the reference's stale images, it is not a fast path.
template <bool GEN>
TraceKernel book_kernel(bool bvh) {
    return bvh ? k_trace_shade<kTabGlobal, GEN, true, false, true, false, true> : k_trace_shade<kTabGlobal, GEN, true, false, false, false, true>;
}
template <bool GEN>
TraceKernel trace_kernel(bool track, uint32_t tab, bool canon, bool bvh, bool stats = false) {
    if (tab == (uint32_t)kTabAll) return pick_kernel<kTabAll, GEN>(track, canon, bvh, stats);
    if (tab == (uint32_t)kTabHit) return pick_kernel<kTabHit, GEN>(track, canon, bvh, stats);
    return pick_kernel<kTabGlobal, GEN>(track, canon, bvh, stats);
}

template <bool GEN>
void launch_trace(const rtx_scene* s, const Params& p, uint32_t level, uint32_t blocks, hipStream_t st, bool stats) {
    // At the depth limit (remaining == 0) nothing refracts (src/world.zig:151) and the Schlick
    // reflectance only weighs two black children (src/world.zig:71-78), so n1 is dead there —
    // provided the reflectance is finite (0 * R must be 0): s->finite_schlick.  The last level
    // holds the most rays and runs the kernel without predecessor tracking.
    const bool track = s->track_n1 && !(level == p.depth && s->finite_schlick);
    const TraceKernel k = (p.book && track) ? book_kernel<GEN>(p.has_bvh != 0) : trace_kernel<GEN>(track, p.tab_mode, p.canon != 0, p.has_bvh != 0, stats);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), p.lds_bytes, st, p, level);
}

// Resident 256-thread blocks per CU of a variant (registers and LDS decide).
template <bool GEN>
int resident_blocks(const rtx_scene* s, uint32_t tab_mode, uint32_t lds_bytes, bool canon, bool bvh) {
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<GEN>(s->track_n1, tab_mode, canon, bvh), 256, lds_bytes);
    return (e == hipSuccess && n > 0) ? n : 4;
}

int validate_desc(const rtx_scene_desc* d) {
    if (!d) return fail(RTX_E_INVALID, "scene descriptor is NULL");
    if (d->n_nodes == 0 || !d->nodes) return fail(RTX_E_INVALID, "scene needs at least the world node");
    if (d->n_shapes && !d->shapes) return fail(RTX_E_INVALID, "shapes is NULL");
    if (d->n_materials == 0 || !d->materials) return fail(RTX_E_INVALID, "scene needs at least one material");
    if (d->n_materials > kHitMaterialMask + 1u) return fail(RTX_E_INVALID, "too many materials (%u, at most %u)", d->n_materials, kHitMaterialMask + 1u);
    if (d->n_patterns && !d->patterns) return fail(RTX_E_INVALID, "patterns is NULL");
    if (d->n_shapes > kLinkIndexMask) return fail(RTX_E_INVALID, "too many shapes");
    for (uint32_t i = 0; i < d->n_shapes; ++i) {
        const rtx_shape& s = d->shapes[i];
        if (s.type > RTX_TRIANGLE) return fail(RTX_E_INVALID, "shape %u: bad type %u", i, s.type);
        if (s.material >= d->n_materials) return fail(RTX_E_INVALID, "shape %u: material %u out of range", i, s.material);
    }
    for (uint32_t i = 0; i < d->n_materials; ++i)
        if (d->materials[i].pattern >= (int32_t)d->n_patterns)
            return fail(RTX_E_INVALID, "material %u: pattern %d out of range", i, d->materials[i].pattern);
    for (uint32_t i = 0; i < d->n_patterns; ++i)
        if (d->patterns[i].color_map > RT_PATTERN_CHECKER) return fail(RTX_E_INVALID, "pattern %u: bad color_map", i);
    uint32_t next_shape = 0;
    for (uint32_t i = 0; i < d->n_nodes; ++i) {
        const rtx_node& n = d->nodes[i];
        // Nodes own consecutive shape ranges in insertion order, and skip must move forward:
        // the traversal loop relies on both.
        if (n.first_shape != next_shape) return fail(RTX_E_INVALID, "node %u: shapes are not in insertion order", i);
        if ((uint64_t)n.first_shape + n.n_shapes > d->n_shapes) return fail(RTX_E_INVALID, "node %u: shape range out of bounds", i);
        if (n.skip <= i || n.skip > d->n_nodes) return fail(RTX_E_INVALID, "node %u: skip %u does not move forward", i, n.skip);
        next_shape += n.n_shapes;
    }
    if (next_shape != d->n_shapes) return fail(RTX_E_INVALID, "nodes cover %u of %u shapes", next_shape, d->n_shapes);
    return RTX_OK;
}

// One triangle BVH: where it is walked and which shapes it holds (for rtx_bvh_selftest).
struct BvhInfo {
    uint32_t root, walk_node;
    std::vector<uint32_t> shapes;
};

// Library-side derived data: identity flags on the shapes and the triangle BVHs.
//
// Nodes that share their gates form a class: a node with a box of its own (or the world) plus every
// descendant reached without passing another box — a lane is active in all of them or in none
// (OBJ groups: the bounded top group and its unbounded named sub-groups, src/obj_file.zig:45-62).
// The trailing triangle runs of a class's nodes go into ONE tree, walked at the class's last node,
// after every linearly tested shape of the class; the walk resolves ties on the shape index, so
// mixing runs of different nodes keeps the insertion-order tie-break (Q13).
// Links in rtx_node.reserved: [0] = root + 1 of the tree walked at this node (0: none), [1] = leading
// shapes of the node tested linearly (all of them if none of its triangles is in a tree).
// `linear` receives the gates without BVH links (rtx_opts.no_bvh).
// `on_device` (may be NULL): classes whose tree the caller builds on the device (rtx_lbvh) instead —
// listed with their hittable triangles, linked later (BvhInfo::root is meaningless until then).
void prepare_shapes(std::vector<rtx_shape>& shapes);
void build_bvhs(std::vector<rtx_shape>& shapes, std::vector<rtx_node>& nodes, std::vector<rtx_node>& linear,
                std::vector<rtx_bvh::Node>& bvh_nodes, std::vector<rtx_bvh::Tri>& bvh_tris, std::vector<BvhInfo>* info = nullptr,
                std::vector<BvhInfo>* on_device = nullptr);

// Identity / diagonal flags and the finished world normals of planes and triangles.
void prepare_shapes(std::vector<rtx_shape>& shapes) {
    static const float kIdentity[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (rtx_shape& sh : shapes) {
        bool ident = true;
        for (int k = 0; k < 16; ++k) ident = ident && sh.inverse[k] == kIdentity[k];
        sh.flags &= 0xFFu;  // bits above are the library's
        if (ident) sh.flags |= kShapeIdentity;
        static const int kOffDiagonal[6] = {1, 2, 4, 6, 8, 9};
        bool diag = !ident && sh.inverse[12] == 0.0f && sh.inverse[13] == 0.0f && sh.inverse[14] == 0.0f && sh.inverse[15] == 1.0f;
        for (int k : kOffDiagonal) diag = diag && sh.inverse[k] == 0.0f;
        if (diag) sh.flags |= kShapeDiagonal;
        if (sh.type == RTX_PLANE || sh.type == RTX_TRIANGLE) {
            float wn[4];
            world_normal_of(sh, wn);
            std::memcpy(sh.normal, wn, sizeof wn);
            sh.flags |= kShapeWorldNormal;
        }
    }
}

void build_bvhs(std::vector<rtx_shape>& shapes, std::vector<rtx_node>& nodes, std::vector<rtx_node>& linear,
                std::vector<rtx_bvh::Node>& bvh_nodes, std::vector<rtx_bvh::Tri>& bvh_tris, std::vector<BvhInfo>* info,
                std::vector<BvhInfo>* on_device) {
    prepare_shapes(shapes);
    for (rtx_node& n : nodes) {
        n.reserved[0] = 0;
        n.reserved[1] = n.n_shapes;
    }
    linear = nodes;
    // class of node i = its nearest ancestor-or-self with a box (the world node 0 otherwise)
    const uint32_t nn = (uint32_t)nodes.size();
    std::vector<uint32_t> cls(nn, 0u), open_nodes;
    for (uint32_t i = 0; i < nn; ++i) {
        while (!open_nodes.empty() && i >= nodes[open_nodes.back()].skip) open_nodes.pop_back();
        cls[i] = (i == 0 || nodes[i].has_bounds || open_nodes.empty()) ? i : cls[open_nodes.back()];
        open_nodes.push_back(i);
    }
    for (uint32_t c = 0; c < nn; ++c) {
        if (cls[c] != c) continue;
        // rt_flatten emits a node's shapes in type order, so its triangles are the tail of its range.
        // The ABI does not require that order: only the maximal trailing run of triangles goes into
        // the tree, whatever precedes it (triangles included) is tested linearly, in insertion order.
        std::vector<uint32_t> tris, members, n_lin;
        for (uint32_t i = c; i < nn; ++i) {
            if (cls[i] != c) continue;
            uint32_t lin = nodes[i].n_shapes;
            while (lin > 0 && shapes[nodes[i].first_shape + lin - 1].type == RTX_TRIANGLE) --lin;
            members.push_back(i);
            n_lin.push_back(lin);
            for (uint32_t k = lin; k < nodes[i].n_shapes; ++k) tris.push_back(nodes[i].first_shape + k);
        }
        if (tris.size() < rtx_bvh::kMinTris) continue;
        if (on_device) {
            std::vector<uint32_t> hittable;  // e0 == 0 can never be hit (see rtx_bvh::build_list)
            for (uint32_t t : tris) {
                const rtx_shape& sh = shapes[t];
                if (!(sh.e0[0] == 0.0f && sh.e0[1] == 0.0f && sh.e0[2] == 0.0f && sh.e0[3] == 0.0f)) hittable.push_back(t);
            }
            if (hittable.size() >= rtx_bvh::kMinTris) {
                for (size_t k = 0; k < members.size(); ++k) nodes[members[k]].reserved[1] = n_lin[k];
                on_device->push_back(BvhInfo{0u, members.back(), hittable});
                continue;
            }
        }
        const uint32_t root = rtx_bvh::build_list(shapes.data(), tris, bvh_nodes, bvh_tris);
        if (root == rtx_bvh::kNoLeaf) continue;
        for (size_t k = 0; k < members.size(); ++k) nodes[members[k]].reserved[1] = n_lin[k];
        nodes[members.back()].reserved[0] = root + 1;
        if (info) info->push_back(BvhInfo{root, members.back(), tris});
    }
}

// The 16-byte hit records of the (library-flagged) shapes: see HitRec.
std::vector<HitRec> build_hitrecs(const std::vector<rtx_shape>& shapes) {
    std::vector<HitRec> out(shapes.size());
    for (size_t i = 0; i < shapes.size(); ++i) {
        const rtx_shape& sh = shapes[i];
        HitRec& h = out[i];
        h.bits = (sh.material & kHitMaterialMask) | (sh.type << kHitTypeShift) | ((sh.flags & kShapeIdentity) ? kHitIdentity : 0u) |
                 ((sh.flags & kShapeDiagonal) ? kHitDiagonal : 0u) | ((sh.flags & kShapeWorldNormal) ? kHitWorldNormal : 0u);
        for (int k = 0; k < 3; ++k) h.n[k] = (sh.flags & kShapeWorldNormal) ? sh.normal[k] : 0.0f;
    }
    return out;
}

int check_camera(const rtx_camera* cam) {
    if (!cam) return fail(RTX_E_INVALID, "camera is NULL");
    if (cam->horizontal_size == 0 || cam->vertical_size == 0) return fail(RTX_E_INVALID, "empty canvas");
    if ((uint64_t)cam->horizontal_size * cam->vertical_size > kLinkIndexMask) return fail(RTX_E_INVALID, "canvas too large");
    return RTX_OK;
}

// Which tables go to LDS (k_trace_shade's TAB) and how many bytes that takes.
void lds_plan(uint32_t n_shapes, uint32_t n_materials, uint32_t n_patterns, uint32_t* tab_mode, uint32_t* lds_bytes) {
    const size_t small = sizeof(rtx_material) * n_materials + sizeof(rtx_pattern) * n_patterns + sizeof(HitRec) * n_shapes;
    const size_t all = small + sizeof(rtx_shape) * n_shapes;
    *tab_mode = all <= kLdsSceneBytes ? (uint32_t)kTabAll : (small <= kLdsSceneBytes ? (uint32_t)kTabHit : (uint32_t)kTabGlobal);
    *lds_bytes = *tab_mode == (uint32_t)kTabAll ? (uint32_t)all : (*tab_mode == (uint32_t)kTabHit ? (uint32_t)small : 0u);
}

void fill_scene_params(const rtx_scene* s, Params& p, uint32_t depth) {
    std::memset(&p, 0, sizeof p);
    p.shapes = s->d_shapes;
    p.nodes = s->d_nodes;
    p.bvh_nodes = s->d_bvh_nodes;
    p.bvh_tris = s->d_bvh_tris;
    p.has_bvh = s->n_bvh_nodes ? 1u : 0u;
    p.finite_schlick = s->finite_schlick ? 1u : 0u;
    p.materials = s->d_materials;
    p.patterns = s->d_patterns;
    p.n_nodes = s->n_nodes;
    p.n_shapes = s->n_shapes;
    p.n_materials = s->n_materials;
    p.n_patterns = s->n_patterns;
    p.hitrecs = s->d_hitrecs;
    lds_plan(s->n_shapes, s->n_materials, s->n_patterns, &p.tab_mode, &p.lds_bytes);
    p.track_n1 = s->track_n1 ? 1u : 0u;
    p.depth = depth;
    std::memcpy(p.light_position, s->light_position, sizeof p.light_position);
    std::memcpy(p.light_intensity, s->light_intensity, sizeof p.light_intensity);
    p.over_eps = kEpsilon * 100.0f;  // f32 product, one ulp below 0.001f (src/world.zig:55)
}

// Sum the pending profile events into `out` and mark them consumed (stream must be idle).
int collect_profile(rtx_scene* s, rtx_stats* out) {
    for (size_t i = 0; i + 1 < s->prof_used && i / 2 < s->prof_kind.size(); i += 2) {
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, s->prof_events[i], s->prof_events[i + 1]));
        if (s->prof_kind[i / 2] == 0) {
            out->ms_trace_shade += ms;
            out->launches_trace_shade++;
        } else {
            out->ms_combine += ms;
            out->launches_combine++;
        }
    }
    s->prof_used = 0;
    s->prof_kind.clear();
    return RTX_OK;
}

// Run the levels for one batch whose level-0 rays are either generated (GEN) or already in
// queue 0 (rtx_color_at).
template <bool GEN>
int run_levels(rtx_scene* s, WorkSet& w, Params& p, uint32_t pixels, uint32_t level0_items, hipStream_t st, bool prof,
               bool first, bool want_totals) {
    // Each batch finds its counter set cleared by the batch before it (k_trace_shade, level 0);
    // both sets are cleared here only when that cannot be relied on (fresh buffers, or an
    // earlier batch that failed half way).
    if (w.counters_dirty) HIP_TRY(hipMemsetAsync(w.d_counters, 0, 2 * sizeof(Counters), st));
    w.counters_dirty = true;
    bind_work(p, w);
    p.shard_cap0 = w.shard_cap0;
    p.level_shift = s->two_children ? 1u : 0u;
    uint32_t off = 0;
    for (int l = 0; l < kLevels; ++l) {
        p.node_off[l] = off;
        if (l < kMaxDepth) off += (kShards * w.shard_cap0) << (l * p.level_shift);
    }
    // Every launch has a multiple of kShards waves (16 blocks of 4 waves), see k_trace_shade.
    constexpr uint32_t kBlockQuantum = kShards / 4u;
    // Persistent grid: one resident round of blocks (what fits the CUs at once), each wave
    // grid-striding over its share.  RTX_BLOCKS_PER_CU overrides the occupancy query (experiments).
    static const int bpc_override = [] {
        const char* e = getenv("RTX_BLOCKS_PER_CU");
        return e ? atoi(e) : 0;
    }();
    const int cz = (p.canon ? 1 : 0) + (p.has_bvh ? 2 : 0);
    if (s->resident_blocks[cz][0] == 0) {
        s->resident_blocks[cz][0] = resident_blocks<false>(s, p.tab_mode, p.lds_bytes, p.canon != 0, p.has_bvh != 0);
        s->resident_blocks[cz][1] = resident_blocks<true>(s, p.tab_mode, p.lds_bytes, p.canon != 0, p.has_bvh != 0);
    }
    auto grid_for = [&](uint64_t items, bool gen) {
        const uint32_t bpc = bpc_override > 0 ? (uint32_t)bpc_override : (uint32_t)s->resident_blocks[cz][gen ? 1 : 0];
        uint32_t max_blocks = ((uint32_t)s->num_cus * bpc) / kBlockQuantum * kBlockQuantum;
        if (max_blocks < kBlockQuantum) max_blocks = kBlockQuantum;
        uint64_t blocks = (items + 255) / 256;
        blocks = (blocks + kBlockQuantum - 1) / kBlockQuantum * kBlockQuantum;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks < kBlockQuantum) blocks = kBlockQuantum;
        return (uint32_t)blocks;
    };
    // Events come in (begin, end) pairs, prof_kind holds one entry per pair.  If an event cannot be
    // created or recorded, profiling stops for the rest of this call and the half-recorded pair is
    // dropped, so collect_profile never pairs events of different launches.
    auto mark = [&](int kind) {
        if (!prof) return;
        const bool begin = kind >= 0;
        if (s->prof_used == s->prof_events.size()) {
            hipEvent_t e;
            if (hipEventCreateWithFlags(&e, hipEventReleaseToDevice) != hipSuccess) {  // device-scope release: no system-wide flush around the timed launch
                prof = false;
                if (!begin) {  // the pair's begin event is already in: take it out again
                    --s->prof_used;
                    s->prof_kind.pop_back();
                }
                return;
            }
            s->prof_events.push_back(e);
        }
        if (hipEventRecord(s->prof_events[s->prof_used], st) != hipSuccess) {
            prof = false;
            if (!begin) {
                --s->prof_used;
                s->prof_kind.pop_back();
            }
            return;
        }
        ++s->prof_used;
        if (begin) s->prof_kind.push_back(kind);
    };
    // A scene without a reflective or transparent material never spawns a child ray
    // (src/world.zig:131,151): its deeper levels would be empty launches.
    const uint32_t last_level = s->can_spawn ? p.depth : 0u;
    for (uint32_t level = 0; level <= last_level; ++level) {
        // Level 0 knows its size; deeper levels read theirs on the device and grid-stride.
        const uint32_t blocks = grid_for(level == 0 ? level0_items : ((uint64_t)pixels << (level * p.level_shift)), GEN && level == 0);
        mark(0);
        // frames that read statistics back run the counting variants (event tallies for rtx_stats)
        if (level == 0) launch_trace<GEN>(s, p, level, blocks, st, want_totals);
        else launch_trace<false>(s, p, level, blocks, st, want_totals);
        mark(-1);
    }
    for (int level = (int)last_level - 1; level >= 0; --level) {
        const uint32_t blocks = grid_for((uint64_t)pixels << ((uint32_t)level * p.level_shift), false);
        mark(1);
        hipLaunchKernelGGL(k_combine, dim3(blocks), dim3(256), 0, st, p, (uint32_t)level);
        mark(-1);
    }
    // Frame totals (ray counts by level, hit kinds) only when the caller reads them back.
    if (want_totals)
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(64), 0, st, p.counters, w.d_totals, level0_items ? pixels : 0u, last_level,
                           first ? 1u : 0u);
    HIP_TRY(hipGetLastError());
    w.counter_set ^= 1u;
    w.counters_dirty = false;
    return RTX_OK;
}

// `fixed`: render through this one work set and allocate nothing (rtx_render_device_static); NULL: the
// scene's own work sets, grown on demand.
int render_impl(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, float* d_rgb, uint8_t* d_rgb8,
                hipStream_t st, rtx_stats* stats, WorkSet* fixed = nullptr) {
    int rc = check_camera(cam);
    if (rc) return rc;
    uint32_t depth = (opts && opts->depth) ? opts->depth : kMaxDepth;
    if (depth > kMaxDepth) return fail(RTX_E_INVALID, "depth %u exceeds ray_depth_limit %d", depth, kMaxDepth);
    if (opts && opts->strip_count > 1 && opts->strip_index >= opts->strip_count)
        return fail(RTX_E_INVALID, "strip_index %u >= strip_count %u", opts->strip_index, opts->strip_count);
    HIP_TRY(hipSetDevice(s->device));

    const RowPlan plan = plan_rows(cam, opts);
    const uint32_t W = cam->horizontal_size, H = cam->vertical_size;
    const bool compact = opts && opts->compact_rows;
    const bool profile = opts && opts->profile && !fixed;  // per-launch events come from a growing pool: not on the static path
    const size_t out_rows = compact ? plan.rows_local : H;

    // Batch: whole 8-row bands, at most batch_pixels pixels.
    // Default: a whole 3840x2160 frame per batch.  Deep levels of small batches hold too few rays
    // to fill the chip (one wave iteration of a BVH scene takes tens of microseconds), and 288 GB
    // of HBM pays for the worst-case queues (about 25 GB at this size when levels can double).
    uint32_t batch_pixels = (opts && opts->batch_pixels) ? opts->batch_pixels : (1u << 23);
    if (batch_pixels > (1u << 25)) batch_pixels = 1u << 25;  // keeps node indices (31 * pixels) inside 30 bits
    uint32_t rows_per_batch = (batch_pixels / W) & ~7u;
    if (rows_per_batch < 8) rows_per_batch = 8;
    if (rows_per_batch > ((plan.rows_local + 7u) & ~7u)) rows_per_batch = (plan.rows_local + 7u) & ~7u;
    if (rows_per_batch == 0) rows_per_batch = 8;
    // Two lanes: a frame that would be one batch is cut in two halves that run on two streams
    // with their own work buffers, so each half's kernels fill the other's tails and empty
    // levels.  Frames with per-launch events (profile) stay on one lane: their kernels must run
    // alone to be timed.  rtx_opts.lanes picks; 0 = auto: only multi-level frames of >= 4 Mpx,
    // because the fork/join between streams costs tens of microseconds per frame (a 1080p frame
    // loop loses 14 % to it, a 4K mesh frame gains 13 %).  RTX_LANES overrides auto (A/B).
    static const int lanes_env = [] {
        const char* e = getenv("RTX_LANES");
        return e ? atoi(e) : 0;
    }();
    int lanes = (int)((opts && opts->lanes) ? opts->lanes : (lanes_env > 0 ? (uint32_t)lanes_env : 0u));
    if (lanes > kLanes) return fail(RTX_E_INVALID, "lanes %d exceeds %d", lanes, kLanes);
    if (lanes == 0) lanes = (s->can_spawn && (uint64_t)plan.rows_local * W >= (1u << 22)) ? kLanes : 1;
    if (profile || plan.rows_local < 64u || fixed) lanes = 1;
    if (lanes > 1) {
        const uint32_t half = ((plan.rows_local + (uint32_t)lanes - 1u) / (uint32_t)lanes + 7u) & ~7u;
        if (half < rows_per_batch) rows_per_batch = half;
    }
    const uint32_t tiles_x = (W + 7u) / 8u;
    if (fixed) {  // batches must fit the work set as it is
        const uint32_t fit = (fixed->cap_pixels / (tiles_x * 8u)) & ~7u;
        if (fit < 8u)
            return fail(RTX_E_INVALID, "rtx_render_device_static: width %u exceeds the static work set's maximum of %u pixels", W, kStaticMaxWidth);
        if (rows_per_batch > fit) rows_per_batch = fit;
    }
    // The clamp on batch_pixels alone does not bound the node index: tile padding (W = 1 pads to 8) and
    // the rounding of shard capacities add to it.  Shrink the batch until the worst-case node count
    // fits the 30-bit link index (a batch is at least one 8-row band).
    while (rows_per_batch > 8u && work_sizes((uint64_t)tiles_x * 8u * rows_per_batch, s->two_children).node_cap > kLinkIndexMask)
        rows_per_batch = ((rows_per_batch / 2u) + 7u) & ~7u;
    if (work_sizes((uint64_t)tiles_x * 8u * rows_per_batch, s->two_children).node_cap > kLinkIndexMask)
        return fail(RTX_E_INVALID, "canvas too wide: one 8-row band of %u pixels exceeds the ray-tree node index", W);
    const uint32_t cap = tiles_x * 8u * rows_per_batch;  // pixels incl. tile padding
    Params p;
    fill_scene_params(s, p, depth);
    {
        const float* ci = cam->inverse_transform;
        const bool cam_canon = ci[12] == 0.0f && ci[13] == 0.0f && ci[14] == 0.0f && ci[15] == 1.0f;
        static const bool off = getenv("RTX_NO_CANON") != nullptr;  // A/B switch for experiments
        p.canon = (s->canon && cam_canon && !off && !(opts && opts->full_w)) ? 1u : 0u;
    }
    if (opts && opts->containers > RTX_CONTAINERS_BOOK) return fail(RTX_E_INVALID, "containers %u: RTX_CONTAINERS_V1 or RTX_CONTAINERS_BOOK", opts->containers);
    if (opts && opts->containers == RTX_CONTAINERS_BOOK && s->track_n1) {
        // every level of the frame runs the plain variants (see book_kernel): one ray format, no LDS tables
        p.book = 1u;
        p.canon = 0u;
        p.tab_mode = (uint32_t)kTabGlobal;
        p.lds_bytes = 0u;
    }
    for (int l = 0; l < lanes && !fixed; ++l) {
        rc = ensure_work(s, &s->ws[l], cap, p.canon == 0);
        if (rc) return rc;
    }
    if (fixed) fixed->counters_dirty = true;  // the array is shared by every scene on this GPU and keeps the last call's contents
    if (opts && opts->no_bvh) {
        p.nodes = s->d_nodes_linear;
        p.has_bvh = 0;
    }
    std::memcpy(p.cam_inv, cam->inverse_transform, sizeof p.cam_inv);
    p.half_width = cam->half_width;
    p.half_height = cam->half_height;
    p.pixel_size = cam->pixel_size;
    p.width = W;
    p.tiles_x = tiles_x;
    p.strip_rows = plan.strip_rows;
    p.strip_count = plan.strip_count;
    p.strip_index = plan.strip_index;
    p.compact_rows = compact ? 1u : 0u;
    p.canvas = d_rgb;

    if (!fixed) HIP_TRY(hipEventRecord(s->ev_begin, st));
    // Canvas starts black (canvas.zig:23-25); rows never rendered stay black.  Every rendered
    // pixel is written exactly once, so only a canvas with unrendered rows needs the fill.
    if (!compact && plan.rows_local < H) HIP_TRY(hipMemsetAsync(d_rgb, 0, out_rows * W * 3 * sizeof(float), st));
    // Lane streams start after everything queued on the caller's stream so far (one lane: the
    // caller's stream itself carries the work).
    hipStream_t lane_stream[kLanes];
    for (int l = 0; l < lanes; ++l) lane_stream[l] = lanes == 1 ? st : s->ws[l].stream;
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(s->ev_fork, st));
        for (int l = 0; l < lanes; ++l) HIP_TRY(hipStreamWaitEvent(lane_stream[l], s->ev_fork, 0));
    }

    uint32_t batch = 0;
    for (uint32_t row0 = 0; row0 < plan.rows_local; row0 += rows_per_batch, ++batch) {
        const uint32_t n_rows = (plan.rows_local - row0 < rows_per_batch) ? plan.rows_local - row0 : rows_per_batch;
        p.row0 = row0;
        p.n_rows = n_rows;
        const uint32_t items = tiles_x * ((n_rows + 7u) / 8u) * 64u;
        const int l = (int)(batch % (uint32_t)lanes);
        rc = run_levels<true>(s, fixed ? *fixed : s->ws[l], p, cap, items, lane_stream[l], profile, batch < (uint32_t)lanes, stats != nullptr);
        if (rc) return rc;
    }
    const int lanes_used = batch < (uint32_t)lanes ? (int)batch : lanes;
    if (lanes > 1) {
        for (int l = 0; l < lanes; ++l) {
            HIP_TRY(hipEventRecord(s->ws[l].done, lane_stream[l]));
            HIP_TRY(hipStreamWaitEvent(st, s->ws[l].done, 0));
        }
    }
    if (d_rgb8) {
        const size_t n = out_rows * W * 3;
        const bool aligned = (reinterpret_cast<uintptr_t>(d_rgb) % 16u == 0u) && (reinterpret_cast<uintptr_t>(d_rgb8) % 4u == 0u);
        hipLaunchKernelGGL(k_quantise, dim3(2048), dim3(256), 0, st, d_rgb, d_rgb8, n, aligned ? n / 4 : (size_t)0);
    }
    if (!fixed) HIP_TRY(hipEventRecord(s->ev_end, st));
    HIP_TRY(hipGetLastError());

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        Totals t{}, part[kLanes];
        for (int l = 0; l < lanes_used; ++l) HIP_TRY(hipMemcpyAsync(&part[l], s->ws[l].d_totals, sizeof(Totals), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int l = 0; l < lanes_used; ++l) {
            for (int k = 0; k < kLevels; ++k) {
                t.level_rays[k] += part[l].level_rays[k];
                t.level_hits[k] += part[l].level_hits[k];
            }
            t.nodes += part[l].nodes;
            t.hits_patterned += part[l].hits_patterned;
            t.hits_transparent += part[l].hits_transparent;
            t.bvh_box += part[l].bvh_box;
            t.bvh_tri += part[l].bvh_tri;
            t.gate += part[l].gate;
            for (int k = 0; k < 6; ++k) t.shape[k] += part[l].shape[k];
        }
        // k_accumulate counts tile padding at level 0; the rays that exist are the pixels.
        stats->rays_primary = (uint64_t)plan.rows_local * W;
        stats->level_rays[0] = stats->rays_primary;
        for (uint32_t l = 1; l <= depth; ++l) {
            stats->level_rays[l] = t.level_rays[l];
            stats->rays_secondary += t.level_rays[l];
        }
        for (uint32_t l = 0; l <= depth; ++l) {
            stats->level_hits[l] = t.level_hits[l];
            stats->rays_shadow += t.level_hits[l];
        }
        stats->nodes = t.nodes;
        stats->hits_patterned = t.hits_patterned;
        stats->hits_transparent = t.hits_transparent;
        stats->bvh_box_tests = t.bvh_box;
        stats->bvh_triangle_tests = t.bvh_tri;
        stats->gate_tests = t.gate;
        for (int k = 0; k < 6; ++k) stats->shape_tests[k] = t.shape[k];
        stats->rows_rendered = plan.rows_local;
        stats->kernel_flags = (p.canon ? RTX_KERNEL_CANONICAL_W : 0u) | (p.has_bvh ? RTX_KERNEL_BVH : 0u) |
                              (p.tab_mode == (uint32_t)kTabAll ? RTX_KERNEL_LDS_TABLES : 0u) | (p.tab_mode == (uint32_t)kTabHit ? RTX_KERNEL_LDS_HIT_RECORDS : 0u) | (s->track_n1 ? RTX_KERNEL_TRACK_N1 : 0u);
        HIP_TRY(hipEventElapsedTime(&stats->ms_total, s->ev_begin, s->ev_end));
        if (profile) {
            rc = collect_profile(s, stats);
            if (rc) return rc;
        }
    }
    return RTX_OK;
}

}  // namespace

#ifdef RTX_DIAG
extern "C" __attribute__((visibility("default"))) int rtx_diag_read(unsigned long long* out16, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_diag), sizeof(unsigned long long) * 16));
    if (reset) {
        unsigned long long z[16] = {0};
        HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_diag), z, sizeof z));
    }
    return 0;
}
#endif

extern "C" {

const char* rtx_last_error(void) { return g_err; }

// For the library's other translation units (rtx_multi.hip): set this thread's message.  Not in rtx.h.
__attribute__((visibility("default"))) void rtx_set_error(int code, const char* msg) { (void)fail(code, "%s", msg ? msg : ""); }

int rtx_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(RTX_E_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}

int rtx_scene_create(const rtx_scene_desc* desc, int device, rtx_scene** out) {
    if (!out) return fail(RTX_E_INVALID, "out is NULL");
    *out = nullptr;
    int rc = validate_desc(desc);
    if (rc) return rc;
    int n = rtx_device_count();
    if (n <= 0) return fail(RTX_E_NO_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(RTX_E_INVALID, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));

    rtx_scene* s = new (std::nothrow) rtx_scene();
    if (!s) return fail(RTX_E_NOMEM, "out of host memory");
    s->device = device;
    auto bail = [&](int code) {
        rtx_scene_destroy(s);
        return code;
    };
#define HIP_TRY_S(expr)                                                                                  \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return bail(fail(RTX_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)));      \
    } while (0)
    hipDeviceProp_t prop;
    HIP_TRY_S(hipGetDeviceProperties(&prop, device));
    s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    HIP_TRY_S(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    HIP_TRY_S(hipEventCreate(&s->ev_begin));
    HIP_TRY_S(hipEventCreate(&s->ev_end));

    s->n_shapes = desc->n_shapes;
    s->n_nodes = desc->n_nodes;
    s->n_materials = desc->n_materials;
    s->n_patterns = desc->n_patterns;
    std::memcpy(s->light_position, desc->light_position, sizeof s->light_position);
    std::memcpy(s->light_intensity, desc->light_intensity, sizeof s->light_intensity);
    for (uint32_t i = 0; i < desc->n_shapes; ++i)
    {
        const rtx_material& m = desc->materials[desc->shapes[i].material];
        if (m.transparency != 0.0f) s->track_n1 = true;
        if (m.transparency != 0.0f || m.reflective != 0.0f) s->can_spawn = true;
        if (m.transparency != 0.0f && m.reflective != 0.0f) s->two_children = true;
        if (!(m.refractive_index > 0.0f && m.refractive_index < INFINITY)) s->finite_schlick = false;
    }

    auto upload = [&](void** dst, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(dst, bytes ? bytes : 16);
        if (e != hipSuccess) return e;
        if (bytes) e = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
        return e;
    };
    // Library-side derived data: identity flags and one BVH per long triangle run.
    std::vector<rtx_shape> shapes(desc->shapes, desc->shapes + desc->n_shapes);
    std::vector<rtx_node> nodes(desc->nodes, desc->nodes + desc->n_nodes);
    // Canonical w lanes (see dot<C>): affine inverses, untranslated cubes, light at w == 1.
    s->canon = desc->light_position[3] == 1.0f;
    for (const rtx_shape& sh : shapes) {
        const float* m = sh.inverse;
        if (!(m[12] == 0.0f && m[13] == 0.0f && m[14] == 0.0f && m[15] == 1.0f)) s->canon = false;
        if (sh.type == RTX_CUBE && !(m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f)) s->canon = false;
        // A triangle whose e0 is exactly zero (the OBJ fan's degenerate first triangle, src/obj_file.zig:156-164)
        // has det == 0 (or NaN) for every ray and is never hit (src/triangle.zig:36-37): its normal —
        // normalize(0) = NaN — is never read and does not decide anything here.
        const bool never_hit = sh.type == RTX_TRIANGLE && sh.e0[0] == 0.0f && sh.e0[1] == 0.0f && sh.e0[2] == 0.0f && sh.e0[3] == 0.0f;
        if (sh.type == RTX_TRIANGLE && !never_hit && !(sh.normal[3] == 0.0f)) s->canon = false;
    }
    std::vector<rtx_node> linear;
    std::vector<rtx_bvh::Node> bvh_nodes;
    std::vector<rtx_bvh::Tri> bvh_tris;
    // Trees come from the host builder (binned SAH) unless the descriptor asks for the device builder
    // (LBVH, rtx_lbvh.hip): RTX_SCENE_BVH_DEVICE, or RTX_BVH_BUILD=device in the environment (experiments).
    static const bool env_device = [] {
        const char* e = getenv("RTX_BVH_BUILD");
        return e && std::strcmp(e, "device") == 0;
    }();
    const bool device_build = (desc->flags & RTX_SCENE_BVH_DEVICE) != 0u || env_device;
    // RTX_TIMING=1: wall time of the phases of this call on stderr (tools/bvh_build_time.py)
    static const bool timing = getenv("RTX_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        if (!timing) return;
        const auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "rtx_scene_create: %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t_last).count());
        t_last = now;
    };
    std::vector<BvhInfo> deferred;
    build_bvhs(shapes, nodes, linear, bvh_nodes, bvh_tris, nullptr, device_build ? &deferred : nullptr);
    s->host_trees = !bvh_nodes.empty();
    lap(device_build ? "shape records + classes" : "shape records + host trees");
    HIP_TRY_S(upload((void**)&s->d_shapes, shapes.data(), sizeof(rtx_shape) * shapes.size()));
    lap("shape upload");
    {
        const std::vector<HitRec> hitrecs = build_hitrecs(shapes);
        HIP_TRY_S(upload((void**)&s->d_hitrecs, hitrecs.data(), sizeof(HitRec) * hitrecs.size()));
    }
    {
        // Device-built trees: plan each (sizes), then one node array and one triangle array hold the host
        // trees followed by the device trees, which are emitted in place.
        size_t total_nodes = bvh_nodes.size(), total_tris = bvh_tris.size();
        std::string err;
        for (const BvhInfo& c : deferred) {
            rtx_lbvh::Tree* tree = nullptr;
            uint32_t per_layout = 0;
            const int trc = rtx_lbvh::plan(s->d_shapes, c.shapes.data(), (uint32_t)c.shapes.size(), s->stream, &tree, &per_layout, err);
            if (trc) return bail(fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH build: %s", err.c_str()));
            lap("device tree: plan");
            s->device_trees.push_back(rtx_scene::DeviceTree{tree, (uint32_t)total_nodes, (uint32_t)total_tris});
            nodes[c.walk_node].reserved[0] = (uint32_t)total_nodes + 1u;
            total_nodes += (size_t)per_layout * rtx_bvh::kLayouts;
            total_tris += c.shapes.size();
        }
        if (total_nodes > 0xFFFFFFF0ull) return bail(fail(RTX_E_INVALID, "BVH node array too large"));
        HIP_TRY_S(hipMalloc(&s->d_bvh_nodes, (total_nodes ? total_nodes : 1) * sizeof(rtx_bvh::Node)));
        HIP_TRY_S(hipMalloc(&s->d_bvh_tris, (total_tris ? total_tris : 1) * sizeof(rtx_bvh::Tri)));
        if (!bvh_nodes.empty()) HIP_TRY_S(hipMemcpy(s->d_bvh_nodes, bvh_nodes.data(), sizeof(rtx_bvh::Node) * bvh_nodes.size(), hipMemcpyHostToDevice));
        if (!bvh_tris.empty()) HIP_TRY_S(hipMemcpy(s->d_bvh_tris, bvh_tris.data(), sizeof(rtx_bvh::Tri) * bvh_tris.size(), hipMemcpyHostToDevice));
        for (const rtx_scene::DeviceTree& t : s->device_trees) {
            const int trc = rtx_lbvh::emit(t.tree, s->d_shapes, s->d_bvh_nodes, t.node_base, s->d_bvh_tris, t.tri_base, s->stream, err);
            if (trc) return bail(fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH build: %s", err.c_str()));
            lap("device tree: emit + check");
        }
        s->n_bvh_nodes = (uint32_t)total_nodes;
        s->n_bvh_tris = (uint32_t)total_tris;
    }
    lap("tree upload");
    s->h_nodes = nodes;
    HIP_TRY_S(upload((void**)&s->d_nodes, nodes.data(), sizeof(rtx_node) * nodes.size()));
    HIP_TRY_S(upload((void**)&s->d_nodes_linear, linear.data(), sizeof(rtx_node) * linear.size()));
    HIP_TRY_S(upload((void**)&s->d_materials, desc->materials, sizeof(rtx_material) * desc->n_materials));
    HIP_TRY_S(upload((void**)&s->d_patterns, desc->patterns, sizeof(rtx_pattern) * desc->n_patterns));
    HIP_TRY_S(hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming));
    // Library set-up on this device (it is current): the device address of the static work set.
    rc = bind_static_work(device, &s->ws_static);
    if (rc) return bail(rc);
    {  // occupancy of every kernel variant this scene can dispatch, so that no render call has to ask
        uint32_t tab = 0, lds = 0;
        lds_plan(s->n_shapes, s->n_materials, s->n_patterns, &tab, &lds);
        for (int cz = 0; cz < 4; ++cz) {
            s->resident_blocks[cz][0] = resident_blocks<false>(s, tab, lds, (cz & 1) != 0, (cz & 2) != 0);
            s->resident_blocks[cz][1] = resident_blocks<true>(s, tab, lds, (cz & 1) != 0, (cz & 2) != 0);
        }
    }
    for (WorkSet& w : s->ws) {
        HIP_TRY_S(hipMalloc(&w.d_counters, 2 * sizeof(Counters)));
        HIP_TRY_S(hipMalloc(&w.d_totals, sizeof(Totals)));
        HIP_TRY_S(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
        HIP_TRY_S(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    }
#undef HIP_TRY_S
    *out = s;
    return RTX_OK;
}

void rtx_scene_destroy(rtx_scene* s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (WorkSet& w : s->ws) {
        if (w.stream) (void)hipStreamSynchronize(w.stream);
        free_work(&w);
        if (w.d_counters) (void)hipFree(w.d_counters);
        if (w.d_totals) (void)hipFree(w.d_totals);
        if (w.done) (void)hipEventDestroy(w.done);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->d_canvas) (void)hipFree(s->d_canvas);
    if (s->d_shapes) (void)hipFree(s->d_shapes);
    if (s->d_hitrecs) (void)hipFree(s->d_hitrecs);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_nodes_linear) (void)hipFree(s->d_nodes_linear);
    for (rtx_scene::DeviceTree& t : s->device_trees) rtx_lbvh::destroy(t.tree);
    s->device_trees.clear();
    if (s->d_bvh_nodes) (void)hipFree(s->d_bvh_nodes);
    if (s->d_bvh_tris) (void)hipFree(s->d_bvh_tris);
    if (s->d_materials) (void)hipFree(s->d_materials);
    if (s->d_patterns) (void)hipFree(s->d_patterns);
    for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
    if (s->ev_begin) (void)hipEventDestroy(s->ev_begin);
    if (s->ev_end) (void)hipEventDestroy(s->ev_end);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int rtx_scene_refit(rtx_scene* s, const rtx_scene_desc* desc) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (s->host_trees) return fail(RTX_E_INVALID, "rtx_scene_refit needs a scene whose trees were built on the device (RTX_SCENE_BVH_DEVICE)");
    if (desc->n_shapes != s->n_shapes || desc->n_nodes != s->n_nodes || desc->n_materials != s->n_materials || desc->n_patterns != s->n_patterns)
        return fail(RTX_E_INVALID, "rtx_scene_refit: the descriptor has another shape / node / material / pattern count than the scene");
    for (uint32_t i = 0; i < desc->n_nodes; ++i) {
        const rtx_node &a = desc->nodes[i], &b = s->h_nodes[i];
        if (a.first_shape != b.first_shape || a.n_shapes != b.n_shapes || a.skip != b.skip || a.has_bounds != b.has_bounds)
            return fail(RTX_E_INVALID, "rtx_scene_refit: node %u changed its structure", i);
    }
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipStreamSynchronize(s->stream));
    std::vector<rtx_shape> shapes(desc->shapes, desc->shapes + desc->n_shapes);
    prepare_shapes(shapes);
    {   // what a tree holds may move, not change kind: a triangle stays a (hittable) triangle
        std::vector<rtx_shape> old(s->n_shapes);
        HIP_TRY(hipMemcpy(old.data(), s->d_shapes, sizeof(rtx_shape) * old.size(), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < s->n_shapes; ++i) {
            const bool was_flat = old[i].e0[0] == 0.0f && old[i].e0[1] == 0.0f && old[i].e0[2] == 0.0f && old[i].e0[3] == 0.0f;
            const bool is_flat = shapes[i].e0[0] == 0.0f && shapes[i].e0[1] == 0.0f && shapes[i].e0[2] == 0.0f && shapes[i].e0[3] == 0.0f;
            if (old[i].type != shapes[i].type || (shapes[i].type == RTX_TRIANGLE && was_flat != is_flat))
                return fail(RTX_E_INVALID, "rtx_scene_refit: shape %u changed its type (or became / stopped being degenerate)", i);
        }
    }
    // the scene-wide switches follow the new records
    s->canon = desc->light_position[3] == 1.0f;
    for (const rtx_shape& sh : shapes) {
        const float* m = sh.inverse;
        if (!(m[12] == 0.0f && m[13] == 0.0f && m[14] == 0.0f && m[15] == 1.0f)) s->canon = false;
        if (sh.type == RTX_CUBE && !(m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f)) s->canon = false;
    }
    std::memcpy(s->light_position, desc->light_position, sizeof s->light_position);
    std::memcpy(s->light_intensity, desc->light_intensity, sizeof s->light_intensity);
    HIP_TRY(hipMemcpy(s->d_shapes, shapes.data(), sizeof(rtx_shape) * shapes.size(), hipMemcpyHostToDevice));
    const std::vector<HitRec> hitrecs = build_hitrecs(shapes);
    HIP_TRY(hipMemcpy(s->d_hitrecs, hitrecs.data(), sizeof(HitRec) * hitrecs.size(), hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < desc->n_nodes; ++i) {
        std::memcpy(s->h_nodes[i].bounds_min, desc->nodes[i].bounds_min, sizeof s->h_nodes[i].bounds_min);
        std::memcpy(s->h_nodes[i].bounds_max, desc->nodes[i].bounds_max, sizeof s->h_nodes[i].bounds_max);
    }
    HIP_TRY(hipMemcpy(s->d_nodes, s->h_nodes.data(), sizeof(rtx_node) * s->h_nodes.size(), hipMemcpyHostToDevice));
    {
        std::vector<rtx_node> linear = s->h_nodes;
        for (rtx_node& n : linear) {
            n.reserved[0] = 0;
            n.reserved[1] = n.n_shapes;
        }
        HIP_TRY(hipMemcpy(s->d_nodes_linear, linear.data(), sizeof(rtx_node) * linear.size(), hipMemcpyHostToDevice));
    }
    std::string err;
    for (const rtx_scene::DeviceTree& t : s->device_trees) {
        const int trc = rtx_lbvh::refit(t.tree, s->d_shapes, s->d_bvh_nodes, t.node_base, s->d_bvh_tris, t.tri_base, s->stream, err);
        if (trc) return fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH refit: %s", err.c_str());
    }
    return RTX_OK;
}

int rtx_profile_collect(rtx_scene* s, rtx_stats* out) {
    if (!s || !out) return fail(RTX_E_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(s->device));
    // Events may sit on a caller's stream: wait for the whole device.
    HIP_TRY(hipDeviceSynchronize());
    std::memset(out, 0, sizeof *out);
    return collect_profile(s, out);
}

int rtx_rows_for(const rtx_camera* cam, const rtx_opts* opts, uint32_t* n_rows) {
    int rc = check_camera(cam);
    if (rc) return rc;
    if (!n_rows) return fail(RTX_E_INVALID, "n_rows is NULL");
    if (opts && opts->strip_count > 1 && opts->strip_index >= opts->strip_count)
        return fail(RTX_E_INVALID, "strip_index %u >= strip_count %u", opts->strip_index, opts->strip_count);
    *n_rows = plan_rows(cam, opts).rows_local;
    return RTX_OK;
}

int rtx_render_device(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, void* d_rgb_out, void* d_rgb8_out,
                      void* stream, rtx_stats* stats) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    if (!d_rgb_out) return fail(RTX_E_INVALID, "d_rgb_out is NULL");
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    return render_impl(s, cam, opts, (float*)d_rgb_out, (uint8_t*)d_rgb8_out, st, stats);
}

int rtx_render_device_static(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, void* d_rgb_out, void* d_rgb8_out,
                             void* stream) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    if (!d_rgb_out) return fail(RTX_E_INVALID, "d_rgb_out is NULL");
    if (!s->ws_static.d_counters) return fail(RTX_E_HIP, "the static work set was not set up for device %d", s->device);
    hipStream_t st = stream ? (hipStream_t)stream : s->stream;
    return render_impl(s, cam, opts, (float*)d_rgb_out, (uint8_t*)d_rgb8_out, st, nullptr, &s->ws_static);
}

uint32_t rtx_static_max_width(void) { return kStaticMaxWidth; }

int rtx_render(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, float* rgb_out, rtx_stats* stats) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    if (!rgb_out) return fail(RTX_E_INVALID, "rgb_out is NULL");
    int rc = check_camera(cam);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    const RowPlan plan = plan_rows(cam, opts);
    const size_t rows = (opts && opts->compact_rows) ? plan.rows_local : cam->vertical_size;
    const size_t floats = rows * cam->horizontal_size * 3;
    rc = ensure_canvas(s, floats ? floats : 1);
    if (rc) return rc;
    rc = render_impl(s, cam, opts, s->d_canvas, nullptr, s->stream, stats);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_out, s->d_canvas, floats * sizeof(float), hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return RTX_OK;
}

thread_local uint64_t g_selftest_nodes = 0, g_selftest_tris = 0;  // work of the last rtx_bvh_selftest's closest-hit walks

// Closest-hit visitor that also counts the work of the walk (host statistics only).
struct CountingVisitor : ClosestVisitor<true> {
    uint64_t nodes = 0, tris = 0;
    void visited_node() { ++nodes; }
    void tested_triangle() { ++tris; }
};

// Host-only check of the triangle BVHs a scene would get (no device needed): structure of every
// tree, and for n rays the closest hit and the n1 predecessor found by the device's own
// traversal code (compiled for the host) against brute force over the same triangles.
int rtx_bvh_selftest(const rtx_scene_desc* desc, const float* origins, const float* directions, uint32_t n,
                     uint32_t* n_bvh_nodes, uint32_t* n_bvh_tris, uint32_t* mismatches) {
    int rc = validate_desc(desc);
    if (rc) return rc;
    if (n && (!origins || !directions)) return fail(RTX_E_INVALID, "NULL ray buffer");
    std::vector<rtx_shape> shapes(desc->shapes, desc->shapes + desc->n_shapes);
    std::vector<rtx_node> nodes(desc->nodes, desc->nodes + desc->n_nodes), linear;
    std::vector<rtx_bvh::Node> bn;
    std::vector<rtx_bvh::Tri> bt;
    std::vector<BvhInfo> trees;
    build_bvhs(shapes, nodes, linear, bn, bt, &trees);
    g_selftest_nodes = g_selftest_tris = 0;
    if (n_bvh_nodes) *n_bvh_nodes = (uint32_t)bn.size();
    if (n_bvh_tris) *n_bvh_tris = (uint32_t)bt.size();
    uint32_t bad = 0;
    for (const BvhInfo& tree : trees) {
        const uint32_t root = tree.root;
        if (nodes[tree.walk_node].reserved[0] != root + 1u) return fail(RTX_E_INVALID, "bvh %u: not linked at node %u", root, tree.walk_node);
        const std::vector<uint32_t>& members = tree.shapes;  // ascending shape indices
        auto slot_of = [&](uint32_t shape) -> long {
            const auto it = std::lower_bound(members.begin(), members.end(), shape);
            return (it != members.end() && *it == shape) ? (long)(it - members.begin()) : -1L;
        };
        const uint32_t stride = bn[root].skip - root;
        if (bn[root].skip <= root || (size_t)root + (size_t)rtx_bvh::kLayouts * stride > bn.size())
            return fail(RTX_E_INVALID, "bvh %u: bad extent", root);
        // structure of every layout: skips move forward and stay inside the parent, every hittable
        // triangle of the run sits in exactly one leaf
        for (uint32_t q = 0; q < rtx_bvh::kLayouts; ++q) {
            const uint32_t lo = root + q * stride, end = lo + stride;
            if (bn[lo].skip != end) return fail(RTX_E_INVALID, "bvh %u layout %u: root skip %u != %u", root, q, bn[lo].skip, end);
            std::vector<uint32_t> seen(members.size(), 0u);
            for (uint32_t k = lo; k < end; ++k) {
                const rtx_bvh::Node& nd = bn[k];
                if (nd.skip <= k || nd.skip > end) return fail(RTX_E_INVALID, "bvh node %u: skip %u leaves [%u, %u)", k, nd.skip, lo, end);
                if (nd.leaf == rtx_bvh::kNoLeaf) {
                    if (nd.skip == k + 1u) return fail(RTX_E_INVALID, "bvh node %u: interior node without children", k);
                    continue;
                }
                if (nd.skip != k + 1u) return fail(RTX_E_INVALID, "bvh node %u: a leaf with a subtree", k);
                const uint32_t t0 = nd.leaf >> 3, cnt = nd.leaf & 7u;
                if ((size_t)t0 + cnt > bt.size()) return fail(RTX_E_INVALID, "bvh node %u: leaf out of range", k);
                for (uint32_t t = t0; t < t0 + cnt; ++t) {
                    const rtx_bvh::Tri& tr = bt[t];
                    const long slot = slot_of(tr.shape);
                    if (slot < 0) return fail(RTX_E_INVALID, "bvh leaf %u: foreign triangle %u", k, tr.shape);
                    seen[(size_t)slot]++;
                    if (std::memcmp(tr.p0, shapes[tr.shape].p0, 12) || std::memcmp(tr.e0, shapes[tr.shape].e0, 12) ||
                        std::memcmp(tr.e1, shapes[tr.shape].e1, 12) || tr.flags != shapes[tr.shape].flags)
                        return fail(RTX_E_INVALID, "bvh leaf %u: triangle %u differs from its shape", k, tr.shape);
                }
            }
            for (size_t m = 0; m < members.size(); ++m) {
                const rtx_shape& sh = shapes[members[m]];
                if (sh.type != RTX_TRIANGLE) return fail(RTX_E_INVALID, "bvh %u: shape %u is not a triangle", root, members[m]);
                const bool degenerate = sh.e0[0] == 0.0f && sh.e0[1] == 0.0f && sh.e0[2] == 0.0f && sh.e0[3] == 0.0f;
                if (seen[m] != (degenerate ? 0u : 1u))
                    return fail(RTX_E_INVALID, "layout %u: triangle %u sits in %u leaves", q, members[m], seen[m]);
            }
        }
        // rays: the traversal (closest hit, then the predecessor walk) against brute force
        Params p;
        std::memset(&p, 0, sizeof p);
        p.shapes = shapes.data();
        p.bvh_nodes = bn.data();
        p.bvh_tris = bt.data();
        for (uint32_t r = 0; r < n; ++r) {
            const V4 o = v4(origins[4 * r], origins[4 * r + 1], origins[4 * r + 2], origins[4 * r + 3]);
            const V4 d = v4(directions[4 * r], directions[4 * r + 1], directions[4 * r + 2], directions[4 * r + 3]);
            CountingVisitor cv;
            cv.init();
            Tally<false> no_tally;
            traverse_bvh<false>(cv, no_tally, p, root, o, d);
            g_selftest_nodes += cv.nodes;
            g_selftest_tris += cv.tris;
            PredVisitor pv{-INFINITY, kNone};
            traverse_bvh<false>(pv, no_tally, p, root, o, d);
            ClosestVisitor<true> ref;
            ref.init();
            PredVisitor pref{-INFINITY, kNone};
            for (uint32_t t : members) {
                const rtx_shape& sh = shapes[t];
                V4 ro = o, rd = d;
                if (!(sh.flags & kShapeIdentity)) {
                    ro = mul_point<false>(sh.inverse, o);
                    rd = mul_dir<false>(sh.inverse, d);
                }
                float tt;
                if (triangle_root(ro, rd, sh.p0, sh.e0, sh.e1, tt)) {
                    ref.hit<false>(tt, t);
                    if (!(sh.flags & RTX_SHAPE_N1_INVISIBLE)) pref.pred<false>(tt, t);
                }
            }
            const bool same_hit = cv.c.idx == ref.c.idx && (cv.c.idx == kNone || bits_of(cv.c.t) == bits_of(ref.c.t));
            const bool same_pred = pv.pred_idx == pref.pred_idx && (pv.pred_idx == kNone || bits_of(pv.pred_t) == bits_of(pref.pred_t));
            if (!same_hit || !same_pred) ++bad;
        }
    }
    if (mismatches) *mismatches = bad;
    return RTX_OK;
}

// Box tests and triangle tests the closest-hit walks of the last rtx_bvh_selftest on this thread made.
void rtx_bvh_selftest_work(uint64_t* box_tests, uint64_t* triangle_tests) {
    if (box_tests) *box_tests = g_selftest_nodes;
    if (triangle_tests) *triangle_tests = g_selftest_tris;
}

int rtx_color_at(rtx_scene* s, const float* origins, const float* directions, uint32_t n, uint32_t depth, float* rgb_out) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    if (n == 0) return RTX_OK;
    if (!origins || !directions || !rgb_out) return fail(RTX_E_INVALID, "NULL ray or output buffer");
    if (depth == 0) depth = kMaxDepth;
    if (depth > kMaxDepth) return fail(RTX_E_INVALID, "depth %u exceeds ray_depth_limit %d", depth, kMaxDepth);
    if (n > (1u << 22)) return fail(RTX_E_INVALID, "at most %u rays per call", 1u << 22);
    HIP_TRY(hipSetDevice(s->device));
    // Canonical kernels need canonical rays too: origins with w == 1, directions with w == 0.
    bool canon = s->canon;
    for (uint32_t i = 0; i < n && canon; ++i) canon = origins[4 * i + 3] == 1.0f && directions[4 * i + 3] == 0.0f;
    WorkSet& w = s->ws[0];
    int rc = ensure_work(s, &w, n, !canon);
    if (rc) return rc;
    rc = ensure_canvas(s, (size_t)n * 3);
    if (rc) return rc;
    hipStream_t st = s->stream;
    std::vector<float> org, dir;  // staging must outlive the copies (synchronised below)
    std::vector<uint2> links;
    if (canon) {  // level-0 records in the 32-byte format: link in origin.w, weight in direction.w
        org.assign(origins, origins + (size_t)n * 4);
        dir.assign(directions, directions + (size_t)n * 4);
        for (uint32_t i = 0; i < n; ++i) {
            std::memcpy(&org[4 * (size_t)i + 3], &i, sizeof i);
            dir[4 * (size_t)i + 3] = 1.0f;  // weight, unused at the root
        }
        HIP_TRY(hipMemcpyAsync(w.q_org[0], org.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(w.q_dir[0], dir.data(), (size_t)n * 16, hipMemcpyHostToDevice, st));
    } else {
        links.resize(n);
        for (uint32_t i = 0; i < n; ++i) links[i] = make_uint2(i, 0x3F800000u /* weight 1.0f, unused at the root */);
        HIP_TRY(hipMemcpyAsync(w.q_org[0], origins, (size_t)n * 16, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(w.q_dir[0], directions, (size_t)n * 16, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(w.q_link[0], links.data(), (size_t)n * 8, hipMemcpyHostToDevice, st));
    }
    Params p;
    fill_scene_params(s, p, depth);
    p.canon = canon ? 1u : 0u;
    p.canvas = s->d_canvas;
    p.root_count = n;
    rc = run_levels<false>(s, w, p, n, n, st, false, true, false);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_out, s->d_canvas, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RTX_OK;
}

}  // extern "C"
