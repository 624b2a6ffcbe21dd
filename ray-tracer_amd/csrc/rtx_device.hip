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

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <type_traits>
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
constexpr uint32_t kMaxCull = 16;  // Params.cull: camera-ray tile culling entries per frame

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
    // per level d: low 32 bits = reflected rays queued for level d + 1, high 32 bits = interior nodes at d with ONE child
    // (allocated from the bottom of the shard's node region; folded at delivery, never combined)
    unsigned long long alloc[kLevels];
    // per level d: low 32 bits = refracted rays queued for level d + 1 (their own region, from the top), high 32 bits =
    // interior nodes at d with TWO children (allocated from the top of the node region: all that k_combine walks)
    unsigned long long refr[kLevels];
    unsigned long long hit_kinds;  // low 32: hits on patterned materials, high 32: on transparent ones
    uint32_t hits[kLevels];        // hits at level d == isShadowed calls
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
    unsigned long long dark;      // hits with the light behind the surface whose isShadowed call was not traversed (every kernel)
    unsigned long long pad[16 - 10];
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

// Twin shapes behind different group boxes (rtx_shape.reserved != 0), resolved per ray by the TWINS kernels
// (rtx_trace.inc: ClosestVisitor, resolve_twins).  Every such shape owns one bit of a 32-bit per-ray mask
// ("slot"); the members of a class hold consecutive slots in insertion order.
constexpr uint32_t kMaxTwinShapes = 32;
struct TwinTable {
    uint32_t n_classes;
    uint32_t pad[3];
    uint32_t member_shape[kMaxTwinShapes];  // slot -> shape index
    uint32_t cls_first[kMaxTwinShapes];     // class -> its first slot
    uint32_t cls_n[kMaxTwinShapes];         // class -> number of members
};

struct Totals {
    unsigned long long level_rays[kLevels];
    unsigned long long level_hits[kLevels];
    unsigned long long nodes;
    unsigned long long hits_patterned;
    unsigned long long hits_transparent;
    unsigned long long bvh_box, bvh_tri, gate, shape[6];
    unsigned long long shadow_skipped;
};

// Event tally of one wave of a STATS kernel; the other kernels carry the empty form and pay nothing.
// bvh_* are per lane (the BVH walk is per lane); gate / shape are wave totals kept by every lane alike.
template <bool ON>
struct Tally {
    uint32_t bvh_box = 0, bvh_tri = 0, gate = 0, shape[6] = {0, 0, 0, 0, 0, 0};
#ifdef RTX_EXP_TIMELINE
    uint32_t box_turns = 0, tri_turns = 0;  // not counted by the statistics kernels
#endif
#if defined(RTX_EXP_SLOTS) && defined(__HIP_DEVICE_COMPILE__)
    // Experiment (tools/build_variant.sh slots -DRTX_EXP_SLOTS): count the lane SLOTS each loop iteration
    // occupies (64 per wave iteration) instead of the lanes that test; events / slots = lane utilisation.
    __device__ __forceinline__ void slot(uint32_t& c) {
        const unsigned long long m = __ballot(1);
        if (__lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) c += 64u;
    }
    __device__ __forceinline__ void box() { slot(bvh_box); }
    __device__ __forceinline__ void tri() { slot(bvh_tri); }
    __device__ __forceinline__ void gates(bool lane_tests) { gate += __ballot(lane_tests) ? 64u : 0u; }
    __device__ __forceinline__ void shapes(uint32_t type, bool lane_tests) { shape[type] += __ballot(lane_tests) ? 64u : 0u; }
#else
    RTX_HD void box() { ++bvh_box; }
    RTX_HD void tri() { ++bvh_tri; }
    __device__ __forceinline__ void gates(bool lane_tests) { gate += __popcll(__ballot(lane_tests)); }
    __device__ __forceinline__ void shapes(uint32_t type, bool lane_tests) { shape[type] += __popcll(__ballot(lane_tests)); }
#endif
};
template <>
struct Tally<false> {
#ifdef RTX_EXP_TIMELINE
    // loop turns of the wave (not lane events): the first active lane of each turn counts it
    uint32_t box_turns = 0, tri_turns = 0;
#ifdef __HIP_DEVICE_COMPILE__
    __device__ __forceinline__ void turn(uint32_t& c) {
        const unsigned long long m = __ballot(1);
        if (__lane_id() == (uint32_t)(__ffsll((long long)m) - 1)) c += 1u;
    }
    __device__ __forceinline__ void box() { turn(box_turns); }
    __device__ __forceinline__ void tri() { turn(tri_turns); }
#else
    void box() {}
    void tri() {}
#endif
#else
    RTX_HD void box() {}
    RTX_HD void tri() {}
#endif
    __device__ __forceinline__ void gates(bool) {}
    __device__ __forceinline__ void shapes(uint32_t, bool) {}
};

// The arithmetic of the path lives in rtx_trace.inc, written over `real`, and is compiled twice.
RTX_HD float r_sqrt(float x) { return sqrtf(x); }
RTX_HD double r_sqrt(double x) { return sqrt(x); }
RTX_HD float r_abs(float x) { return fabsf(x); }
RTX_HD double r_abs(double x) { return fabs(x); }
RTX_HD float r_max(float a, float b) { return fmaxf(a, b); }
RTX_HD double r_max(double a, double b) { return fmax(a, b); }
RTX_HD float r_min(float a, float b) { return fminf(a, b); }
RTX_HD double r_min(double a, double b) { return fmin(a, b); }
RTX_HD float r_fma(float a, float b, float c) { return fmaf(a, b, c); }
RTX_HD double r_fma(double a, double b, double c) { return fma(a, b, c); }
RTX_HD bool r_signbit(float x) { return (__builtin_bit_cast(uint32_t, x) >> 31) != 0u; }
RTX_HD bool r_signbit(double x) { return (__builtin_bit_cast(uint64_t, x) >> 63) != 0ull; }
RTX_HD float r_floor(float x) { return floorf(x); }
RTX_HD double r_floor(double x) { return floor(x); }
RTX_HD float r_trunc(float x) { return truncf(x); }
RTX_HD double r_trunc(double x) { return trunc(x); }

namespace f32 {
using real = float;
using real4 = float4;
RTX_HD real4 make_real4(real x, real y, real z, real w) { return make_float4(x, y, z, w); }
#define RTX_F64 0
#include "rtx_trace.inc"
#undef RTX_F64
}  // namespace f32

namespace f64 {
using real = double;
using real4 = double4;
RTX_HD real4 make_real4(real x, real y, real z, real w) { return make_double4(x, y, z, w); }
#define RTX_F64 1
#include "rtx_trace.inc"
#undef RTX_F64
}  // namespace f64

using namespace f32;  // the host side below speaks f32 unless it says f64::

// Fold this batch's counters into the running totals and clear them for the next batch:
// one wave, lane k owns shard k.  `reset` starts a new frame's totals.
__global__ void k_accumulate(Counters* c, Totals* t, uint32_t level0_rays, uint32_t depth, uint32_t reset) {
    const uint32_t k = threadIdx.x;  // one wave; lane k owns shards k, k + 64, ...
    unsigned long long rays[kLevels] = {}, hits[kLevels] = {}, nodes = 0, patterned = 0, transparent = 0;
    unsigned long long ev[10] = {};  // StatLine's ten counts
    for (uint32_t sh = k; sh < kShards; sh += 64u) {
        ShardCounters* sc = &c->shard[sh];
        unsigned long long* sl = reinterpret_cast<unsigned long long*>(&c->stats[sh]);
#pragma unroll
        for (int e = 0; e < 10; ++e) {
            ev[e] += sl[e];
            sl[e] = 0ull;
        }
#pragma unroll
        for (int l = 0; l < kLevels; ++l) {
            const unsigned long long a = sc->alloc[l];
            const unsigned long long r = sc->refr[l];
            rays[l] += (static_cast<uint32_t>(l) <= depth) ? static_cast<unsigned long long>(static_cast<uint32_t>(a)) + static_cast<uint32_t>(r) : 0ull;
            nodes += (static_cast<uint32_t>(l) <= depth) ? (a >> 32) + (r >> 32) : 0ull;
            hits[l] += sc->hits[l];
            sc->alloc[l] = 0ull;
            sc->hits[l] = 0u;
            sc->refr[l] = 0ull;
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
        for (int e = 0; e < 10; ++e) ev[e] += __shfl_down(ev[e], off);
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
        t->shadow_skipped += ev[9];
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

constexpr int kLanes = 2;     // a frame can be cut into this many halves that run on streams of their own (rtx_opts.lanes)
constexpr int kWorkSets = 3;  // independent sets of work buffers per scene: as many frames can be in flight (rtx_opts.work_set)

// One set of work buffers, sized for `cap_pixels` pixels per batch at depth kMaxDepth.
struct WorkSet {
    uint32_t cap_pixels = 0;
    uint32_t real_scale = 1;  // bytes per stored real / 4: 1 for the f32 kernels, 2 for the f64 ones (queues and nodes hold reals)
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
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;  // bracket a frame rendered through this set (rtx_stats.ms_total)
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

// A factor of lighting()'s ambient term that can never make it -0 or NaN: finite, not negative, not -0.
inline bool plain_factor(float x) { return std::isfinite(x) && !std::signbit(x); }

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
        std::vector<uint32_t> shapes;  // the triangles it holds (rtx_scene_refit checks them on the host before it uploads anything)
    };
    std::vector<DeviceTree> device_trees;
    std::vector<rtx_node> h_nodes;  // the gates as uploaded (BVH links included): refit re-uploads them with new boxes
    bool host_trees = false;        // some tree came from the host builder (such a scene cannot be refitted)
    TwinTable h_twins{};            // twin shapes resolved per ray (n_classes == 0: none)
    struct CullBox {                // world-space box of a gate or of a linearly tested shape (Params.cull)
        uint32_t id;                // index | kind << 8 (kind 1: gate, index into the walkable list)
        bool quadratic;             // a sphere or cylinder: its test's discriminant cancels (fill_cull's margin)
        double lo[3], hi[3];
    };
    std::vector<CullBox> cull_boxes;
    TwinTable* d_twins = nullptr;
    bool broken = false;            // a refit failed after it had started to overwrite device records: every render call fails
    rtx_material* d_materials = nullptr;
    rtx_pattern* d_patterns = nullptr;
    uint32_t n_shapes = 0, n_nodes = 0, n_materials = 0, n_patterns = 0;
    uint32_t n_nodes_walk = 0;  // nodes in d_nodes: the gates with something to test (walkable_nodes); d_nodes_linear holds all n_nodes
    bool track_n1 = false;   // some material is transparent: n1/n2 matter
    bool can_spawn = false;  // some material is reflective or transparent: child rays exist
    bool canon = false;      // the scene keeps w == 1 / w == 0 exact (see dot<C>)
    bool two_children = false;  // some material is reflective AND transparent: a hit may spawn two rays
    bool finite_schlick = true;  // every refractive index is finite and > 0: schlick() cannot return inf / NaN
    bool plain_materials = true; // every material / pattern colour and ambient is finite and not negative (nor -0)
    bool shadow_skip = true;     // ... and the light's intensity too: no ambient term can be -0 or NaN, so a hit with the
                                 // light behind it is `ambient` shadowed or not
    float light_position[4];
    float light_intensity[3];

    // Work buffers come in kLanes independent sets: the batches of one frame alternate between
    // them on two streams, so the kernels of one batch fill the tails (and the empty deep
    // levels) of the other's.
    WorkSet ws[kWorkSets];
    WorkSet ws_static;  // pointers into this device's `scratch` (rtx_render_device_static); owns nothing
    hipEvent_t ev_fork = nullptr;
    float* d_canvas = nullptr;  // staging canvas for the host-output entry points
    size_t canvas_floats = 0;
    int num_cus = 256;
    int resident_blocks[8][2] = {};  // k_trace_shade blocks per CU: [canonical + 2 * bvh + 4 * f64][0 queue-fed, 1 ray-generating]
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
    w->real_scale = 1;
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

int ensure_work(const rtx_scene* s, WorkSet* w, uint32_t pixels, bool need_link, uint32_t real_scale = 1) {
    if (pixels <= w->cap_pixels && (!need_link || w->q_link[0]) && real_scale <= w->real_scale) return RTX_OK;
    if (pixels < w->cap_pixels) pixels = w->cap_pixels;
    if (real_scale < w->real_scale) real_scale = w->real_scale;
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
        HIP_TRY(hipMalloc(&w->q_org[i], q_cap[i] * sizeof(float4) * real_scale));
        HIP_TRY(hipMalloc(&w->q_dir[i], q_cap[i] * sizeof(float4) * real_scale));
        if (need_link) HIP_TRY(hipMalloc(&w->q_link[i], q_cap[i] * sizeof(uint2)));
    }
    HIP_TRY(hipMalloc(&w->node_a, node_cap * sizeof(float4) * real_scale));
    HIP_TRY(hipMalloc(&w->node_b, node_cap * sizeof(float4) * real_scale));
    HIP_TRY(hipMalloc(&w->node_c, node_cap * sizeof(float4) * real_scale));
    HIP_TRY(hipMalloc(&w->node_d, node_cap * sizeof(uint2)));
    w->cap_pixels = pixels;
    w->real_scale = real_scale;
    w->shard_cap0 = shard_cap0;
    return RTX_OK;
}

// Point a batch's parameters at a work set.
template <class P>
void bind_work(P& p, const WorkSet& w) {
    using Real4 = std::remove_pointer_t<decltype(p.node_a)>;  // float4 or double4: the buffers are sized by WorkSet::real_scale
    for (int i = 0; i < 2; ++i) {
        p.q_org[i] = reinterpret_cast<Real4*>(w.q_org[i]);
        p.q_dir[i] = reinterpret_cast<Real4*>(w.q_dir[i]);
        p.q_link[i] = w.q_link[i];
    }
    p.node_a = reinterpret_cast<Real4*>(w.node_a);
    p.node_b = reinterpret_cast<Real4*>(w.node_b);
    p.node_c = reinterpret_cast<Real4*>(w.node_c);
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
// The same for scenes with twin shapes behind different group boxes (TWINS): the plainest form, correctness over speed.
template <bool GEN>
TraceKernel twin_kernel(bool bvh) {
    return bvh ? k_trace_shade<kTabGlobal, GEN, true, false, true, false, false, true> : k_trace_shade<kTabGlobal, GEN, true, false, false, false, false, true>;
}
// The ray-generating kernels of frames with tile-culling entries (CULL).
template <int TAB, bool... Flags>
TraceKernel pick_cull() {
    return k_trace_shade<TAB, true, Flags..., false, false, true>;
}
template <int TAB, bool... Flags, class... Rest>
TraceKernel pick_cull(bool f, Rest... rest) {
    return f ? pick_cull<TAB, Flags..., true>(rest...) : pick_cull<TAB, Flags..., false>(rest...);
}
template <bool GEN>
TraceKernel trace_kernel(bool track, uint32_t tab, bool canon, bool bvh, bool stats = false, bool cull = false) {
    if (GEN && cull) {
        if (tab == (uint32_t)kTabAll) return pick_cull<kTabAll>(track, canon, bvh, stats);
        if (tab == (uint32_t)kTabHit) return pick_cull<kTabHit>(track, canon, bvh, stats);
        return pick_cull<kTabGlobal>(track, canon, bvh, stats);
    }
    if (tab == (uint32_t)kTabAll) return pick_kernel<kTabAll, GEN>(track, canon, bvh, stats);
    if (tab == (uint32_t)kTabHit) return pick_kernel<kTabHit, GEN>(track, canon, bvh, stats);
    return pick_kernel<kTabGlobal, GEN>(track, canon, bvh, stats);
}

template <bool GEN>
void launch_trace(const rtx_scene* s, const f32::Params& p, uint32_t level, uint32_t blocks, hipStream_t st, bool stats) {
    // At the depth limit (remaining == 0) nothing refracts (src/world.zig:151) and the Schlick
    // reflectance only weighs two black children (src/world.zig:71-78), so n1 is dead there —
    // provided the reflectance is finite (0 * R must be 0): s->finite_schlick.  The last level
    // holds the most rays and runs the kernel without predecessor tracking.
    const bool track = s->track_n1 && !(level == p.depth && s->finite_schlick);
    const TraceKernel k = (p.book && track) ? book_kernel<GEN>(p.has_bvh != 0)
                          : (p.twins && track) ? twin_kernel<GEN>(p.has_bvh != 0)
                                               : trace_kernel<GEN>(track, p.tab_mode, p.canon != 0, p.has_bvh != 0, stats, p.n_cull != 0u);
    const bool stash = stash_enabled(GEN, p.book && track, p.twins && track && !p.book);
    const uint32_t lds = p.lds_bytes + (stash ? stash_bytes_per_block(track, (int)p.tab_mode) : 0u);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), lds, st, p, level);
}

template <bool GEN>
int resident_blocks(const rtx_scene* s, uint32_t tab_mode, uint32_t lds_bytes, bool canon, bool bvh);

// The f64 reporting instantiation (rtx_opts.real = RTX_REAL_F64): the four-lane variants with every table in global memory,
// and — for frames that keep w canonical — the variants without the w lanes, with the hit records in LDS where they fit.
// No event tallies, the reference's own container walk.
using TraceKernel64 = void (*)(const f64::Params, const uint32_t);
template <int TAB, bool GEN, bool C>
TraceKernel64 trace_kernel64_t(bool track, bool bvh) {
    if (track) return bvh ? f64::k_trace_shade<TAB, GEN, true, C, true, false> : f64::k_trace_shade<TAB, GEN, true, C, false, false>;
    return bvh ? f64::k_trace_shade<TAB, GEN, false, C, true, false> : f64::k_trace_shade<TAB, GEN, false, C, false, false>;
}
template <bool GEN>
TraceKernel64 trace_kernel64(bool track, bool bvh, bool canon, uint32_t tab) {
    if (!canon) return trace_kernel64_t<kTabGlobal, GEN, false>(track, bvh);
    return tab == (uint32_t)kTabHit ? trace_kernel64_t<kTabHit, GEN, true>(track, bvh) : trace_kernel64_t<kTabGlobal, GEN, true>(track, bvh);
}
template <bool GEN>
void launch_trace(const rtx_scene* s, const f64::Params& p, uint32_t level, uint32_t blocks, hipStream_t st, bool) {
    const bool track = s->track_n1 && !(level == p.depth && s->finite_schlick);
    hipLaunchKernelGGL(trace_kernel64<GEN>(track, p.has_bvh != 0, p.canon != 0, p.tab_mode), dim3(blocks), dim3(256), p.lds_bytes, st, p, level);
}
void launch_combine(const f32::Params& p, uint32_t level, uint32_t blocks, hipStream_t st) {
    hipLaunchKernelGGL(f32::k_combine, dim3(blocks), dim3(256), 0, st, p, level);
}
void launch_combine(const f64::Params& p, uint32_t level, uint32_t blocks, hipStream_t st) {
    hipLaunchKernelGGL(f64::k_combine, dim3(blocks), dim3(256), 0, st, p, level);
}

// Resident 256-thread blocks per CU of a variant (registers and LDS decide).
template <bool GEN>
int resident_blocks(const rtx_scene* s, const f64::Params& p) {
    int n = 0;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel64<GEN>(s->track_n1, p.has_bvh != 0, p.canon != 0, p.tab_mode), 256, p.lds_bytes);
    return (e == hipSuccess && n > 0) ? n : 2;
}
template <bool GEN>
int resident_blocks(const rtx_scene* s, const f32::Params& p) {
    return resident_blocks<GEN>(s, p.tab_mode, p.lds_bytes, p.canon != 0, p.has_bvh != 0);
}
template <bool GEN>
int resident_blocks(const rtx_scene* s, uint32_t tab_mode, uint32_t lds_bytes, bool canon, bool bvh) {
    int n = 0;
    const uint32_t stash = stash_enabled(GEN, false, false) ? stash_bytes_per_block(s->track_n1, (int)tab_mode) : 0u;
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, trace_kernel<GEN>(s->track_n1, tab_mode, canon, bvh), 256, lds_bytes + stash);
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
    std::vector<uint32_t> members;  // on_device classes: the nodes whose trailing triangle runs the tree takes over
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
int prepare_shapes(std::vector<rtx_shape>& shapes, TwinTable* twins = nullptr);
int build_bvhs(std::vector<rtx_shape>& shapes, std::vector<rtx_node>& nodes, std::vector<rtx_node>& linear,
               std::vector<rtx_bvh::Node>& bvh_nodes, std::vector<rtx_bvh::Tri>& bvh_tris, std::vector<BvhInfo>* info = nullptr,
               std::vector<BvhInfo>* on_device = nullptr, TwinTable* twins = nullptr);

// Identity / diagonal flags, the finished world normals of planes and triangles, and the slots of twin shapes
// that are resolved per ray (rtx_shape.reserved; `twins`, if given, receives their table).
int prepare_shapes(std::vector<rtx_shape>& shapes, TwinTable* twins) {
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
    // Twin classes (rtx_shape.reserved = k >= 1): members in insertion order get consecutive mask bits.
    TwinTable table{};
    std::vector<uint32_t> class_ids;
    for (const rtx_shape& sh : shapes)
        if (sh.reserved && std::find(class_ids.begin(), class_ids.end(), sh.reserved) == class_ids.end()) class_ids.push_back(sh.reserved);
    uint32_t slot = 0;
    for (uint32_t id : class_ids) {
        std::vector<uint32_t> members;
        for (uint32_t i = 0; i < shapes.size(); ++i)
            if (shapes[i].reserved == id) members.push_back(i);
        if (members.size() < 2) continue;  // a class of one has nothing to cancel against: an ordinary shape
        if (slot + members.size() > kMaxTwinShapes || table.n_classes >= kMaxTwinShapes)
            return fail(RTX_E_INVALID, "more than %u twin shapes behind different group boxes (rtx_shape.reserved)", kMaxTwinShapes);
        table.cls_first[table.n_classes] = slot;
        table.cls_n[table.n_classes] = (uint32_t)members.size();
        ++table.n_classes;
        for (size_t k = 0; k < members.size(); ++k, ++slot) {
            rtx_shape& sh = shapes[members[k]];
            if (sh.type != shapes[members[0]].type || sh.material != shapes[members[0]].material ||
                std::memcmp(sh.inverse, shapes[members[0]].inverse, sizeof sh.inverse) != 0)
                return fail(RTX_E_INVALID, "twin class %u: shapes %u and %u are not identical", id, members[0], members[k]);
            table.member_shape[slot] = members[k];
            sh.flags |= kShapeTwin | (slot << kShapeTwinSlotShift);
            const bool survivor = (members.size() & 1u) && k + 1 == members.size();  // the rule when every member is reached
            if (!survivor) sh.flags |= kShapeTwinStaticInvisible;
        }
    }
    if (twins) *twins = table;
    return RTX_OK;
}

int build_bvhs(std::vector<rtx_shape>& shapes, std::vector<rtx_node>& nodes, std::vector<rtx_node>& linear,
               std::vector<rtx_bvh::Node>& bvh_nodes, std::vector<rtx_bvh::Tri>& bvh_tris, std::vector<BvhInfo>* info,
               std::vector<BvhInfo>* on_device, TwinTable* twins) {
    const int prc = prepare_shapes(shapes, twins);
    if (prc) return prc;
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
            // (a twin triangle that is resolved per ray stays on the linear path, where the walk sees it)
            while (lin > 0 && shapes[nodes[i].first_shape + lin - 1].type == RTX_TRIANGLE && !(shapes[nodes[i].first_shape + lin - 1].flags & kShapeTwin)) --lin;
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
                on_device->push_back(BvhInfo{0u, members.back(), hittable, members});
                continue;
            }
        }
        const uint32_t root = rtx_bvh::build_list(shapes.data(), tris, bvh_nodes, bvh_tris);
        if (root == rtx_bvh::kNoLeaf) continue;
        for (size_t k = 0; k < members.size(); ++k) nodes[members[k]].reserved[1] = n_lin[k];
        nodes[members.back()].reserved[0] = root + 1;
        if (info) info->push_back(BvhInfo{root, members.back(), tris, {}});
    }
    return RTX_OK;
}

// The gates as the kernels walk them: a node that has nothing to test — no box of its own, no linearly tested
// shape, no tree walked at it (an OBJ file's named sub-groups whose triangles sit in their class's tree, or an
// empty group) — costs every traversal of every wave a record fetch and a loop turn for nothing, so it is
// left out and the skip links are renumbered.  (A lane parked by a box wakes at `skip`, the first node after
// that box's subtree: the first KEPT node at or after it serves the same.)
std::vector<rtx_node> walkable_nodes(const std::vector<rtx_node>& nodes) {
    const uint32_t nn = (uint32_t)nodes.size();
    std::vector<uint32_t> new_index(nn + 1u, 0u);  // new index of the first kept node at or after old index i
    std::vector<rtx_node> out;
    out.reserve(nn);
    std::vector<bool> keep(nn, true);
    static const bool keep_all = getenv("RTX_ALL_NODES") != nullptr;  // A/B switch for experiments
    for (uint32_t i = 1; i < nn; ++i) keep[i] = keep_all || nodes[i].has_bounds || nodes[i].reserved[0] != 0u || nodes[i].reserved[1] != 0u;
    uint32_t kept = 0;
    for (uint32_t i = 0; i < nn; ++i) kept += keep[i] ? 1u : 0u;
    new_index[nn] = kept;
    for (uint32_t i = nn; i-- > 0;) new_index[i] = keep[i] ? new_index[i + 1] - 1u : new_index[i + 1];
    for (uint32_t i = 0; i < nn; ++i) {
        if (!keep[i]) continue;
        rtx_node n = nodes[i];
        n.skip = new_index[n.skip];
        out.push_back(n);
    }
    return out;
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

// Inverse of a row-major 4x4 in double (Gauss-Jordan with partial pivoting); false if singular.
bool invert4(const float* m, double* out) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = m[r * 4 + c];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (!(std::fabs(a[piv][c]) > 1e-300) || !std::isfinite(a[piv][c])) return false;
        for (int k = 0; k < 8; ++k) std::swap(a[c][k], a[piv][k]);
        const double inv = 1.0 / a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] *= inv;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = a[r][4 + c];
    return true;
}

// World-space boxes for the camera-ray tile culling (Params.cull): the gates of the walkable list that have a box, then
// the linearly tested shapes with a finite extent, indices below 32 only, at most kMaxCull.  Conservative by
// construction: a gate's box is the very box its test uses; a shape's is the box of its object-space extent's eight
// corners under the inverse of its (f32) `inverse`, padded.
std::vector<rtx_scene::CullBox> cull_boxes(const std::vector<rtx_shape>& shapes, const std::vector<rtx_node>& walk) {
    std::vector<rtx_scene::CullBox> out;
    for (uint32_t n = 0; n < walk.size() && n < 32u && out.size() < kMaxCull; ++n) {
        if (!walk[n].has_bounds) continue;
        rtx_scene::CullBox b;
        b.id = n | 0x100u;
        b.quadratic = false;
        bool finite = true;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = walk[n].bounds_min[k];
            b.hi[k] = walk[n].bounds_max[k];
            finite = finite && std::isfinite(b.lo[k]) && std::isfinite(b.hi[k]) && b.lo[k] <= b.hi[k];
        }
        if (finite) out.push_back(b);
    }
    std::vector<bool> linear(shapes.size() < 32u ? shapes.size() : 32u, false);
    for (const rtx_node& nd : walk)
        for (uint32_t k = nd.first_shape; k < nd.first_shape + nd.reserved[1] && k < linear.size(); ++k) linear[k] = true;
    for (uint32_t i = 0; i < linear.size() && out.size() < kMaxCull; ++i) {
        const rtx_shape& sh = shapes[i];
        if (!linear[i] || (sh.flags & kShapeTwin)) continue;  // a twin's test also records that the ray reached it
        double lo[3] = {-1.0, -1.0, -1.0}, hi[3] = {1.0, 1.0, 1.0};
        double pts[8][3];
        int n_pts = 8;
        switch (sh.type) {
        case RTX_SPHERE:
        case RTX_CUBE: break;
        case RTX_CYLINDER: {
            if (!std::isfinite(sh.minimum) || !std::isfinite(sh.maximum) || !(sh.minimum <= sh.maximum)) continue;
            lo[1] = sh.minimum;
            hi[1] = sh.maximum;
            break;
        }
        // no cones: a ray parallel to one of the cone's halves gets its single root WITHOUT the check against minimum /
        // maximum (src/cone.zig:35-39), so even a truncated cone can be hit anywhere on the infinite one
        case RTX_TRIANGLE: n_pts = 3; break;
        default: continue;  // planes have no extent; cones: see above
        }
        if (sh.type == RTX_TRIANGLE) {
            for (int k = 0; k < 3; ++k) {
                pts[0][k] = sh.p0[k];
                pts[1][k] = (double)sh.p0[k] + sh.e0[k];
                pts[2][k] = (double)sh.p0[k] + sh.e1[k];
            }
        } else {
            for (int c = 0; c < 8; ++c)
                for (int k = 0; k < 3; ++k) pts[c][k] = ((c >> k) & 1) ? hi[k] : lo[k];
        }
        double fwd[16];
        if (!invert4(sh.inverse, fwd)) continue;
        rtx_scene::CullBox b;
        b.id = i;
        b.quadratic = sh.type == RTX_SPHERE || sh.type == RTX_CYLINDER;
        for (int k = 0; k < 3; ++k) {
            b.lo[k] = INFINITY;
            b.hi[k] = -INFINITY;
        }
        bool finite = true;
        for (int c = 0; c < n_pts; ++c) {
            const double w = fwd[12] * pts[c][0] + fwd[13] * pts[c][1] + fwd[14] * pts[c][2] + fwd[15];
            for (int k = 0; k < 3; ++k) {
                const double x = (fwd[k * 4] * pts[c][0] + fwd[k * 4 + 1] * pts[c][1] + fwd[k * 4 + 2] * pts[c][2] + fwd[k * 4 + 3]) / w;
                finite = finite && std::isfinite(x);
                b.lo[k] = std::fmin(b.lo[k], x);
                b.hi[k] = std::fmax(b.hi[k], x);
            }
        }
        if (!finite) continue;
        for (int k = 0; k < 3; ++k) {  // the shape's own f32 arithmetic reaches a little beyond its exact extent
            const double pad = 1e-4 * (b.hi[k] - b.lo[k]) + 1e-5 * std::fmax(std::fabs(b.lo[k]), std::fabs(b.hi[k])) + 1e-9;
            b.lo[k] -= pad;
            b.hi[k] += pad;
        }
        out.push_back(b);
    }
    // Which entries pay (measured): a cube's or a cylinder's test without a root costs ~100 instructions (six
    // IEEE divisions; two quadratics and the caps), a sphere's ends after its discriminant (35) and a gate is decided from
    // rcp products (45) — skipping a handful of those bought nothing on the composite, the teapot and glass_spheres, while a
    // lone cube's frame gained 11 %.  So spheres, triangles and gates get entries only where there are many of them.
    constexpr size_t kManyEntries = 8;
    if (out.size() < kManyEntries) {
        std::vector<rtx_scene::CullBox> costly;
        for (const rtx_scene::CullBox& b : out)
            if (!(b.id & 0x100u) && (shapes[b.id].type == RTX_CUBE || shapes[b.id].type == RTX_CYLINDER)) costly.push_back(b);
        out.swap(costly);
    }
    return out;
}

// Params.cull for one frame: every box of the scene's list projected through the camera.  A camera ray's line meets a
// box only if its pixel lies in the hull of the box's projected corners — provided the box lies wholly on one side of
// the plane through the eye parallel to the image (either side: the line reaches behind the eye, and negative roots
// feed n1); a box that straddles that plane gets no entry.  Two pixels of margin cover the rounding of rayForPixel
// and of the tests themselves.  RTX_NO_TILE_CULL=1 switches it off (A/B; same canvas).
template <class P>
void fill_cull(const rtx_scene* s, const rtx_camera* cam, const RowPlan& plan, bool gates_too, P& p) {
    p.n_cull = 0;
    if (getenv("RTX_NO_TILE_CULL")) return;
    const uint32_t W = cam->horizontal_size, H = cam->vertical_size;
    if (W > 65000u || H > 65000u) return;
    if (plan.strip_count > 1 && plan.strip_rows % 8u != 0u) return;  // a tile's eight rows must be consecutive image rows
    double view[16];  // world -> camera
    if (!invert4(cam->inverse_transform, view)) return;
    const double ps = cam->pixel_size, hw = cam->half_width, hh = cam->half_height;
    if (!(ps > 0.0) || !std::isfinite(ps)) return;
    for (const rtx_scene::CullBox& b : s->cull_boxes) {
        if ((b.id & 0x100u) && !gates_too) continue;
        double x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        int front = 0, back = 0;
        bool ok = true;
        for (int c = 0; c < 8 && ok; ++c) {
            const double q[3] = {(c & 1) ? b.hi[0] : b.lo[0], (c & 2) ? b.hi[1] : b.lo[1], (c & 4) ? b.hi[2] : b.lo[2]};
            double v[4];
            for (int r = 0; r < 4; ++r) v[r] = view[r * 4] * q[0] + view[r * 4 + 1] * q[1] + view[r * 4 + 2] * q[2] + view[r * 4 + 3];
            const double x = v[0] / v[3], y = v[1] / v[3], z = v[2] / v[3];
            const double reach = std::fabs(x) + std::fabs(y) + std::fabs(z);
            if (!std::isfinite(reach) || !(std::fabs(z) > 1e-6 * reach)) {
                ok = false;
                break;
            }
            (z < 0.0 ? front : back) += 1;
            const double px = (hw + x / z) / ps - 0.5, py = (hh + y / z) / ps - 0.5;  // the pixel whose ray's line passes through the corner
            x0 = std::fmin(x0, px);
            x1 = std::fmax(x1, px);
            y0 = std::fmin(y0, py);
            y1 = std::fmax(y1, py);
        }
        if (!ok || (front != 0 && back != 0)) continue;
        if (!std::isfinite(x0) || !std::isfinite(x1) || !std::isfinite(y0) || !std::isfinite(y1)) continue;
        // Margin: two pixels, plus what the f32 tests of a far, small shape can add — the discriminant b^2 - 4ac of a
        // sphere or cylinder D radii away carries an error of ~1e-7 D^2 in units of the squared radius, which moves
        // the silhouette by ~1e-7 D^2 radii, i.e. 1e-7 (D / r) / pixel_size pixels (a 1/3-pixel sphere: 8 pixels).
        // Ten times that; an entry whose margin would exceed 64 pixels is dropped.
        double dist2 = 0.0, r_min = INFINITY;
        for (int k = 0; k < 3; ++k) {
            const double c = 0.5 * (b.lo[k] + b.hi[k]) - (double)cam->inverse_transform[k * 4 + 3] / (double)cam->inverse_transform[15];
            dist2 += c * c;
            r_min = std::fmin(r_min, 0.5 * (b.hi[k] - b.lo[k]));
        }
        const double m = 2.0 + (b.quadratic ? 1e-6 * std::sqrt(dist2) / (std::fmax(r_min, 1e-30) * ps) : 0.0);
        if (!(m <= 64.0)) continue;
        double fx0 = std::floor(x0 - m), fx1 = std::ceil(x1 + m), fy0 = std::floor(y0 - m), fy1 = std::ceil(y1 + m);
        if (fx0 <= 0.0 && fy0 <= 0.0 && fx1 >= W - 1.0 && fy1 >= H - 1.0) continue;  // covers the frame: nothing to gain
        uint32_t r[4];
        if (fx1 < 0.0 || fy1 < 0.0 || fx0 > W - 1.0 || fy0 > H - 1.0) {
            r[0] = 65535u, r[1] = 0u, r[2] = 65535u, r[3] = 0u;  // off the frame: no tile touches it
        } else {
            r[0] = (uint32_t)std::fmax(fx0, 0.0);
            r[1] = (uint32_t)std::fmin(fx1, 65535.0);
            r[2] = (uint32_t)std::fmax(fy0, 0.0);
            r[3] = (uint32_t)std::fmin(fy1, 65535.0);
        }
        uint32_t* e = p.cull[p.n_cull++];
        e[0] = r[0] | (r[1] << 16);
        e[1] = r[2] | (r[3] << 16);
        e[2] = b.id;
        if (p.n_cull == kMaxCull) break;
    }
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
    static const int cap = [] {  // RTX_TAB_MODE=2 / 0: at most that much in LDS (experiments: what a table layout costs)
        const char* e = getenv("RTX_TAB_MODE");
        return e ? atoi(e) : -1;
    }();
    if (cap == (int)kTabGlobal) *tab_mode = (uint32_t)kTabGlobal;
    if (cap == (int)kTabHit && *tab_mode == (uint32_t)kTabAll) *tab_mode = (uint32_t)kTabHit;
    *lds_bytes = *tab_mode == (uint32_t)kTabAll ? (uint32_t)all : (*tab_mode == (uint32_t)kTabHit ? (uint32_t)small : 0u);
}

template <class P>
void fill_scene_params(const rtx_scene* s, P& p, uint32_t depth) {
    std::memset(&p, 0, sizeof p);
    p.shapes = s->d_shapes;
    p.nodes = s->d_nodes;
    p.bvh_nodes = s->d_bvh_nodes;
    p.bvh_tris = s->d_bvh_tris;
    p.twins = s->d_twins;
    static const bool force_bvh = getenv("RTX_FORCE_BVH_KERNEL") != nullptr;  // experiments: the tree-walking variants on a scene without a tree
    p.has_bvh = (s->n_bvh_nodes || force_bvh) ? 1u : 0u;
    p.finite_schlick = s->finite_schlick ? 1u : 0u;
    p.shadow_skip = s->shadow_skip ? 1u : 0u;  // render_impl_t clears it for rtx_opts.all_shadows
    p.materials = s->d_materials;
    p.patterns = s->d_patterns;
    p.n_nodes = s->n_nodes_walk;
    p.n_shapes = s->n_shapes;
    p.n_materials = s->n_materials;
    p.n_patterns = s->n_patterns;
    p.hitrecs = s->d_hitrecs;
    lds_plan(s->n_shapes, s->n_materials, s->n_patterns, &p.tab_mode, &p.lds_bytes);
    p.track_n1 = s->track_n1 ? 1u : 0u;
    p.depth = depth;
    std::memcpy(p.light_position, s->light_position, sizeof p.light_position);
    std::memcpy(p.light_intensity, s->light_intensity, sizeof p.light_intensity);
    using Real = decltype(p.over_eps);
    p.over_eps = static_cast<Real>(0.00001f) * static_cast<Real>(100.0f);  // f32: an f32 product, one ulp below 0.001f (src/world.zig:55)
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
template <bool GEN, class P>
int run_levels(rtx_scene* s, WorkSet& w, P& p, uint32_t pixels, uint32_t level0_items, hipStream_t st, bool prof,
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
    constexpr bool kF64 = std::is_same<P, f64::Params>::value;
    const int cz = (p.canon ? 1 : 0) + (p.has_bvh ? 2 : 0) + (kF64 ? 4 : 0);
    if (s->resident_blocks[cz][0] == 0) {
        s->resident_blocks[cz][0] = resident_blocks<false>(s, p);
        s->resident_blocks[cz][1] = resident_blocks<true>(s, p);
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
        launch_combine(p, (uint32_t)level, blocks, st);
        mark(-1);
    }
    // Frame totals (ray counts by level, hit kinds) only when the caller reads them back.
    if (want_totals)
        hipLaunchKernelGGL(k_accumulate, dim3(1), dim3(64), 0, st, p.counters, w.d_totals, level0_items ? pixels : 0u, last_level,
                           first ? 1u : 0u);
    HIP_TRY(hipGetLastError());
#ifdef RTX_EXP_DUMP_SHARDS
    if (getenv("RTX_DUMP_SHARDS")) {  // experiment: how evenly a level's rays are spread over the shards
        HIP_TRY(hipStreamSynchronize(st));
        static Counters hc;
        HIP_TRY(hipMemcpy(&hc, p.counters, sizeof(Counters), hipMemcpyDeviceToHost));
        for (uint32_t l = 0; l < last_level; ++l) {
            uint64_t mn = ~0ull, mx = 0, sum = 0, cmn = ~0ull, cmx = 0, csum = 0;
            for (uint32_t k = 0; k < kShards; ++k) {
                const uint64_t a = (uint32_t)hc.shard[k].alloc[l], r = (uint32_t)hc.shard[k].refr[l], n = a + r;
                const uint64_t c = (a + 63) / 64 + (r + 63) / 64;
                mn = n < mn ? n : mn; mx = n > mx ? n : mx; sum += n;
                cmn = c < cmn ? c : cmn; cmx = c > cmx ? c : cmx; csum += c;
            }
            fprintf(stderr, "level %u in: rays per shard min %llu mean %.0f max %llu (max/mean %.3f); chunks min %llu mean %.1f max %llu\n", l + 1,
                    (unsigned long long)mn, (double)sum / kShards, (unsigned long long)mx, (double)mx * kShards / (double)sum,
                    (unsigned long long)cmn, (double)csum / kShards, (unsigned long long)cmx);
        }
    }
#endif
    w.counter_set ^= 1u;
    w.counters_dirty = false;
    return RTX_OK;
}

// `fixed`: render through this one work set and allocate nothing (rtx_render_device_static); NULL: the
// scene's own work sets, grown on demand.
template <class P>
int render_impl_t(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, float* d_rgb, uint8_t* d_rgb8,
                  hipStream_t st, rtx_stats* stats, WorkSet* fixed) {
    constexpr bool kF64 = std::is_same<P, f64::Params>::value;
    int rc = check_camera(cam);
    if (rc) return rc;
    uint32_t depth = (opts && opts->depth) ? opts->depth : kMaxDepth;
    if (depth > kMaxDepth) return fail(RTX_E_INVALID, "depth %u exceeds ray_depth_limit %d", depth, kMaxDepth);
    if (opts && opts->strip_count > 1 && opts->strip_index >= opts->strip_count)
        return fail(RTX_E_INVALID, "strip_index %u >= strip_count %u", opts->strip_index, opts->strip_count);
    if (s->broken) return fail(RTX_E_INVALID, "the scene is inconsistent after a failed rtx_scene_refit: destroy it");
    HIP_TRY(hipSetDevice(s->device));
    (void)hipGetLastError();  // an error some earlier, unrelated HIP call of this thread left behind is not this frame's

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
    // alone to be timed.  rtx_opts.lanes picks; 0 = one lane.  Two lanes are opt-in because what
    // they gain is small and what they can lose is not: the runtime multiplexes streams onto a few
    // hardware queues (GPU_MAX_HW_QUEUES, 4 here), and which queue a stream gets depends on what
    // else the process created before.  On two queues the halves overlap (4K composite frame 1.69
    // instead of 1.71 ms, 8K 5.65 instead of 6.10 ms); on one queue they run one after the other,
    // each with the longer tails of a half-size batch (2.13 ms - the same library, the same frame,
    // in bench.py's process; kernel traces of both in DESIGN.md section 5c).  A caller that wants
    // them measures both in its own process.  Lane streams at different priorities (separate
    // queue pools) were tried as a way to force two queues and hung that same process: not kept.
    // RTX_LANES sets the default for opts.lanes == 0 (A/B).
    static const int lanes_env = [] {
        const char* e = getenv("RTX_LANES");
        return e ? atoi(e) : 0;
    }();
    int lanes = (int)((opts && opts->lanes) ? opts->lanes : (lanes_env > 0 ? (uint32_t)lanes_env : 0u));
    if (lanes > kLanes) return fail(RTX_E_INVALID, "lanes %d exceeds %d", lanes, kLanes);
    if (lanes == 0) lanes = 1;
    if (profile || plan.rows_local < 64u || fixed) lanes = 1;
    // rtx_opts.work_set: which of the scene's sets of work buffers a single-lane frame runs through.  Frames on different
    // sets (and different streams) may be in flight together; frames on one set must be ordered by their caller.
    const int set0 = (int)((opts && !fixed) ? opts->work_set : 0u);
    if (set0 >= kWorkSets) return fail(RTX_E_INVALID, "work_set %d: 0..%d", set0, kWorkSets - 1);
    if (set0 != 0 && lanes > 1) return fail(RTX_E_INVALID, "work_set %d: a frame cut into lanes uses sets 0 and 1", set0);
    if (set0 != 0 && profile) return fail(RTX_E_INVALID, "work_set %d: frames with per-launch events run through set 0", set0);
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
    P p;
    fill_scene_params(s, p, depth);
    {
        const float* ci = cam->inverse_transform;
        const bool cam_canon = ci[12] == 0.0f && ci[13] == 0.0f && ci[14] == 0.0f && ci[15] == 1.0f;
        static const bool off = getenv("RTX_NO_CANON") != nullptr;  // A/B switch for experiments
        p.canon = (s->canon && cam_canon && !off && !(opts && opts->full_w)) ? 1u : 0u;
    }
    {
        static const bool off = getenv("RTX_NO_SHADOW_SKIP") != nullptr;  // A/B switch for experiments
        if (off || (opts && opts->all_shadows)) p.shadow_skip = 0u;
    }
    if (opts && opts->containers > RTX_CONTAINERS_BOOK) return fail(RTX_E_INVALID, "containers %u: RTX_CONTAINERS_V1 or RTX_CONTAINERS_BOOK", opts->containers);
    if (opts && opts->containers == RTX_CONTAINERS_BOOK && s->track_n1) {
        // every level of the frame runs the plain variants (see book_kernel): one ray format, no LDS tables
        p.book = 1u;
        p.canon = 0u;
        p.tab_mode = (uint32_t)kTabGlobal;
        p.lds_bytes = 0u;
    }
    if (p.twins && !p.book && s->track_n1) {
        // twin shapes behind different group boxes: every level runs the plain variants (see twin_kernel)
        p.canon = 0u;
        p.tab_mode = (uint32_t)kTabGlobal;
        p.lds_bytes = 0u;
    }
    if (kF64) {
        if (p.book) return fail(RTX_E_INVALID, "RTX_REAL_F64 renders with the reference's own container walk only");
        if (fixed) return fail(RTX_E_INVALID, "rtx_render_device_static renders in f32 only");
        // canonical frames: hit records, materials and patterns in LDS if they fit (the 160-byte shape records stay in
        // global memory: TAB 1 is not instantiated over double); four-lane frames: everything in global memory
        if (!p.canon || p.tab_mode == (uint32_t)kTabGlobal) {
            p.tab_mode = (uint32_t)kTabGlobal;
            p.lds_bytes = 0u;
        } else {
            p.tab_mode = (uint32_t)kTabHit;
            p.lds_bytes = (uint32_t)(sizeof(rtx_material) * s->n_materials + sizeof(rtx_pattern) * s->n_patterns + sizeof(HitRec) * s->n_shapes);
        }
    }
    for (int l = 0; l < lanes && !fixed; ++l) {
        rc = ensure_work(s, &s->ws[set0 + l], cap, p.canon == 0, kF64 ? 2u : 1u);
        if (rc) return rc;
    }
    if (fixed) fixed->counters_dirty = true;  // the array is shared by every scene on this GPU and keeps the last call's contents
    if (opts && opts->no_bvh) {
        p.nodes = s->d_nodes_linear;
        p.n_nodes = s->n_nodes;
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
    // camera-ray tile culling: the plain walk only (a twin's test records that the ray reached it; the book walk keeps to itself)
    if (!p.book && !p.twins && !kF64) fill_cull(s, cam, plan, !(opts && opts->no_bvh), p);  // (the f64 reporting kernels have no CULL variant)

    if (!fixed) HIP_TRY(hipEventRecord(s->ws[set0].ev_begin, st));
    // Canvas starts black (canvas.zig:23-25); rows never rendered stay black.  Every rendered
    // pixel is written exactly once, so only a canvas with unrendered rows needs the fill.
    if (!compact && plan.rows_local < H) HIP_TRY(hipMemsetAsync(d_rgb, 0, out_rows * W * 3 * sizeof(float), st));
    // Lane streams start after everything queued on the caller's stream so far (one lane: the
    // caller's stream itself carries the work).
    hipStream_t lane_stream[kLanes];
    for (int l = 0; l < lanes; ++l) lane_stream[l] = lanes == 1 ? st : s->ws[l].stream;
    static const bool trace_lanes = getenv("RTX_TRACE_LANES") != nullptr;  // experiments: host-side progress of a two-lane frame on stderr
    if (lanes > 1) {
        HIP_TRY(hipEventRecord(s->ev_fork, st));
        for (int l = 0; l < lanes; ++l) HIP_TRY(hipStreamWaitEvent(lane_stream[l], s->ev_fork, 0));
        if (trace_lanes) fprintf(stderr, "rtx lanes: forked %d lanes\n", lanes);
    }

    uint32_t batch = 0;
    for (uint32_t row0 = 0; row0 < plan.rows_local; row0 += rows_per_batch, ++batch) {
        const uint32_t n_rows = (plan.rows_local - row0 < rows_per_batch) ? plan.rows_local - row0 : rows_per_batch;
        p.row0 = row0;
        p.n_rows = n_rows;
        const uint32_t items = tiles_x * ((n_rows + 7u) / 8u) * 64u;
        const int l = (int)(batch % (uint32_t)lanes);
        rc = run_levels<true>(s, fixed ? *fixed : s->ws[set0 + l], p, cap, items, lane_stream[l], profile, batch < (uint32_t)lanes, stats != nullptr);
        if (rc) return rc;
    }
    const int lanes_used = batch < (uint32_t)lanes ? (int)batch : lanes;
    if (lanes > 1) {
        for (int l = 0; l < lanes; ++l) {
            HIP_TRY(hipEventRecord(s->ws[l].done, lane_stream[l]));
            HIP_TRY(hipStreamWaitEvent(st, s->ws[l].done, 0));
        }
        if (trace_lanes) fprintf(stderr, "rtx lanes: joined (%u batches queued)\n", batch);
    }
    if (d_rgb8) {
        const size_t n = out_rows * W * 3;
        const bool aligned = (reinterpret_cast<uintptr_t>(d_rgb) % 16u == 0u) && (reinterpret_cast<uintptr_t>(d_rgb8) % 4u == 0u);
        hipLaunchKernelGGL(k_quantise, dim3(2048), dim3(256), 0, st, d_rgb, d_rgb8, n, aligned ? n / 4 : (size_t)0);
    }
    if (!fixed) HIP_TRY(hipEventRecord(s->ws[set0].ev_end, st));
    HIP_TRY(hipGetLastError());

    if (stats) {
        std::memset(stats, 0, sizeof *stats);
        Totals t{}, part[kLanes];
        for (int l = 0; l < lanes_used; ++l) HIP_TRY(hipMemcpyAsync(&part[l], s->ws[set0 + l].d_totals, sizeof(Totals), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        for (int l = 0; l < lanes_used; ++l) {
            for (int k = 0; k < kLevels; ++k) {
                t.level_rays[k] += part[l].level_rays[k];
                t.level_hits[k] += part[l].level_hits[k];
            }
            t.nodes += part[l].nodes;
            t.hits_patterned += part[l].hits_patterned;
            t.hits_transparent += part[l].hits_transparent;
            t.shadow_skipped += part[l].shadow_skipped;
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
        stats->shadow_rays_skipped = t.shadow_skipped;
        stats->bvh_box_tests = t.bvh_box;
        stats->bvh_triangle_tests = t.bvh_tri;
        stats->gate_tests = t.gate;
        for (int k = 0; k < 6; ++k) stats->shape_tests[k] = t.shape[k];
        stats->rows_rendered = plan.rows_local;
        stats->kernel_flags = (p.canon ? RTX_KERNEL_CANONICAL_W : 0u) | (p.has_bvh ? RTX_KERNEL_BVH : 0u) |
                              (p.tab_mode == (uint32_t)kTabAll ? RTX_KERNEL_LDS_TABLES : 0u) | (p.tab_mode == (uint32_t)kTabHit ? RTX_KERNEL_LDS_HIT_RECORDS : 0u) | (s->track_n1 ? RTX_KERNEL_TRACK_N1 : 0u) |
                              (kF64 ? RTX_KERNEL_F64 : 0u);
        HIP_TRY(hipEventElapsedTime(&stats->ms_total, s->ws[set0].ev_begin, s->ws[set0].ev_end));
        if (profile) {
            rc = collect_profile(s, stats);
            if (rc) return rc;
        }
    }
    return RTX_OK;
}

// `fixed`: render through this one work set and allocate nothing (rtx_render_device_static); NULL: the
// scene's own work sets, grown on demand.  rtx_opts.real picks the arithmetic.
int render_impl(rtx_scene* s, const rtx_camera* cam, const rtx_opts* opts, float* d_rgb, uint8_t* d_rgb8,
                hipStream_t st, rtx_stats* stats, WorkSet* fixed = nullptr) {
    if (opts && opts->real > RTX_REAL_F64) return fail(RTX_E_INVALID, "real %u: RTX_REAL_F32 or RTX_REAL_F64", opts->real);
    if (opts && opts->real == RTX_REAL_F64) return render_impl_t<f64::Params>(s, cam, opts, d_rgb, d_rgb8, st, stats, fixed);
    return render_impl_t<f32::Params>(s, cam, opts, d_rgb, d_rgb8, st, stats, fixed);
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

#ifdef RTX_EXP_TIMELINE
extern "C" __attribute__((visibility("default"))) int rtx_exp_timeline_read(unsigned long long* out, int reset) {
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(f32::g_timeline), sizeof(unsigned long long) * kLevels * 8192 * 12));
    if (reset) {
        void* ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(f32::g_timeline)));
        HIP_TRY(hipMemset(ptr, 0, sizeof(unsigned long long) * kLevels * 8192 * 12));
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
    {
        // lighting()'s ambient term is (colour * intensity) * ambient per channel; with every factor finite
        // and not negative (no -0 either) it is never -0 and never NaN, and `ambient + 0 + 0` is `ambient`.
        bool ok = true;
        for (uint32_t i = 0; i < desc->n_materials; ++i) {
            const rtx_material& m = desc->materials[i];
            ok = ok && plain_factor(m.ambient) && plain_factor(m.color[0]) && plain_factor(m.color[1]) && plain_factor(m.color[2]);
        }
        for (uint32_t i = 0; i < desc->n_patterns; ++i)
            for (int k = 0; k < 3; ++k) ok = ok && plain_factor(desc->patterns[i].a[k]) && plain_factor(desc->patterns[i].b[k]);
        s->plain_materials = ok;
        s->shadow_skip = ok && plain_factor(desc->light_intensity[0]) && plain_factor(desc->light_intensity[1]) && plain_factor(desc->light_intensity[2]);
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
    rc = build_bvhs(shapes, nodes, linear, bvh_nodes, bvh_tris, nullptr, device_build ? &deferred : nullptr, &s->h_twins);
    if (rc) return bail(rc);
    if (s->h_twins.n_classes) HIP_TRY_S(upload((void**)&s->d_twins, &s->h_twins, sizeof(TwinTable)));
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
            if (trc == rtx_lbvh::kUnboundable) {
                // The host builder leaves a run with an unboundable triangle on the linear path (rtx_bvh::build_list
                // returns kNoLeaf and build_bvhs skips the class): same scene, same canvas, whichever builder was asked for.
                for (uint32_t member : c.members) nodes[member].reserved[1] = nodes[member].n_shapes;
                continue;
            }
            if (trc) return bail(fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH build: %s", err.c_str()));
            lap("device tree: plan");
            s->device_trees.push_back(rtx_scene::DeviceTree{tree, (uint32_t)total_nodes, (uint32_t)total_tris, c.shapes});
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
    {
        const std::vector<rtx_node> walk = walkable_nodes(nodes);
        s->n_nodes_walk = (uint32_t)walk.size();
        s->cull_boxes = cull_boxes(shapes, walk);
        HIP_TRY_S(upload((void**)&s->d_nodes, walk.data(), sizeof(rtx_node) * walk.size()));
    }
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
#ifdef RTX_EXP_LANE_PRIORITY
        // Experiment only (tools/build_variant.sh prio -DRTX_EXP_LANE_PRIORITY): the two lane streams at the two ends of the
        // priority range, which puts them on hardware queues of different pools.  Round 2 saw bench.py's process hang with
        // this; DESIGN.md 5c records what the bounded re-run of round 3 found.
        {
            int least = 0, greatest = 0;
            HIP_TRY_S(hipDeviceGetStreamPriorityRange(&least, &greatest));
            HIP_TRY_S(hipStreamCreateWithPriority(&w.stream, hipStreamNonBlocking, (&w == &s->ws[0]) ? greatest : least));
        }
#else
        HIP_TRY_S(hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
#endif
        HIP_TRY_S(hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
        HIP_TRY_S(hipEventCreate(&w.ev_begin));
        HIP_TRY_S(hipEventCreate(&w.ev_end));
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
        if (w.ev_begin) (void)hipEventDestroy(w.ev_begin);
        if (w.ev_end) (void)hipEventDestroy(w.ev_end);
        if (w.stream) (void)hipStreamDestroy(w.stream);
    }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->d_canvas) (void)hipFree(s->d_canvas);
    if (s->d_shapes) (void)hipFree(s->d_shapes);
    if (s->d_hitrecs) (void)hipFree(s->d_hitrecs);
    if (s->d_twins) (void)hipFree(s->d_twins);
    if (s->d_nodes) (void)hipFree(s->d_nodes);
    if (s->d_nodes_linear) (void)hipFree(s->d_nodes_linear);
    for (rtx_scene::DeviceTree& t : s->device_trees) rtx_lbvh::destroy(t.tree);
    s->device_trees.clear();
    if (s->d_bvh_nodes) (void)hipFree(s->d_bvh_nodes);
    if (s->d_bvh_tris) (void)hipFree(s->d_bvh_tris);
    if (s->d_materials) (void)hipFree(s->d_materials);
    if (s->d_patterns) (void)hipFree(s->d_patterns);
    for (hipEvent_t e : s->prof_events) (void)hipEventDestroy(e);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

int rtx_scene_refit(rtx_scene* s, const rtx_scene_desc* desc) {
    if (!s) return fail(RTX_E_INVALID, "scene is NULL");
    if (s->broken) return fail(RTX_E_INVALID, "the scene is inconsistent after a failed rtx_scene_refit: destroy it");
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
    // Frames may be in flight on any stream of the device (caller streams, lane streams: all non-blocking, so
    // nothing orders them against the copies below): wait for all of them before a record changes.
    HIP_TRY(hipDeviceSynchronize());
    std::vector<rtx_shape> shapes(desc->shapes, desc->shapes + desc->n_shapes);
    {
        TwinTable twins{};
        rc = prepare_shapes(shapes, &twins);
        if (rc) return rc;
        if (std::memcmp(&twins, &s->h_twins, sizeof twins) != 0)
            return fail(RTX_E_INVALID, "rtx_scene_refit: the twin classes (rtx_shape.reserved) changed");
    }
    {   // what a tree holds may move, not change kind: a triangle stays a (hittable) triangle.  Material indices stay
        // too: track_n1 / can_spawn / two_children / finite_schlick — and with them the sizes of the ray queues and
        // the ray-tree node regions — were fixed at creation from the materials the shapes referenced then.
        std::vector<rtx_shape> old(s->n_shapes);
        HIP_TRY(hipMemcpy(old.data(), s->d_shapes, sizeof(rtx_shape) * old.size(), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < s->n_shapes; ++i) {
            const bool was_flat = old[i].e0[0] == 0.0f && old[i].e0[1] == 0.0f && old[i].e0[2] == 0.0f && old[i].e0[3] == 0.0f;
            const bool is_flat = shapes[i].e0[0] == 0.0f && shapes[i].e0[1] == 0.0f && shapes[i].e0[2] == 0.0f && shapes[i].e0[3] == 0.0f;
            if (old[i].type != shapes[i].type || (shapes[i].type == RTX_TRIANGLE && was_flat != is_flat))
                return fail(RTX_E_INVALID, "rtx_scene_refit: shape %u changed its type (or became / stopped being degenerate)", i);
            if (old[i].material != shapes[i].material)
                return fail(RTX_E_INVALID, "rtx_scene_refit: shape %u changed its material index (%u -> %u); materials stay as created", i,
                            old[i].material, shapes[i].material);
        }
    }
    // The new records go to an array of their own first, and every tree is asked whether it can bound its triangles
    // there: a refit that cannot succeed fails here, with the scene as it was.
    rtx_shape* d_new = nullptr;
    HIP_TRY(hipMalloc(&d_new, sizeof(rtx_shape) * (shapes.empty() ? 1 : shapes.size())));
    std::string err;
    {
        hipError_t e = hipMemcpy(d_new, shapes.data(), sizeof(rtx_shape) * shapes.size(), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(d_new);
            return fail(RTX_E_HIP, "rtx_scene_refit: shape upload: %s", hipGetErrorString(e));
        }
        for (const rtx_scene::DeviceTree& t : s->device_trees) {
            const int trc = rtx_lbvh::check(t.tree, d_new, s->stream, err);
            if (trc) {
                (void)hipFree(d_new);
                return fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH refit: %s", err.c_str());
            }
        }
    }
    // From here on device records change: an error leaves the scene unusable (s->broken) instead of half-updated.
    s->broken = true;
    (void)hipFree(s->d_shapes);
    s->d_shapes = d_new;
    // the scene-wide switches follow the new records
    s->canon = desc->light_position[3] == 1.0f;
    for (const rtx_shape& sh : shapes) {
        const float* m = sh.inverse;
        if (!(m[12] == 0.0f && m[13] == 0.0f && m[14] == 0.0f && m[15] == 1.0f)) s->canon = false;
        if (sh.type == RTX_CUBE && !(m[3] == 0.0f && m[7] == 0.0f && m[11] == 0.0f)) s->canon = false;
    }
    std::memcpy(s->light_position, desc->light_position, sizeof s->light_position);
    std::memcpy(s->light_intensity, desc->light_intensity, sizeof s->light_intensity);
    s->shadow_skip = s->plain_materials && plain_factor(desc->light_intensity[0]) && plain_factor(desc->light_intensity[1]) && plain_factor(desc->light_intensity[2]);
    const std::vector<HitRec> hitrecs = build_hitrecs(shapes);
    HIP_TRY(hipMemcpy(s->d_hitrecs, hitrecs.data(), sizeof(HitRec) * hitrecs.size(), hipMemcpyHostToDevice));
    for (uint32_t i = 0; i < desc->n_nodes; ++i) {
        std::memcpy(s->h_nodes[i].bounds_min, desc->nodes[i].bounds_min, sizeof s->h_nodes[i].bounds_min);
        std::memcpy(s->h_nodes[i].bounds_max, desc->nodes[i].bounds_max, sizeof s->h_nodes[i].bounds_max);
    }
    {
        const std::vector<rtx_node> walk = walkable_nodes(s->h_nodes);  // boxes change, the structure (and so the count) does not
        s->cull_boxes = cull_boxes(shapes, walk);
        HIP_TRY(hipMemcpy(s->d_nodes, walk.data(), sizeof(rtx_node) * walk.size(), hipMemcpyHostToDevice));
    }
    {
        std::vector<rtx_node> linear = s->h_nodes;
        for (rtx_node& n : linear) {
            n.reserved[0] = 0;
            n.reserved[1] = n.n_shapes;
        }
        HIP_TRY(hipMemcpy(s->d_nodes_linear, linear.data(), sizeof(rtx_node) * linear.size(), hipMemcpyHostToDevice));
    }
    for (const rtx_scene::DeviceTree& t : s->device_trees) {
        const int trc = rtx_lbvh::refit(t.tree, s->d_shapes, s->d_bvh_nodes, t.node_base, s->d_bvh_tris, t.tri_base, s->stream, err);
        if (trc) return fail(trc == -3 ? RTX_E_HIP : RTX_E_INVALID, "device BVH refit: %s", err.c_str());
    }
    s->broken = false;
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
    rc = build_bvhs(shapes, nodes, linear, bn, bt, &trees);
    if (rc) return rc;
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
    if (s->broken) return fail(RTX_E_INVALID, "the scene is inconsistent after a failed rtx_scene_refit: destroy it");
    HIP_TRY(hipSetDevice(s->device));
    // Canonical kernels need canonical rays too: origins with w == 1, directions with w == 0.
    bool canon = s->canon;
    for (uint32_t i = 0; i < n && canon; ++i) canon = origins[4 * i + 3] == 1.0f && directions[4 * i + 3] == 0.0f;
    const bool twins = s->d_twins && s->track_n1;  // the plain variants, as in render_impl_t
    if (twins) canon = false;
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
    if (twins) {
        p.tab_mode = (uint32_t)kTabGlobal;
        p.lds_bytes = 0u;
    }
    p.canvas = s->d_canvas;
    p.root_count = n;
    rc = run_levels<false>(s, w, p, n, n, st, false, true, false);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(rgb_out, s->d_canvas, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    return RTX_OK;
}

}  // extern "C"
