// rtx_lbvh.hip — triangle BVH built and refitted on the device (new design: the reference
// has no BVH at all, src/group.zig:55-87 brute-forces every member).
//
// The host builder (rtx_bvh.hpp: binned SAH, 1 ms for the 1.5 k-triangle teapot) takes 70 ms for 104 k and
// 0.25 s for 417 k triangles; this one is the LBVH of Karras 2012 — Morton codes of the triangle centroids,
// one radix sort (hipCUB), one thread per internal node for the hierarchy, one bottom-up pass for the
// boxes — and emits the same structure the kernels walk (rtx_bvh::Node in eight threaded pre-order
// layouts, leaves of at most kLeafTris triangles, rtx_bvh::Tri in leaf order), so traversal code and the
// "pure accelerator" contract are unchanged: every surviving triangle still goes through the reference's
// test and ties still resolve on the shape index, hence the canvas equals brute force bit for bit
// whichever builder made the tree (tests/test_gpu_parity.py).  A refit recomputes the boxes (and the
// triangle records) over the same topology after the vertices moved.
//
// Every tree built here is copied back and validated on the host before a kernel may walk it (forward
// skips inside the parent, each triangle in exactly one leaf of every layout): a malformed skip would
// make the stackless walk loop forever.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "rtx_lbvh.hpp"

namespace rtx_lbvh {

namespace {

constexpr uint32_t kNone = 0xFFFFFFFFu;
constexpr uint32_t kLeafTris = rtx_bvh::kLeafTris;
constexpr uint32_t kShapeIdentityBit = 1u << 8;  // rtx_device.hip kShapeIdentity

#define LB_TRY(expr)                                                                    \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess) {                                                         \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                    \
            return -3; /* RTX_E_HIP */                                                  \
        }                                                                               \
    } while (0)

// Per-node data is exchanged between threads of one launch (bottom-up pass): stores and loads that
// other threads depend on go through agent-scope atomics so that no stale L1 line is ever read.
__device__ __forceinline__ float ld(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint32_t ldu(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stu(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// 4x4 inverse in double (Gauss-Jordan, as rtx_bvh::invert4 on the host); false if singular.
__device__ bool invert4(const double in[16], double out[16]) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = in[r * 4 + c];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (fabs(a[r][c]) > fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 8; ++k) {
                const double t = a[piv][k];
                a[piv][k] = a[c][k];
                a[c][k] = t;
            }
        const double d = a[c][c];
        for (int k = 0; k < 8; ++k) a[c][k] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = a[r][c];
            if (f != 0.0)
                for (int k = 0; k < 8; ++k) a[r][k] -= f * a[c][k];
        }
    }
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) out[r * 4 + c] = a[r][4 + c];
    return true;
}

// float atomics on min / max through the usual order-preserving integer view
__device__ __forceinline__ void atomic_min_float(float* addr, float v) {
    if (v >= 0.0f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_float(float* addr, float v) {
    if (v >= 0.0f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}

// World-space box of every triangle — the host's bound_triangle (rtx_bvh.hpp): vertices through the
// inverse of the shape's inverse in double, padded, rounded outwards — into the leaf slots of the node
// arrays, plus the centroid bounds of the whole set.  `bad` is raised if a triangle cannot be bounded.
__global__ void k_bounds(const rtx_shape* shapes, const uint32_t* indices, uint32_t n, float* tri_box /* n x 6 */,
                         float* scene /* min xyz, max xyz of the centroids */, uint32_t* bad) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const rtx_shape& s = shapes[indices[k]];
    double m[16];
    bool ok = true;
    if (s.flags & kShapeIdentityBit) {
        for (int i = 0; i < 16; ++i) m[i] = (i % 5 == 0) ? 1.0 : 0.0;
    } else {
        double inv[16];
        for (int i = 0; i < 16; ++i) inv[i] = s.inverse[i];
        ok = invert4(inv, m);
    }
    double mn[3] = {INFINITY, INFINITY, INFINITY}, mx[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int v = 0; v < 3 && ok; ++v) {
        double p[3];
        for (int c = 0; c < 3; ++c) p[c] = (double)s.p0[c] + (v == 1 ? (double)s.e0[c] : (v == 2 ? (double)s.e1[c] : 0.0));
        for (int r = 0; r < 3; ++r) {
            const double w = m[r * 4 + 0] * p[0] + m[r * 4 + 1] * p[1] + m[r * 4 + 2] * p[2] + m[r * 4 + 3];
            if (!isfinite(w)) ok = false;
            mn[r] = fmin(mn[r], w);
            mx[r] = fmax(mx[r], w);
        }
    }
    if (!ok) {
        atomicOr(bad, 1u);
        return;
    }
    float* box = tri_box + 6ull * k;
    for (int c = 0; c < 3; ++c) {
        const double mag = fmax(fabs(mn[c]), fabs(mx[c]));
        const double pad = 1e-4 * (mx[c] - mn[c]) + 1e-5 * mag + 1e-6;  // as rtx_bvh::bound_triangle
        const double lo = mn[c] - pad, hi = mx[c] + pad;
        box[c] = nextafterf((float)lo, -INFINITY);
        box[3 + c] = nextafterf((float)hi, INFINITY);
        const float centre = (float)(0.5 * (lo + hi));
        atomic_min_float(&scene[c], centre);
        atomic_max_float(&scene[3 + c], centre);
    }
}

__device__ __forceinline__ uint64_t spread21(uint64_t v) {  // abc -> a00b00c
    v &= 0x1FFFFFull;
    v = (v | (v << 32)) & 0x1F00000000FFFFull;
    v = (v | (v << 16)) & 0x1F0000FF0000FFull;
    v = (v | (v << 8)) & 0x100F00F00F00F00Full;
    v = (v | (v << 4)) & 0x10C30C30C30C30C3ull;
    v = (v | (v << 2)) & 0x1249249249249249ull;
    return v;
}

// 63-bit Morton code of every centroid: x in bits 62, 59, ..., y in 61, 58, ..., z in 60, 57, ...
__global__ void k_morton(const float* tri_box, const float* scene, uint32_t n, uint64_t* keys, uint32_t* vals) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint64_t code = 0;
    for (int c = 0; c < 3; ++c) {
        const float centre = 0.5f * (tri_box[6ull * k + c] + tri_box[6ull * k + 3 + c]);
        const float ext = scene[3 + c] - scene[c];
        float u = ext > 0.0f ? (centre - scene[c]) / ext : 0.0f;
        u = fminf(fmaxf(u, 0.0f), 1.0f);
        const uint64_t q = (uint64_t)fminf(u * 2097152.0f, 2097151.0f);
        code |= spread21(q) << (2 - c);
    }
    keys[k] = code;
    vals[k] = k;
}

// Length of the common prefix of the keys at sorted positions i and j (ties broken on the position).
__device__ __forceinline__ int delta(const uint64_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint64_t a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// Node ids: internal node i in [0, n - 1) keeps id i (the root is 0), the leaf at sorted position k is (n - 1) + k.
struct Topology {
    uint32_t* parent;  // 2n - 1
    uint32_t* left;    // n - 1 (internal nodes)
    uint32_t* right;
    uint32_t* first;   // leftmost sorted position below an internal node
    uint32_t* axis;    // split axis of an internal node
};

// Karras 2012, one thread per internal node.
__global__ void k_hierarchy(const uint64_t* keys, uint32_t n, Topology t) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    const int N = (int)n;
    if (i >= N - 1) return;
    const int d = (delta(keys, N, i, i + 1) - delta(keys, N, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, N, i, i - d);
    int lmax = 2;
    while (delta(keys, N, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int s = lmax / 2; s >= 1; s /= 2)
        if (delta(keys, N, i, i + (l + s) * d) > dmin) l += s;
    const int j = i + l * d;
    const int dnode = delta(keys, N, i, j);
    int s = 0, step = l;
    do {
        step = (step + 1) / 2;
        if (delta(keys, N, i, i + (s + step) * d) > dnode) s += step;
    } while (step > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int lo = i < j ? i : j, hi = i < j ? j : i;
    const uint32_t lc = (lo == gamma) ? (uint32_t)(N - 1 + gamma) : (uint32_t)gamma;
    const uint32_t rc = (hi == gamma + 1) ? (uint32_t)(N - 1 + gamma + 1) : (uint32_t)(gamma + 1);
    t.left[i] = lc;
    t.right[i] = rc;
    t.first[i] = (uint32_t)lo;
    // the first bit in which the two halves differ names the axis the split is along
    const int bit = 63 - dnode;
    t.axis[i] = (dnode < 64 && bit <= 62) ? (uint32_t)((62 - bit) % 3) : 0u;
    t.parent[lc] = (uint32_t)i;
    t.parent[rc] = (uint32_t)i;
    if (i == 0) t.parent[0] = kNone;
}

// Boxes, triangle counts and emitted sizes, bottom-up: every leaf walks towards the root; the first
// thread to reach an internal node leaves, the second finds both children complete.
__global__ void k_bottom_up(uint32_t n, const uint32_t* sorted, const float* tri_box, Topology t, float* box /* (2n-1) x 6 */,
                            uint32_t* count, uint32_t* esize, uint32_t* arrived /* n - 1, zeroed */) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    uint32_t node = (n - 1) + k;
    const float* tb = tri_box + 6ull * sorted[k];
    for (int c = 0; c < 6; ++c) st(&box[6ull * node + c], tb[c]);
    stu(&count[node], 1u);
    stu(&esize[node], 1u);
    uint32_t cur = t.parent[node];
    while (cur != kNone) {
        __threadfence();
        if (atomicAdd(&arrived[cur], 1u) == 0u) return;
        __threadfence();
        const uint32_t l = t.left[cur], r = t.right[cur];
        for (int c = 0; c < 3; ++c) {
            st(&box[6ull * cur + c], fminf(ld(&box[6ull * l + c]), ld(&box[6ull * r + c])));
            st(&box[6ull * cur + 3 + c], fmaxf(ld(&box[6ull * l + 3 + c]), ld(&box[6ull * r + 3 + c])));
        }
        const uint32_t cnt = ldu(&count[l]) + ldu(&count[r]);
        stu(&count[cur], cnt);
        stu(&esize[cur], cnt <= kLeafTris ? 1u : 1u + ldu(&esize[l]) + ldu(&esize[r]));
        cur = t.parent[cur];
    }
}

// The eight threaded layouts (rtx_bvh.hpp): a node is emitted if it holds more than kLeafTris triangles
// (interior) or is the topmost node that does not (a leaf of the emitted tree).  Its index in layout q is
// found by walking to the root: one step per ancestor, plus the emitted size of the sibling wherever the
// sibling comes first — the child on the side a ray of octant q comes from along the ancestor's split axis.
__global__ void k_emit(uint32_t n, Topology t, const float* box, const uint32_t* count, const uint32_t* esize, rtx_bvh::Node* nodes,
                       uint32_t node_base, uint32_t per_layout, uint32_t tri_base) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= 2u * n - 1u) return;
    const uint32_t cnt = count[id];
    const uint32_t p = t.parent[id];
    if (!(cnt > kLeafTris || p == kNone || count[p] > kLeafTris)) return;  // inside a collapsed leaf
    uint32_t idx[rtx_bvh::kLayouts];
    for (uint32_t q = 0; q < rtx_bvh::kLayouts; ++q) idx[q] = 0u;
    uint32_t c = id, up = p;
    while (up != kNone) {
        const bool c_is_right = t.right[up] == c;
        const uint32_t sibling = c_is_right ? t.left[up] : t.right[up];
        const uint32_t sib = esize[sibling], ax = t.axis[up];
        for (uint32_t q = 0; q < rtx_bvh::kLayouts; ++q) {
            const bool right_first = ((q >> ax) & 1u) != 0u;
            idx[q] += 1u + ((c_is_right == right_first) ? 0u : sib);
        }
        c = up;
        up = t.parent[up];
    }
    const bool leaf = cnt <= kLeafTris;
    const uint32_t first = id >= n - 1u ? id - (n - 1u) : t.first[id];
    for (uint32_t q = 0; q < rtx_bvh::kLayouts; ++q) {
        const uint32_t at = node_base + q * per_layout + idx[q];
        rtx_bvh::Node out;
        for (int k = 0; k < 3; ++k) {
            const bool swap = RTX_NEARFAR && ((q >> k) & 1u);  // near / far planes for a ray of octant q (rtx_bvh::Node)
            out.bmin[k] = box[6ull * id + (swap ? 3 : 0) + k];
            out.bmax[k] = box[6ull * id + (swap ? 0 : 3) + k];
        }
        out.skip = at + esize[id];
        out.leaf = leaf ? (((tri_base + first) << 3) | cnt) : rtx_bvh::kNoLeaf;
        nodes[at] = out;
    }
}

// Triangle records in leaf (= sorted) order.
__global__ void k_tris(uint32_t n, const rtx_shape* shapes, const uint32_t* indices, const uint32_t* sorted, rtx_bvh::Tri* tris, uint32_t tri_base) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t shape = indices[sorted[k]];
    const rtx_shape& s = shapes[shape];
    rtx_bvh::Tri out;
    for (int c = 0; c < 3; ++c) {
        out.p0[c] = s.p0[c];
        out.e0[c] = s.e0[c];
        out.e1[c] = s.e1[c];
    }
    out.shape = shape;
    out.flags = s.flags;
    out.pad = 0u;
    tris[tri_base + k] = out;
}

template <class T>
hipError_t alloc(T** p, size_t count) {
    return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(T));
}

}  // namespace

struct Tree {
    uint32_t n = 0;           // triangles
    uint32_t per_layout = 0;  // emitted nodes per layout
    uint32_t* d_indices = nullptr;  // shape index of local triangle k
    uint32_t* d_sorted = nullptr;   // local triangle at sorted position k
    float* d_tri_box = nullptr;
    float* d_box = nullptr;
    uint32_t* d_count = nullptr;
    uint32_t* d_esize = nullptr;
    uint32_t* d_arrived = nullptr;
    float* d_scene = nullptr;
    uint32_t* d_bad = nullptr;
    Topology topo{};
    std::vector<uint32_t> indices;  // host copy (validation)
};

void destroy(Tree* t) {
    if (!t) return;
    void* ptrs[] = {t->d_indices, t->d_sorted, t->d_tri_box, t->d_box, t->d_count, t->d_esize, t->d_arrived, t->d_scene, t->d_bad,
                    t->topo.parent, t->topo.left, t->topo.right, t->topo.first, t->topo.axis};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    delete t;
}

namespace {

// Boxes of the triangles as they are now (and, for a build, the bounds of their centroids).
int compute_bounds(Tree* t, const rtx_shape* d_shapes, hipStream_t st, bool with_scene_bounds, std::string& err) {
    const uint32_t n = t->n, blocks = (n + 255u) / 256u;
    static const float init[6] = {INFINITY, INFINITY, INFINITY, -INFINITY, -INFINITY, -INFINITY};
    LB_TRY(hipMemcpyAsync(with_scene_bounds ? t->d_scene : t->d_scene + 6, init, sizeof init, hipMemcpyHostToDevice, st));
    LB_TRY(hipMemsetAsync(t->d_bad, 0, sizeof(uint32_t), st));
    // a refit keeps the Morton order: the centroid bounds it accumulates go to a scratch copy behind the real ones
    hipLaunchKernelGGL(k_bounds, dim3(blocks), dim3(256), 0, st, d_shapes, t->d_indices, n, t->d_tri_box,
                       with_scene_bounds ? t->d_scene : t->d_scene + 6, t->d_bad);
    uint32_t bad = 0;
    LB_TRY(hipMemcpyAsync(&bad, t->d_bad, sizeof bad, hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    if (bad) {
        err = "a triangle cannot be bounded (singular transform or non-finite vertex)";
        return kUnboundable;
    }
    return 0;
}

int run_bottom_up(Tree* t, hipStream_t st, std::string& err) {
    const uint32_t n = t->n, blocks = (n + 255u) / 256u;
    LB_TRY(hipMemsetAsync(t->d_arrived, 0, sizeof(uint32_t) * (n - 1u), st));
    hipLaunchKernelGGL(k_bottom_up, dim3(blocks), dim3(256), 0, st, n, t->d_sorted, t->d_tri_box, t->topo, t->d_box, t->d_count, t->d_esize,
                       t->d_arrived);
    LB_TRY(hipGetLastError());
    return 0;
}

}  // namespace

int check(Tree* t, const rtx_shape* d_shapes, hipStream_t st, std::string& err) { return compute_bounds(t, d_shapes, st, false, err); }

int plan(const rtx_shape* d_shapes, const uint32_t* h_indices, uint32_t n, hipStream_t st, Tree** out, uint32_t* n_nodes, std::string& err) {
    *out = nullptr;
    if (n < 2u) {
        err = "a device-built tree needs at least two triangles";
        return -1;
    }
    if (n > (1u << 27)) {
        err = "too many triangles for one tree";
        return -1;
    }
    Tree* t = new Tree();
    t->n = n;
    t->indices.assign(h_indices, h_indices + n);
    auto fail = [&](int rc) {
        destroy(t);
        return rc;
    };
#define LB_TRY_T(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            return fail(-3);                                                        \
        }                                                                           \
    } while (0)
    const size_t total = 2ull * n - 1ull;
    LB_TRY_T(alloc(&t->d_indices, n));
    LB_TRY_T(alloc(&t->d_sorted, n));
    LB_TRY_T(alloc(&t->d_tri_box, 6ull * n));
    LB_TRY_T(alloc(&t->d_box, 6ull * total));
    LB_TRY_T(alloc(&t->d_count, total));
    LB_TRY_T(alloc(&t->d_esize, total));
    LB_TRY_T(alloc(&t->d_arrived, n));
    LB_TRY_T(alloc(&t->d_scene, 12));
    LB_TRY_T(alloc(&t->d_bad, 1));
    LB_TRY_T(alloc(&t->topo.parent, total));
    LB_TRY_T(alloc(&t->topo.left, n));
    LB_TRY_T(alloc(&t->topo.right, n));
    LB_TRY_T(alloc(&t->topo.first, n));
    LB_TRY_T(alloc(&t->topo.axis, n));
    LB_TRY_T(hipMemcpyAsync(t->d_indices, h_indices, sizeof(uint32_t) * n, hipMemcpyHostToDevice, st));
    int rc = compute_bounds(t, d_shapes, st, true, err);
    if (rc) return fail(rc);

    // Morton codes, sorted
    uint64_t *d_keys = nullptr, *d_keys_sorted = nullptr;
    uint32_t* d_vals = nullptr;
    void* d_tmp = nullptr;
    auto free_sort = [&] {
        if (d_keys) (void)hipFree(d_keys);
        if (d_keys_sorted) (void)hipFree(d_keys_sorted);
        if (d_vals) (void)hipFree(d_vals);
        if (d_tmp) (void)hipFree(d_tmp);
    };
#define LB_TRY_S(expr)                                                              \
    do {                                                                            \
        hipError_t e_ = (expr);                                                     \
        if (e_ != hipSuccess) {                                                     \
            err = std::string(#expr) + ": " + hipGetErrorString(e_);                \
            free_sort();                                                            \
            return fail(-3);                                                        \
        }                                                                           \
    } while (0)
    LB_TRY_S(alloc(&d_keys, n));
    LB_TRY_S(alloc(&d_keys_sorted, n));
    LB_TRY_S(alloc(&d_vals, n));
    const uint32_t blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_morton, dim3(blocks), dim3(256), 0, st, t->d_tri_box, t->d_scene, n, d_keys, d_vals);
    size_t tmp_bytes = 0;
    LB_TRY_S(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_vals, t->d_sorted, (int)n, 0, 63, st));
    LB_TRY_S(hipMalloc(&d_tmp, tmp_bytes ? tmp_bytes : 1));
    LB_TRY_S(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_vals, t->d_sorted, (int)n, 0, 63, st));
    hipLaunchKernelGGL(k_hierarchy, dim3((n - 1u + 255u) / 256u), dim3(256), 0, st, d_keys_sorted, n, t->topo);
    LB_TRY_S(hipGetLastError());
    rc = run_bottom_up(t, st, err);
    if (rc) {
        free_sort();
        return fail(rc);
    }
    LB_TRY_S(hipMemcpyAsync(&t->per_layout, t->d_esize, sizeof(uint32_t), hipMemcpyDeviceToHost, st));  // emitted size of the root
    LB_TRY_S(hipStreamSynchronize(st));
    free_sort();
#undef LB_TRY_S
#undef LB_TRY_T
    if (t->per_layout == 0u || t->per_layout > 2u * n) {
        err = "device tree has an impossible size";
        return fail(-1);
    }
    *n_nodes = t->per_layout;
    *out = t;
    return 0;
}

uint32_t triangles(const Tree* t) { return t->n; }
uint32_t nodes_per_layout(const Tree* t) { return t->per_layout; }

// Structure check on a host copy: see the file comment.
static int validate(const Tree* t, const std::vector<rtx_bvh::Node>& nodes, const std::vector<rtx_bvh::Tri>& tris, uint32_t node_base,
                    uint32_t tri_base, std::string& err) {
    const uint32_t S = t->per_layout, n = t->n;
    std::vector<uint8_t> member_of;  // shape index -> belongs to this tree
    uint32_t max_shape = 0;
    for (uint32_t s : t->indices) max_shape = s > max_shape ? s : max_shape;
    member_of.assign((size_t)max_shape + 1, 0);
    for (uint32_t s : t->indices) member_of[s] = 1;
    std::vector<uint8_t> once((size_t)max_shape + 1, 0);
    for (uint32_t k = 0; k < n; ++k) {
        const uint32_t s = tris[k].shape;
        if (s > max_shape || !member_of[s] || once[s]) {
            err = "device tree: triangle list is not a permutation of the run";
            return -1;
        }
        once[s] = 1;
    }
    for (uint32_t q = 0; q < rtx_bvh::kLayouts; ++q) {
        const uint32_t lo = node_base + q * S, end = lo + S;
        std::vector<uint8_t> seen(n, 0);
        // pre-order with skips: a stack of subtree ends
        std::vector<uint32_t> ends;
        for (uint32_t k = lo; k < end; ++k) {
            const rtx_bvh::Node& nd = nodes[k - node_base];
            while (!ends.empty() && k >= ends.back()) ends.pop_back();
            const uint32_t limit = ends.empty() ? end : ends.back();
            if (nd.skip <= k || nd.skip > limit) {
                err = "device tree: skip of node " + std::to_string(k) + " leaves its parent";
                return -1;
            }
            if (nd.leaf == rtx_bvh::kNoLeaf) {
                if (nd.skip == k + 1u) {
                    err = "device tree: interior node without children";
                    return -1;
                }
                ends.push_back(nd.skip);
                continue;
            }
            if (nd.skip != k + 1u) {
                err = "device tree: a leaf with a subtree";
                return -1;
            }
            const uint32_t t0 = nd.leaf >> 3, cnt = nd.leaf & 7u;
            if (cnt == 0u || cnt > kLeafTris || t0 < tri_base || (uint64_t)t0 + cnt > (uint64_t)tri_base + n) {
                err = "device tree: leaf out of range";
                return -1;
            }
            for (uint32_t x = t0 - tri_base; x < t0 - tri_base + cnt; ++x) {
                if (seen[x]) {
                    err = "device tree: a triangle sits in two leaves";
                    return -1;
                }
                seen[x] = 1;
            }
        }
        if (nodes[lo - node_base].skip != end) {
            err = "device tree: root skip";
            return -1;
        }
        for (uint32_t x = 0; x < n; ++x)
            if (!seen[x]) {
                err = "device tree: a triangle sits in no leaf";
                return -1;
            }
    }
    return 0;
}

int emit(Tree* t, const rtx_shape* d_shapes, rtx_bvh::Node* d_nodes, uint32_t node_base, rtx_bvh::Tri* d_tris, uint32_t tri_base, hipStream_t st,
         std::string& err) {
    const uint32_t n = t->n, S = t->per_layout;
    if ((uint64_t)tri_base + n >= (1ull << 29)) {
        err = "triangle index does not fit a leaf link";
        return -1;
    }
    hipLaunchKernelGGL(k_emit, dim3((2u * n - 1u + 255u) / 256u), dim3(256), 0, st, n, t->topo, t->d_box, t->d_count, t->d_esize, d_nodes, node_base, S,
                       tri_base);
    hipLaunchKernelGGL(k_tris, dim3((n + 255u) / 256u), dim3(256), 0, st, n, d_shapes, t->d_indices, t->d_sorted, d_tris, tri_base);
    LB_TRY(hipGetLastError());
    std::vector<rtx_bvh::Node> nodes((size_t)S * rtx_bvh::kLayouts);
    std::vector<rtx_bvh::Tri> tris(n);
    LB_TRY(hipMemcpyAsync(nodes.data(), d_nodes + node_base, nodes.size() * sizeof(rtx_bvh::Node), hipMemcpyDeviceToHost, st));
    LB_TRY(hipMemcpyAsync(tris.data(), d_tris + tri_base, tris.size() * sizeof(rtx_bvh::Tri), hipMemcpyDeviceToHost, st));
    LB_TRY(hipStreamSynchronize(st));
    return validate(t, nodes, tris, node_base, tri_base, err);
}

int refit(Tree* t, const rtx_shape* d_shapes, rtx_bvh::Node* d_nodes, uint32_t node_base, rtx_bvh::Tri* d_tris, uint32_t tri_base, hipStream_t st,
          std::string& err) {
    int rc = compute_bounds(t, d_shapes, st, false, err);
    if (rc) return rc;
    rc = run_bottom_up(t, st, err);
    if (rc) return rc;
    // same topology => same counts, sizes and indices: only boxes and triangle records change
    return emit(t, d_shapes, d_nodes, node_base, d_tris, tri_base, st, err);
}

}  // namespace rtx_lbvh
