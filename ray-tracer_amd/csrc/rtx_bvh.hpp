// rtx_bvh.hpp — host-side builder of the triangle BVH (new design: the reference has only
// one optional AABB per Group, src/group.zig:24,58, and brute-forces every member).
//
// Contract: the BVH is a pure accelerator.  It only decides which triangles get tested; the
// test itself is the reference's Moller-Trumbore arithmetic (src/triangle.zig:32-50) and the
// winner is still the minimum (t, insertion index), so results equal brute force.  To never
// cull a triangle the f32 test would accept, boxes are built in double precision from the
// world-space vertices and padded; the device slab test adds its own slack.
//
// Layout: "threaded" pre-order array — a node's first child is the next node, `skip` is the
// index after its subtree — so traversal needs no per-lane stack; eight such arrays per tree,
// one per octant of the ray direction, so that a stackless walk still goes front to back.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "../../include/rtx.h"

namespace rtx_bvh {

constexpr uint32_t kNoLeaf = 0xFFFFFFFFu;
#ifndef RTX_LEAF_TRIS
#define RTX_LEAF_TRIS 4
#endif
#ifndef RTX_SAH_BINS
#define RTX_SAH_BINS 32
#endif
#ifndef RTX_SAH_AXES
#define RTX_SAH_AXES 3
#endif
constexpr uint32_t kLeafTris = RTX_LEAF_TRIS;  // at most 7 (three bits in the leaf link)
constexpr uint32_t kMinTris = 16;  // shorter triangle runs stay on the linear path

// RTX_NEARFAR (default on): a node of layout q holds its box as the planes a ray of octant q meets first and last
// along each axis — bmin[k] / bmax[k] trade places where bit k of q is set — so the walk's slab test needs no
// per-axis min / max (six VALU instructions fewer per box).  0 keeps plain min / max boxes in every layout (A/B).
#ifndef RTX_NEARFAR
#define RTX_NEARFAR 1
#endif
struct Node {  // 32 bytes
    float bmin[3];  // RTX_NEARFAR: the NEAR planes for a ray of this layout's octant
    uint32_t skip;  // absolute index of the first node after this subtree
    float bmax[3];  // RTX_NEARFAR: the FAR planes
    uint32_t leaf;  // kNoLeaf, or (first_tri << 3) | count with count <= kLeafTris
};

struct Tri {  // 48 bytes, in leaf order
    float p0[3];
    uint32_t shape;  // index into the flattened shape array == insertion order
    float e0[3];
    uint32_t flags;  // RTX_SHAPE_* of that shape
    float e1[3];
    uint32_t pad;
};

struct Box {
    double mn[3], mx[3];
    void reset() {
        for (int k = 0; k < 3; ++k) {
            mn[k] = INFINITY;
            mx[k] = -INFINITY;
        }
    }
    void add(const double p[3]) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], p[k]);
            mx[k] = std::max(mx[k], p[k]);
        }
    }
    void add(const Box& b) {
        for (int k = 0; k < 3; ++k) {
            mn[k] = std::min(mn[k], b.mn[k]);
            mx[k] = std::max(mx[k], b.mx[k]);
        }
    }
    double area() const {
        const double x = mx[0] - mn[0], y = mx[1] - mn[1], z = mx[2] - mn[2];
        return x * y + y * z + z * x;
    }
};

struct Prim {
    Box box;
    double centroid[3];
    uint32_t shape;
};

// 4x4 inverse in double (Gauss-Jordan); returns false if singular.
inline bool invert4(const double in[16], double out[16]) {
    double a[4][8];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            a[r][c] = in[r * 4 + c];
            a[r][4 + c] = r == c ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (a[piv][c] == 0.0) return false;
        if (piv != c)
            for (int k = 0; k < 8; ++k) std::swap(a[piv][k], a[c][k]);
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

// World-space box of one triangle shape; false if it cannot be bounded (then no BVH is built
// for its run and the run stays on the linear path).
inline bool bound_triangle(const rtx_shape& s, uint32_t index, Prim& out) {
    double inv[16], m[16];
    for (int k = 0; k < 16; ++k) inv[k] = s.inverse[k];
    if (!invert4(inv, m)) return false;
    double v[3][3];
    for (int k = 0; k < 3; ++k) {
        v[0][k] = s.p0[k];
        v[1][k] = (double)s.p0[k] + s.e0[k];
        v[2][k] = (double)s.p0[k] + s.e1[k];
    }
    out.box.reset();
    for (int i = 0; i < 3; ++i) {
        double w[3];
        for (int r = 0; r < 3; ++r) w[r] = m[r * 4 + 0] * v[i][0] + m[r * 4 + 1] * v[i][1] + m[r * 4 + 2] * v[i][2] + m[r * 4 + 3];
        for (int r = 0; r < 3; ++r)
            if (!std::isfinite(w[r])) return false;
        out.box.add(w);
    }
    // Padding: the f32 test can accept points a few ulp outside the exact triangle; the slack
    // scales with the coordinates involved.
    for (int k = 0; k < 3; ++k) {
        const double mag = std::max(std::fabs(out.box.mn[k]), std::fabs(out.box.mx[k]));
        const double ext = out.box.mx[k] - out.box.mn[k];
        const double pad = 1e-4 * ext + 1e-5 * mag + 1e-6;
        out.box.mn[k] -= pad;
        out.box.mx[k] += pad;
        out.centroid[k] = 0.5 * (out.box.mn[k] + out.box.mx[k]);
    }
    out.shape = index;
    return true;
}

// Number of threaded layouts per tree: one per octant of the ray direction.  The layouts hold the
// same nodes and leaves; they differ in which child of an interior node comes first — the one on
// the side the ray comes from along that node's split axis — so a walk meets the nearer child first
// and the [t_lo, t_hi] range of the visitor culls the farther one more often.
constexpr uint32_t kLayouts = 8;

struct TreeNode {
    Box box;
    int axis = 0;           // split axis of an interior node
    int left = -1, right = -1;  // children (left holds the smaller centroids along `axis`), -1 in a leaf
    uint32_t first_tri = 0, n_tris = 0;  // leaf: triangles [first_tri, first_tri + n_tris) of the Tri array
};

struct Builder {
    std::vector<Tri>& tris;
    const rtx_shape* shapes;
    std::vector<Prim> prims;
    std::vector<TreeNode> tree;

    static float down(double x) { return std::nextafter((float)x, -INFINITY); }
    static float up(double x) { return std::nextafter((float)x, INFINITY); }

    int make_leaf(const Box& box, size_t lo, size_t hi) {
        TreeNode n;
        n.box = box;
        n.first_tri = (uint32_t)tris.size();
        n.n_tris = (uint32_t)(hi - lo);
        for (size_t i = lo; i < hi; ++i) {
            const rtx_shape& s = shapes[prims[i].shape];
            Tri t{};
            for (int k = 0; k < 3; ++k) {
                t.p0[k] = s.p0[k];
                t.e0[k] = s.e0[k];
                t.e1[k] = s.e1[k];
            }
            t.shape = prims[i].shape;
            t.flags = s.flags;
            tris.push_back(t);
        }
        tree.push_back(n);
        return (int)tree.size() - 1;
    }

    int build(size_t lo, size_t hi) {
        Box box, cbox;
        box.reset();
        cbox.reset();
        for (size_t i = lo; i < hi; ++i) {
            box.add(prims[i].box);
            cbox.add(prims[i].centroid);
        }
        const size_t n = hi - lo;
        int axis = 0;
        double ext = -1;
        for (int k = 0; k < 3; ++k)
            if (cbox.mx[k] - cbox.mn[k] > ext) {
                ext = cbox.mx[k] - cbox.mn[k];
                axis = k;
            }
        size_t mid;
        if (n <= kLeafTris || !(ext > 0)) {
            // Leaves hold at most 7 triangles (3 bits); degenerate centroid sets are chunked.
            if (n <= 7) return make_leaf(box, lo, hi);
            mid = lo + n / 2;
        } else {
            // Binned SAH: the cheapest of the kBins - 1 splits of each axis (RTX_SAH_AXES = 1: widest centroid axis only).
            constexpr int kBins = RTX_SAH_BINS;
            double best = INFINITY;
            int best_split = -1, best_axis = axis;
            for (int pass = 0; pass < RTX_SAH_AXES; ++pass) {
                const int ax = RTX_SAH_AXES == 1 ? axis : pass;
                const double e = cbox.mx[ax] - cbox.mn[ax];
                if (!(e > 0)) continue;
                Box bbox[kBins];
                size_t bcount[kBins] = {0};
                for (auto& b : bbox) b.reset();
                const double scale = kBins / e;
                for (size_t i = lo; i < hi; ++i) {
                    int b = (int)((prims[i].centroid[ax] - cbox.mn[ax]) * scale);
                    b = b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                    bbox[b].add(prims[i].box);
                    bcount[b]++;
                }
                Box racc[kBins];  // racc[s] = union of bins s .. kBins-1
                size_t rcount[kBins];
                {
                    Box r;
                    r.reset();
                    size_t nr = 0;
                    for (int b = kBins - 1; b >= 0; --b) {
                        if (bcount[b]) {
                            r.add(bbox[b]);
                            nr += bcount[b];
                        }
                        racc[b] = r;
                        rcount[b] = nr;
                    }
                }
                Box l;
                l.reset();
                size_t nl = 0;
                for (int sidx = 1; sidx < kBins; ++sidx) {
                    if (bcount[sidx - 1]) {
                        l.add(bbox[sidx - 1]);
                        nl += bcount[sidx - 1];
                    }
                    const size_t nr = rcount[sidx];
                    if (!nl || !nr) continue;
                    const double cost = l.area() * (double)nl + racc[sidx].area() * (double)nr;
                    if (cost < best) {
                        best = cost;
                        best_split = sidx;
                        best_axis = ax;
                    }
                }
            }
            if (best_split < 0) {
                mid = lo + n / 2;
                std::nth_element(prims.begin() + lo, prims.begin() + mid, prims.begin() + hi,
                                 [&](const Prim& a, const Prim& b) { return a.centroid[axis] < b.centroid[axis]; });
            } else {
                axis = best_axis;
                const double scale = kBins / (cbox.mx[axis] - cbox.mn[axis]);
                auto bin_of = [&](const Prim& p) {
                    int b = (int)((p.centroid[axis] - cbox.mn[axis]) * scale);
                    return b < 0 ? 0 : (b >= kBins ? kBins - 1 : b);
                };
                // stable_partition keeps the build deterministic
                auto it = std::stable_partition(prims.begin() + lo, prims.begin() + hi,
                                                [&](const Prim& p) { return bin_of(p) < best_split; });
                mid = (size_t)(it - prims.begin());
            }
        }
        TreeNode node;
        node.box = box;
        node.axis = axis;
        const int me = (int)tree.size();
        tree.push_back(node);
        const int l = build(lo, mid);
        const int r = build(mid, hi);
        tree[me].left = l;
        tree[me].right = r;
        return me;
    }

    // One threaded pre-order layout: a ray whose direction is negative along a node's split axis
    // (bit `axis` of `octant` set) comes from the high side and takes the right child first.
    void emit(int t, uint32_t octant, std::vector<Node>& nodes) const {
        const TreeNode& tn = tree[t];
        const size_t me = nodes.size();
        nodes.push_back(Node{});
        for (int k = 0; k < 3; ++k) {
            const bool swap = RTX_NEARFAR && ((octant >> k) & 1u);  // a ray going down this axis meets the high plane first
            (swap ? nodes[me].bmax[k] : nodes[me].bmin[k]) = down(tn.box.mn[k]);
            (swap ? nodes[me].bmin[k] : nodes[me].bmax[k]) = up(tn.box.mx[k]);
        }
        if (tn.left < 0) {
            nodes[me].leaf = (tn.first_tri << 3) | tn.n_tris;
        } else {
            nodes[me].leaf = kNoLeaf;
            const bool flip = (octant >> tn.axis) & 1u;
            emit(flip ? tn.right : tn.left, octant, nodes);
            emit(flip ? tn.left : tn.right, octant, nodes);
        }
        nodes[me].skip = (uint32_t)nodes.size();
    }
};

// Build a BVH over the triangles `indices` of `shapes` (any subset, any order: the walk resolves ties on
// the shape index).  Appends the kLayouts threaded layouts (equal size, one after the other: layout q
// starts at root + q * (nodes[root].skip - root)) to `nodes` and the leaf triangles to `tris`; returns
// the root of layout 0, or kNoLeaf if some triangle cannot be bounded.
inline uint32_t build_list(const rtx_shape* shapes, const std::vector<uint32_t>& indices, std::vector<Node>& nodes,
                           std::vector<Tri>& tris) {
    Builder b{tris, shapes, {}, {}};
    b.prims.reserve(indices.size());
    for (uint32_t i : indices) {
        const rtx_shape& s = shapes[i];
        // A triangle whose e0 is exactly zero can never be hit: det = dot(e0, ...) = 0 fails the
        // |det| < epsilon test (src/triangle.zig:36-37).  The OBJ fan emits one per polygon.
        if (s.e0[0] == 0.0f && s.e0[1] == 0.0f && s.e0[2] == 0.0f && s.e0[3] == 0.0f) continue;
        Prim p;
        if (!bound_triangle(s, i, p)) return kNoLeaf;
        b.prims.push_back(p);
    }
    const uint32_t root = (uint32_t)nodes.size();
    if (b.prims.empty()) {  // nothing can be hit: an empty leaf (in every layout)
        for (uint32_t q = 0; q < kLayouts; ++q) {
            Node n{};
            n.leaf = (uint32_t)(tris.size() << 3);
            n.skip = (uint32_t)nodes.size() + 1;
            for (int k = 0; k < 3; ++k) {  // an empty box: culled whatever the ray (near / far planes as in Builder::emit)
                const bool swap = RTX_NEARFAR && ((q >> k) & 1u);
                n.bmin[k] = swap ? -INFINITY : INFINITY;
                n.bmax[k] = swap ? INFINITY : -INFINITY;
            }
            nodes.push_back(n);
        }
        return root;
    }
    const int top = b.build(0, b.prims.size());
    for (uint32_t q = 0; q < kLayouts; ++q) b.emit(top, q, nodes);
    return root;
}

}  // namespace rtx_bvh
