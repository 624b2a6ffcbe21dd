// rtx_lbvh.hpp — device-side build and refit of the triangle BVH (rtx_lbvh.hip); library-internal.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/rtx.h"
#include "rtx_bvh.hpp"

namespace rtx_lbvh {

struct Tree;  // device-resident topology and per-node data of one tree, kept for refits

// Returned by plan / check / refit when a triangle of the run cannot be bounded (singular transform or a
// non-finite vertex): the host builder leaves such a run on the linear path, and so does the caller here.
constexpr int kUnboundable = -2;

// Phase 1: boxes, Morton sort, hierarchy, node sizes for the triangles `h_indices` (shape indices into the
// device array `d_shapes`, degenerate triangles already left out).  *n_nodes = nodes per layout.
// Returns 0, or a negative RTX_E_* code with `err` set.
int plan(const rtx_shape* d_shapes, const uint32_t* h_indices, uint32_t n, hipStream_t st, Tree** out, uint32_t* n_nodes, std::string& err);
// Phase 2: the eight layouts at d_nodes[node_base, node_base + 8 * n_nodes) (absolute skips) and the
// triangle records at d_tris[tri_base, tri_base + n); validated on the host before returning.
int emit(Tree* t, const rtx_shape* d_shapes, rtx_bvh::Node* d_nodes, uint32_t node_base, rtx_bvh::Tri* d_tris, uint32_t tri_base, hipStream_t st,
         std::string& err);
// The same tree over moved vertices: boxes and triangle records recomputed, topology kept.
int refit(Tree* t, const rtx_shape* d_shapes, rtx_bvh::Node* d_nodes, uint32_t node_base, rtx_bvh::Tri* d_tris, uint32_t tri_base, hipStream_t st,
          std::string& err);
// Can every triangle of the tree be bounded in `d_shapes` (a candidate shape array for a refit)?  Touches only the
// tree's per-triangle scratch, which the next refit recomputes: a failed check leaves the emitted tree as it was.
int check(Tree* t, const rtx_shape* d_shapes, hipStream_t st, std::string& err);
uint32_t triangles(const Tree* t);
uint32_t nodes_per_layout(const Tree* t);
void destroy(Tree* t);

}  // namespace rtx_lbvh
