// Host-side BVH construction for the device layout (Node64 + leaf-ordered TriRec).
//
// Topology contract: identical to BVHAccel::new with SplitMethod::SAH (accelerators/src/bvh/mod.rs:43-153,
// sah.rs:26-367) — same bucketed SAH decisions, same partition order (itertools::partition's two-pointer swap), same
// depth-first leaf order — so that closest-hit results agree with the reference even on exact-t ties and on the
// rays the un-widened z slab (bounds3.rs:315-319) culls.  Only the in-memory encoding differs (scene_types.h).
//
// Unlike the reference's single-threaded recursion the build forks subtrees onto a thread pool; each node's
// decision depends only on its own primitive range, so the result is independent of the schedule.
#pragma once
#include "scene_types.h"
#include <cstddef>
#include <vector>

namespace phost {

struct BuildInput {
    const float* P;         // world-space vertices, 3 floats each
    const uint32_t* idx;    // 3 per triangle
    size_t n_tris;
    const uint32_t* tri_flags;  // per-triangle PH_TRI_BOGUS/ALPHA0/SALPHA0 bits (LAST is set by the builder)
    const uint32_t* tri_mesh = nullptr;  // mesh id per triangle (copied into TriRec::mesh), may be null
    // Optional primitive list (default: triangles 0..n_tris-1): entry = triangle id, or PH_ITEM_INST | k for a
    // TransformedPrimitive whose world bound is inst_bounds[6k..6k+5] = {lo xyz, hi xyz}.
    const uint32_t* items = nullptr;
    size_t n_items = 0;
    const float* inst_bounds = nullptr;
};
#define PH_ITEM_INST 0x80000000u

struct BuildOutput {
    std::vector<Node64> nodes;
    std::vector<TriRec> tris;        // leaf-contiguous order
    uint32_t root_ref = PH_INVALID_REF;
    float root_lo[3] = {0, 0, 0}, root_hi[3] = {0, 0, 0};
    // statistics with the reference's names (bvh/common.rs:8-23)
    size_t interior_nodes = 0, leaf_nodes = 0, total_nodes = 0, max_leaf_prims = 0;
    int max_depth = 0;
    double build_seconds = 0;
};

// split_method: 0 SAH, 1 HLBVH, 3 EqualCounts.  (2 Middle is not offered: it panics in the reference, DESIGN.md.)
// Returns 0 on success, -1 on invalid arguments, -2 where the reference's HLBVH build would hit an assertion.
int build_bvh(const BuildInput& in, int split_method, int max_prims_in_node, int n_threads, BuildOutput& out);

// ---- scenes with object instances: a FOREST of aggregates ------------------------------------------------------------------------------------------------
// The reference makes one accelerator per instanced object when the object is first instanced (api/src/lib.rs:953-971) and one over the scene's own primitives, among them the
// TransformedPrimitives (core/src/primitives/transformed_primitive.rs:33-73).  Here all of them share one node array and one TriRec array, laid out
// [scene | object | object ..] (objects in the order of their first ObjectInstance); tree 0 is the scene's.
struct InstancedScene {
    const uint32_t* obj_tri0; const uint32_t* obj_tri1; size_t n_objects;   // per object definition: its triangle range
    const uint32_t* inst_object; const float* inst_i2w; size_t n_inst;      // per ObjectInstance: the object and the row-major instance-to-world matrix
    const uint32_t* top_items; size_t n_top;                                // the scene's primitive list: triangle id, or PH_ITEM_INST | instance
};
struct ForestSpec {
    const uint32_t* items; size_t n_items;          // [the scene's items | object's triangles | ..]
    const uint32_t* tree_start; uint32_t n_trees;   // n_trees + 1 entries
    const uint32_t* inst_tree; const float* inst_i2w; size_t n_inst;   // per instance: the tree of its object (>= 1)
};
struct ForestLayout {
    std::vector<uint32_t> items, tree_start, inst_tree;
    ForestSpec spec(const InstancedScene& sc) const { return ForestSpec{items.data(), items.size(), tree_start.data(), (uint32_t)(tree_start.size() - 1), inst_tree.data(), sc.inst_i2w, sc.n_inst}; }
};
struct ForestTreeOut { uint32_t root_ref; float lo[3], hi[3]; uint32_t n_items; };
void forest_layout(const InstancedScene& sc, ForestLayout& out);
// Transform::transform_bounds (core/src/geometry/transform.rs:552-561): the eight corners in the reference's order, each through transform_point (:288-302); out = {lo xyz, hi xyz}
void transform_bounds(const float* m, const float* lo, const float* hi, float* out);
// every tree with build_bvh, concatenated as above; out's statistics cover all trees (leaves, largest leaf, deepest tree)
int build_forest_host(const BuildInput& in, const InstancedScene& sc, const ForestLayout& layout, int split_method, int max_prims_in_node, BuildOutput& out, std::vector<ForestTreeOut>& trees);

// HLBVH's SAH over the treelet roots (hlbvh.rs:296-432) on its own, for the device builder (bvh_device.hip): root_bounds = {lo xyz, hi xyz} per treelet.
// Nodes come out in pre-order; a child >= 0 is another UpperNode, a child < 0 is treelet -1 - child.  root likewise (a single treelet: root = -1, no nodes).
// Returns 0, or -2 where the reference's assertions fire (:338, :356, :418).
struct UpperNode { float lo[3], hi[3]; int kid[2]; int axis; };
int build_upper_sah(const float* root_bounds, size_t n_roots, std::vector<UpperNode>& out_nodes, int& out_root);

}  // namespace phost
