// bvh_build.h — host-side acceleration-structure builder (replaces the driver work behind
// vkCmdBuildAccelerationStructuresKHR, reference src/main.cpp:495-498 (BLAS) and :730-733 (TLAS)).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "rt_device.h"

namespace rt {

struct Aabb {
  float lo[3], hi[3];
};

// Topology kept on the host so a TLAS can be refitted in place (Vulkan UPDATE mode).
struct BuildNode {
  Aabb box;
  int32_t left = -1, right = -1;  // BuildNode indices, -1 for leaves
  uint32_t first = 0, count = 0;  // leaf: range in the reordered primitive list
};

struct BuiltBvh {
  std::vector<BvhNode> nodes;     // node 0 is the root and is always interior
  std::vector<uint32_t> order;    // leaf order -> original primitive index
  std::vector<BuildNode> topo;    // build tree (topo[0] = root)
  std::vector<int32_t> emit_of;   // topo index -> index in nodes (interior) or -1
  Aabb bounds;
  int depth = 0;
  uint32_t leaves = 0;
};

// Binned-SAH BVH2 over primitive boxes.  max_leaf in 1..8; max_depth bounds the tree depth
// (median splits take over where SAH would exceed it), which bounds the traversal stack.
void build_bvh(const Aabb* prim_boxes, uint32_t n, int max_leaf, int max_depth, BuiltBvh& out);

// Recompute every box of `bvh` bottom-up from new primitive boxes, keeping the topology, and
// rewrite bvh.nodes in place.
void refit_bvh(const Aabb* prim_boxes, BuiltBvh& bvh);

// BVH4 obtained by collapsing a BuiltBvh (children of a node = up to 4 subtrees of the BVH2).
struct Bvh4 {
  std::vector<Bvh4Node> nodes;                    // node 0 = root
  std::vector<int32_t> child_topo;                // 4 per node: BuildNode index of each child or -1
  int depth = 0;
  int stack_need = 0;   // worst-case number of stack entries the quad traversal can hold inside this tree
};
// area_driven: expand the child with the largest surface area first (BLAS); otherwise expand in
// order, which makes the topology independent of the boxes (TLAS: refit keeps it).
// direct_ids: leaves hold ~primitive (TLAS instances) instead of ~((first << 3) | (count - 1)).
void collapse_bvh4(const BuiltBvh& b2, bool area_driven, bool direct_ids, Bvh4& out);
// rewrite the child boxes of b4 from b2.topo (after refit_bvh)
void refit_bvh4(const BuiltBvh& b2, Bvh4& b4);

// 32-byte quantized form of bvh.nodes (rt_device.h BvhNodeQ).  q_lo/q_scale receive the dequantisation
// of the tree's bounds.  A missing child (synthetic root of a one-leaf tree) becomes an inverted box no ray can enter; its link repeats the sibling's.
void quantize_bvh2(const BuiltBvh& bvh, std::vector<BvhNodeQ>& out, float q_lo[3], float q_scale[3]);
// the same in steps, for several trees that share ONE quantisation (the TLAS of every frame of a batch): accumulate the bounds of
// each tree, derive the dequantisation, quantize each tree in it
void bvh2_bounds(const BuiltBvh& bvh, double lo[3], double hi[3]);
void quant_params(const double lo[3], const double hi[3], float q_lo[3], float q_scale[3]);
void quantize_bvh2_in(const BuiltBvh& bvh, std::vector<BvhNodeQ>& out, const float q_lo[3], const float q_scale[3]);

// 4-ary records (rt_device.h WideNodeQ) of `count` linked quantized nodes: out[i] holds the grandchildren of
// nodes[i].  Interior links are global indices; node g of this tree sits at nodes[g - base].
void widen_bvh2(const BvhNodeQ* nodes, size_t count, int32_t base, WideNodeQ* out);
// number of interior levels below (and including) node `root` of a quantized tree (links local to `nodes`);
// returns -1 when the links do not form a tree of at most `count` nodes
int bvh2_levels(const BvhNodeQ* nodes, size_t count, int32_t root);

// BLAS helper: boxes + 48-byte packets for an indexed triangle mesh in the reference's layout
// (positions at verts6[6*i .. 6*i+2], object-local uint32 indices).
void build_blas(const float* verts6, const uint32_t* idx, uint32_t n_prims, BuiltBvh& bvh, std::vector<TriPacket>& tris);

}  // namespace rt
