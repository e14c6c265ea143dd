// bvh.h — per-mesh bounding-volume hierarchy built on the host at scene upload.
//
// The reference has no acceleration structure beyond one AABB per mesh
// (inc/default_schema.hpp:99-114): every ray that hits the box walks all triangles
// (:133).  The result of that walk is just "smallest valid t, first triangle in file
// order on ties" (:134), which is independent of the ORDER in which triangles are
// tested as long as ties are broken by the original index.  So the kernel is free to
// visit triangles through a hierarchy and skip whole subtrees no lane's ray can touch.
//
// Layout ("children in parent"): only INNER nodes are stored.  One node = one 64-byte
// line = one s_load_dwordx16 and holds the boxes of BOTH children plus what each child is
// (another inner node, or a leaf = a contiguous range of the reordered triangle array).
// A child whose box no lane touches is therefore never loaded at all, and leaves cost no
// node load.  The whole WAVE walks the tree together with a small wave-uniform stack.
#ifndef CUTRACE_AMD_BVH_H
#define CUTRACE_AMD_BVH_H

#include <stdint.h>
#include <vector>

// child descriptor: bit 31 set -> leaf: bits 24..30 = triangle count (1..127), bits 0..23 = first
// triangle (relative to the mesh's first triangle); bit 31 clear -> index of the inner child node
// (relative to the mesh's first node)
#define BVH_LEAF_FLAG 0x80000000u
#define BVH_MAX_LEAF 127u
#define BVH_MAX_DEPTH 60  // the kernel keeps its stack in the 64 lanes of one VGPR

struct DNode {
  // [axis][child]: the two children's boxes interleaved, so that (left, right) of one coordinate is an
  // aligned SGPR pair and one v_pk_fma_f32 evaluates a slab distance for BOTH boxes
  float mn[3][2];        // min corners: mn[a][0] left child, mn[a][1] right child
  float mx[3][2];        // max corners
  uint32_t left, right;  // child descriptors
  uint32_t axis;         // split axis (0,1,2): the child on the ray's near side is visited first
  uint32_t pad;
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte line");

struct BvhInput {
  // per triangle: bounds and centroid
  float mn[3], mx[3], c[3];
};

// Builds a binned-SAH BVH over `n` primitives.  `root` receives the root's child descriptor (a
// leaf descriptor when everything fits one leaf, in which case `nodes` is empty); `order` is the
// permutation such that permuted[i] = original[order[i]].
void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order, uint32_t &root);

#endif
