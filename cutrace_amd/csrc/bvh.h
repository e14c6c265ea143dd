// bvh.h — per-mesh bounding-volume hierarchy built on the host at scene upload.
//
// The reference has no acceleration structure beyond one AABB per mesh
// (inc/default_schema.hpp:99-114): every ray that hits the box walks all triangles
// (:133).  The result of that walk is just "smallest valid t, first triangle in file
// order on ties" (:134), which is independent of the ORDER in which triangles are
// tested as long as ties are broken by the original index.  So the kernel is free to
// visit triangles through a hierarchy and skip whole subtrees no lane's ray can touch.
//
// Layout: nodes in depth-first pre-order with a skip link (stackless "threaded" BVH),
// because the whole WAVE walks the tree together (a subtree is entered when ANY lane
// hits its box): next = hit ? i+1 : skip[i].  One node = one 64-byte line = one
// s_load_dwordx16.  Leaves own a contiguous range of the (reordered) triangle array.
#ifndef CUTRACE_AMD_BVH_H
#define CUTRACE_AMD_BVH_H

#include <stdint.h>
#include <vector>

struct DNode {
  float mnx, mny, mnz;   // box min
  float mxx, mxy, mxz;   // box max
  uint32_t skip;         // node index (relative to the mesh's first node) to continue with when culled
  uint32_t first;        // leaf: first triangle (absolute index into the device triangle array)
  uint32_t count;        // leaf: number of triangles; inner node: 0
  uint32_t pad[7];
};
static_assert(sizeof(DNode) == 64, "DNode must be one 64-byte line");

struct BvhInput {
  // per triangle: bounds and centroid
  float mn[3], mx[3], c[3];
};

// Builds a binned-SAH BVH over `n` primitives.  Returns nodes (pre-order, skip links relative
// to node 0; leaf.first relative to 0 in the PERMUTED order) and `order` = permutation such that
// permuted[i] = original[order[i]].
void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order);

#endif
