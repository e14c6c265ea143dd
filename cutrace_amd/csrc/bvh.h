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

// Four-wide node for the per-mesh walk (the top-level tree over meshes keeps DNode): 128 bytes = two
// s_load_dwordx16.  One visit tests FOUR child boxes (pairs of children per v_pk_fma_f32), so the chain of
// dependent scalar loads a cast walks through is about half as long as with two-wide nodes, and the
// per-visit bookkeeping (address, stack, branches) is paid half as often.  Built by collapsing the
// binned-SAH binary tree (largest-area child expanded first).  Children are sorted by centroid along
// `axis`; a ray that points the other way visits them in reverse.  Unused slots hold a far-away point box
// and an empty leaf.
struct DNode4 {
  float lo[3][4];        // min corners: lo[a][child]; (0,1) and (2,3) are aligned SGPR pairs
  float hi[3][4];        // max corners
  uint32_t child[4];     // child descriptors (see above), empty slot = BVH_LEAF_FLAG (leaf of 0 triangles)
  uint32_t axis;         // order axis
  uint32_t pad[3];
};
static_assert(sizeof(DNode4) == 128, "DNode4 must be two 64-byte lines");
#define BVH4_MAX_DEPTH 20  // the walk pushes at most 3 entries per level onto a 64-entry stack

struct BvhInput {
  // per triangle: bounds and centroid
  float mn[3], mx[3], c[3];
};

// Builds a binned-SAH BVH over `n` primitives.  `root` receives the root's child descriptor (a
// leaf descriptor when everything fits one leaf, in which case `nodes` is empty); `order` is the
// permutation such that permuted[i] = original[order[i]].
void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order, uint32_t &root);

// The same tree collapsed to four-wide nodes.  `nodes4` always holds at least one node when there is at
// least one primitive (a mesh that fits one leaf gets a root with one child), node 0 is the root;
// depth <= BVH4_MAX_DEPTH is guaranteed (balanced rebuild otherwise).
void bvh4_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode4> &nodes4,
                std::vector<uint32_t> &order);

#endif
