// bvh.cpp — binned-SAH BVH builder (host).  See bvh.h.
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace {

struct Box {
  float mn[3], mx[3];
  Box() {
    for (int a = 0; a < 3; a++) { mn[a] = std::numeric_limits<float>::infinity(); mx[a] = -mn[a]; }
  }
  void grow(const float *lo, const float *hi) {
    for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], lo[a]); mx[a] = std::max(mx[a], hi[a]); }
  }
  void grow_pt(const float *p) { grow(p, p); }
  float half_area() const {
    float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    if (!(dx >= 0) || !(dy >= 0) || !(dz >= 0)) return 0.f;
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Builder {
  const std::vector<BvhInput> &prims;
  uint32_t leaf_size;
  bool median_only;  // depth-bounded fallback
  std::vector<DNode> &nodes;
  std::vector<uint32_t> &order;
  int max_depth = 0;

  static constexpr int NBINS = 16;

  Box bounds(uint32_t begin, uint32_t end) const {
    Box b;
    for (uint32_t i = begin; i < end; i++) b.grow(prims[order[i]].mn, prims[order[i]].mx);
    return b;
  }

  // returns the child descriptor of the subtree over [begin, end)
  uint32_t build(uint32_t begin, uint32_t end, int depth) {
    max_depth = std::max(max_depth, depth);
    const uint32_t n = end - begin;
    if (n <= leaf_size) return BVH_LEAF_FLAG | (n << 24) | begin;
    Box cb;
    for (uint32_t i = begin; i < end; i++) cb.grow_pt(prims[order[i]].c);
    uint32_t mid = begin;
    int axis = 0;
    {
      float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
      axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2);
    }
    if (!median_only) {
      // binned SAH over the three axes
      float best_cost = std::numeric_limits<float>::infinity();
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        const float lo = cb.mn[a], ext = cb.mx[a] - cb.mn[a];
        if (!(ext > 0.f)) continue;
        Box bins[NBINS];
        uint32_t cnt[NBINS] = {0};
        const float scale = (float)NBINS / ext;
        for (uint32_t i = begin; i < end; i++) {
          const BvhInput &p = prims[order[i]];
          int bi = std::min(NBINS - 1, std::max(0, (int)((p.c[a] - lo) * scale)));
          bins[bi].grow(p.mn, p.mx);
          cnt[bi]++;
        }
        float right_area[NBINS];
        uint32_t right_cnt[NBINS];
        Box acc;
        uint32_t c = 0;
        for (int i = NBINS - 1; i > 0; i--) {
          acc.grow(bins[i].mn, bins[i].mx);
          c += cnt[i];
          right_area[i] = acc.half_area();
          right_cnt[i] = c;
        }
        Box lacc;
        uint32_t lc = 0;
        for (int i = 0; i < NBINS - 1; i++) {
          lacc.grow(bins[i].mn, bins[i].mx);
          lc += cnt[i];
          if (lc == 0 || right_cnt[i + 1] == 0) continue;
          float cost = lacc.half_area() * (float)lc + right_area[i + 1] * (float)right_cnt[i + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = i; }
        }
      }
      if (best_axis >= 0) {
        const int a = best_axis;
        axis = a;
        const float lo = cb.mn[a], scale = (float)NBINS / (cb.mx[a] - cb.mn[a]);
        auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t id) {
          int bi = std::min(NBINS - 1, std::max(0, (int)((prims[id].c[a] - lo) * scale)));
          return bi <= best_bin;
        });
        mid = (uint32_t)(it - order.begin());
      }
    }
    if (mid == begin || mid == end) {
      // coincident centroids / SAH failed / median mode: split by count along the widest axis
      mid = begin + n / 2;
      std::nth_element(order.begin() + begin, order.begin() + mid, order.begin() + end,
                       [&](uint32_t x, uint32_t y) { return prims[x].c[axis] < prims[y].c[axis]; });
    }
    const uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    const uint32_t l = build(begin, mid, depth + 1);
    const uint32_t r = build(mid, end, depth + 1);
    DNode nd;
    memset(&nd, 0, sizeof(nd));
    Box lb = bounds(begin, mid), rb = bounds(mid, end);
    for (int a = 0; a < 3; a++) { nd.mn[a][0] = lb.mn[a]; nd.mx[a][0] = lb.mx[a]; nd.mn[a][1] = rb.mn[a]; nd.mx[a][1] = rb.mx[a]; }
    nd.left = l;
    nd.right = r;
    nd.axis = (uint32_t)axis;
    nodes[me] = nd;
    return me;
  }
};

}  // namespace

static void bvh_build_mode(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
                           std::vector<uint32_t> &order, uint32_t &root, bool force_median);

void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order, uint32_t &root) {
  bvh_build_mode(prims, leaf_size, nodes, order, root, false);
}

static void bvh_build_mode(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
                           std::vector<uint32_t> &order, uint32_t &root, bool force_median) {
  nodes.clear();
  order.resize(prims.size());
  root = BVH_LEAF_FLAG;  // empty leaf
  if (prims.empty()) return;
  if (leaf_size < 1) leaf_size = 1;
  if (leaf_size > BVH_MAX_LEAF) leaf_size = BVH_MAX_LEAF;
  for (int attempt = force_median ? 1 : 0; attempt < 2; attempt++) {
    nodes.clear();
    for (uint32_t i = 0; i < prims.size(); i++) order[i] = i;
    Builder b{prims, leaf_size, attempt == 1, nodes, order};
    root = b.build(0, (uint32_t)prims.size(), 0);
    if (b.max_depth <= BVH_MAX_DEPTH) break;  // else rebuild with balanced (median) splits: depth = log2(n)
  }
  // keep file order inside every leaf (cheap determinism; ties are broken by original index anyway)
  auto sort_leaf = [&](uint32_t d) {
    if (d & BVH_LEAF_FLAG) {
      uint32_t first = d & 0xFFFFFFu, cnt = (d >> 24) & 0x7Fu;
      std::sort(order.begin() + first, order.begin() + first + cnt);
    }
  };
  sort_leaf(root);
  for (const DNode &n : nodes) { sort_leaf(n.left); sort_leaf(n.right); }
}


// ---- collapse to four-wide nodes ----
namespace {

struct Child4 {
  uint32_t desc;        // binary-tree descriptor (leaf, or inner index into the binary node array)
  float mn[3], mx[3];
  float area() const {
    const float dx = mx[0] - mn[0], dy = mx[1] - mn[1], dz = mx[2] - mn[2];
    return dx * dy + dy * dz + dz * dx;
  }
};

struct Collapser {
  const std::vector<DNode> &n2;
  std::vector<DNode4> &n4;
  bool balanced;  // expand both seeds once (depth halves exactly) instead of largest-area-first
  int max_depth = 0;

  static Child4 child_of(const DNode &n, int c) {
    Child4 k;
    k.desc = c ? n.right : n.left;
    for (int a = 0; a < 3; a++) { k.mn[a] = n.mn[a][c]; k.mx[a] = n.mx[a][c]; }
    return k;
  }

  // emits the four-wide node for the binary subtree whose two top children are `seed`; returns its index
  uint32_t emit(std::vector<Child4> kids, int depth) {
    max_depth = std::max(max_depth, depth);
    if (balanced) {
      const size_t n0 = kids.size();
      for (size_t i = 0; i < n0; i++)
        if (!(kids[i].desc & BVH_LEAF_FLAG)) {
          const DNode &n = n2[kids[i].desc];
          kids[i] = child_of(n, 0);
          kids.push_back(child_of(n, 1));
        }
    }
    while (!balanced && kids.size() < 4) {
      int best = -1;
      float best_area = -1.f;
      for (size_t i = 0; i < kids.size(); i++)
        if (!(kids[i].desc & BVH_LEAF_FLAG) && kids[i].area() > best_area) { best_area = kids[i].area(); best = (int)i; }
      if (best < 0) break;
      const DNode &n = n2[kids[best].desc];
      kids[best] = child_of(n, 0);
      kids.push_back(child_of(n, 1));
    }
    // order axis: largest spread of the children's centroids
    float cmin[3] = {INFINITY, INFINITY, INFINITY}, cmax[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (const Child4 &k : kids)
      for (int a = 0; a < 3; a++) {
        const float c = 0.5f * (k.mn[a] + k.mx[a]);
        cmin[a] = std::min(cmin[a], c);
        cmax[a] = std::max(cmax[a], c);
      }
    int axis = 0;
    for (int a = 1; a < 3; a++)
      if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
    std::stable_sort(kids.begin(), kids.end(), [&](const Child4 &x, const Child4 &y) {
      return x.mn[axis] + x.mx[axis] < y.mn[axis] + y.mx[axis];
    });
    const uint32_t me = (uint32_t)n4.size();
    n4.emplace_back();
    DNode4 nd;
    memset(&nd, 0, sizeof(nd));
    nd.axis = (uint32_t)axis;
    for (int c = 0; c < 4; c++) {
      if (c < (int)kids.size()) {
        for (int a = 0; a < 3; a++) { nd.lo[a][c] = kids[c].mn[a]; nd.hi[a][c] = kids[c].mx[a]; }
        if (kids[c].desc & BVH_LEAF_FLAG) {
          nd.child[c] = kids[c].desc;
        } else {
          const DNode &n = n2[kids[c].desc];
          nd.child[c] = emit({child_of(n, 0), child_of(n, 1)}, depth + 1);
        }
      } else {
        // unused slot: a point far away (never inside a finite ray interval) and an empty leaf
        for (int a = 0; a < 3; a++) { nd.lo[a][c] = std::numeric_limits<float>::max(); nd.hi[a][c] = std::numeric_limits<float>::max(); }
        nd.child[c] = BVH_LEAF_FLAG;
      }
    }
    n4[me] = nd;
    return me;
  }
};

}  // namespace

void bvh4_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode4> &nodes4,
                std::vector<uint32_t> &order) {
  nodes4.clear();
  order.clear();
  if (prims.empty()) return;
  for (int attempt = 0; attempt < 2; attempt++) {
    std::vector<DNode> n2;
    uint32_t root = BVH_LEAF_FLAG;
    bvh_build_mode(prims, leaf_size, n2, order, root, attempt == 1);
    nodes4.clear();
    Collapser c{n2, nodes4, attempt == 1};
    if (root & BVH_LEAF_FLAG) {
      // the whole mesh fits one leaf: a root with that single child (keeps the walk free of special cases)
      Child4 k;
      k.desc = root;
      for (int a = 0; a < 3; a++) { k.mn[a] = INFINITY; k.mx[a] = -INFINITY; }
      for (const BvhInput &p : prims)
        for (int a = 0; a < 3; a++) { k.mn[a] = std::min(k.mn[a], p.mn[a]); k.mx[a] = std::max(k.mx[a], p.mx[a]); }
      c.emit({k}, 0);
    } else {
      const DNode &n = n2[root];
      c.emit({Collapser::child_of(n, 0), Collapser::child_of(n, 1)}, 0);
    }
    if (c.max_depth <= BVH4_MAX_DEPTH) break;  // else: balanced binary tree, depth log2(n) -> four-wide depth ~ half
  }
}
