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

void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order, uint32_t &root) {
  nodes.clear();
  order.resize(prims.size());
  root = BVH_LEAF_FLAG;  // empty leaf
  if (prims.empty()) return;
  if (leaf_size < 1) leaf_size = 1;
  if (leaf_size > BVH_MAX_LEAF) leaf_size = BVH_MAX_LEAF;
  for (int attempt = 0; attempt < 2; attempt++) {
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
