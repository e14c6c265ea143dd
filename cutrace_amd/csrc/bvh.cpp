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
  std::vector<DNode> &nodes;
  std::vector<uint32_t> &order;

  static constexpr int NBINS = 16;

  uint32_t build(uint32_t begin, uint32_t end) {
    const uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    Box b, cb;
    for (uint32_t i = begin; i < end; i++) {
      const BvhInput &p = prims[order[i]];
      b.grow(p.mn, p.mx);
      cb.grow_pt(p.c);
    }
    DNode nd;
    memset(&nd, 0, sizeof(nd));
    nd.mnx = b.mn[0]; nd.mny = b.mn[1]; nd.mnz = b.mn[2];
    nd.mxx = b.mx[0]; nd.mxy = b.mx[1]; nd.mxz = b.mx[2];
    const uint32_t n = end - begin;
    bool leaf = n <= leaf_size;
    uint32_t mid = begin;
    if (!leaf) {
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
        const float lo = cb.mn[a], scale = (float)NBINS / (cb.mx[a] - cb.mn[a]);
        auto it = std::partition(order.begin() + begin, order.begin() + end, [&](uint32_t id) {
          int bi = std::min(NBINS - 1, std::max(0, (int)((prims[id].c[a] - lo) * scale)));
          return bi <= best_bin;
        });
        mid = (uint32_t)(it - order.begin());
      }
      if (mid == begin || mid == end) mid = begin + n / 2;  // coincident centroids: split by count
    }
    if (leaf) {
      nd.first = begin;
      nd.count = n;
      nd.skip = me + 1;
      nodes[me] = nd;
      return me;
    }
    nd.count = 0;
    nodes[me] = nd;
    build(begin, mid);
    build(mid, end);
    nodes[me].skip = (uint32_t)nodes.size();
    return me;
  }
};

}  // namespace

void bvh_build(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
               std::vector<uint32_t> &order) {
  nodes.clear();
  order.resize(prims.size());
  for (uint32_t i = 0; i < prims.size(); i++) order[i] = i;
  if (prims.empty()) return;
  if (leaf_size < 1) leaf_size = 1;
  Builder b{prims, leaf_size, nodes, order};
  b.build(0, (uint32_t)prims.size());
  // keep file order inside every leaf (cheap determinism; ties are broken by original index anyway)
  for (const DNode &n : nodes)
    if (n.count) std::sort(order.begin() + n.first, order.begin() + n.first + n.count);
}
