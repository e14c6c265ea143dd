// bvh.cpp — binned-SAH BVH builder (host).  See bvh.h.
#include "bvh.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <future>
#include <limits>
#include <thread>

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

// a primitive as the builder moves it around: the records themselves are partitioned (not an index array into them), so
// that every pass over a range streams through memory — with indices each of the ~3 passes per level was a random access
// into the primitive array, which is where a 64 000-triangle build spent its 100 ms
struct Item {
  BvhInput b;
  uint32_t id;
};

struct Builder {
  std::vector<Item> &items;
  uint32_t leaf_size;
  bool median_only;  // depth-bounded fallback
  std::vector<DNode> &nodes;
  int max_depth = 0;

  static constexpr int NBINS = 16;

  Box bounds(uint32_t begin, uint32_t end) const {
    Box b;
    for (uint32_t i = begin; i < end; i++) b.grow(items[i].b.mn, items[i].b.mx);
    return b;
  }

  // One split of [begin, end): partitions `order` in place, returns the split position, the split axis and the two
  // children's boxes.  Binned SAH over the three axes in ONE pass over the primitives (every primitive is read once and
  // dropped into its bin of each axis); the children's boxes are the accumulated bin boxes on either side of the chosen
  // boundary — the same min / max over the same sets that a second pass would compute.
  uint32_t split(uint32_t begin, uint32_t end, int &axis, Box &lb, Box &rb) {
    const uint32_t n = end - begin;
    Box cb;
    for (uint32_t i = begin; i < end; i++) cb.grow_pt(items[i].b.c);
    uint32_t mid = begin;
    {
      float e0 = cb.mx[0] - cb.mn[0], e1 = cb.mx[1] - cb.mn[1], e2 = cb.mx[2] - cb.mn[2];
      axis = (e0 >= e1 && e0 >= e2) ? 0 : (e1 >= e2 ? 1 : 2);
    }
    bool have_boxes = false;
    if (!median_only) {
      float lo[3], scale[3];
      bool use[3];
      for (int a = 0; a < 3; a++) {
        const float ext = cb.mx[a] - cb.mn[a];
        use[a] = ext > 0.f;
        lo[a] = cb.mn[a];
        scale[a] = use[a] ? (float)NBINS / ext : 0.f;
      }
      // (the sweeps visit the NON-EMPTY bins only — most splits are of 5-30 primitives.  A boundary between two non-empty
      //  bins with empty ones in between has the same two sets, hence the same cost, as the first of them: the chosen split,
      //  and the tree, are unchanged.)
      Box bins[3][NBINS];
      uint32_t cnt[3][NBINS] = {{0}};
      for (uint32_t i = begin; i < end; i++) {
        const BvhInput &p = items[i].b;
        for (int a = 0; a < 3; a++) {
          if (!use[a]) continue;
          const int bi = std::min(NBINS - 1, std::max(0, (int)((p.c[a] - lo[a]) * scale[a])));
          bins[a][bi].grow(p.mn, p.mx);  // (grow ignores a NaN corner: whatever the order, a box is the min / max of the other values)
          cnt[a][bi]++;
        }
      }
      float best_cost = std::numeric_limits<float>::infinity();
      int best_axis = -1, best_bin = -1;
      for (int a = 0; a < 3; a++) {
        if (!use[a]) continue;
        int ne[NBINS], m = 0;  // the non-empty bins, ascending
        for (int i = 0; i < NBINS; i++)
          if (cnt[a][i]) ne[m++] = i;
        if (m < 2) continue;
        float right_area[NBINS];
        uint32_t right_cnt[NBINS];
        Box acc;
        uint32_t c = 0;
        for (int k = m - 1; k > 0; k--) {
          acc.grow(bins[a][ne[k]].mn, bins[a][ne[k]].mx);
          c += cnt[a][ne[k]];
          right_area[k] = acc.half_area();
          right_cnt[k] = c;
        }
        Box lacc;
        uint32_t lc = 0;
        for (int k = 0; k < m - 1; k++) {
          lacc.grow(bins[a][ne[k]].mn, bins[a][ne[k]].mx);
          lc += cnt[a][ne[k]];
          float cost = lacc.half_area() * (float)lc + right_area[k + 1] * (float)right_cnt[k + 1];
          if (cost < best_cost) { best_cost = cost; best_axis = a; best_bin = ne[k]; }
        }
      }
      if (best_axis >= 0) {
        const int a = best_axis;
        axis = a;
        const float l0 = lo[a], sc = scale[a];
        auto it = std::partition(items.begin() + begin, items.begin() + end, [&](const Item &it_) {
          int bi = std::min(NBINS - 1, std::max(0, (int)((it_.b.c[a] - l0) * sc)));
          return bi <= best_bin;
        });
        mid = (uint32_t)(it - items.begin());
        if (mid != begin && mid != end) {
          lb = Box();
          rb = Box();
          for (int i = 0; i <= best_bin; i++)
            if (cnt[a][i]) lb.grow(bins[a][i].mn, bins[a][i].mx);
          for (int i = best_bin + 1; i < NBINS; i++)
            if (cnt[a][i]) rb.grow(bins[a][i].mn, bins[a][i].mx);
          have_boxes = true;
        }
      }
    }
    if (mid == begin || mid == end) {
      // coincident centroids / SAH failed / median mode: split by count along the widest axis
      mid = begin + n / 2;
      std::nth_element(items.begin() + begin, items.begin() + mid, items.begin() + end,
                       [&](const Item &x, const Item &y) { return x.b.c[axis] < y.b.c[axis]; });
    }
    if (!have_boxes) { lb = bounds(begin, mid); rb = bounds(mid, end); }
    return mid;
  }

  static DNode make_node(const Box &lb, const Box &rb, uint32_t l, uint32_t r, int axis) {
    DNode nd;
    memset(&nd, 0, sizeof(nd));
    for (int a = 0; a < 3; a++) { nd.mn[a][0] = lb.mn[a]; nd.mx[a][0] = lb.mx[a]; nd.mn[a][1] = rb.mn[a]; nd.mx[a][1] = rb.mx[a]; }
    nd.left = l;
    nd.right = r;
    nd.axis = (uint32_t)axis;
    return nd;
  }

  // returns the child descriptor of the subtree over [begin, end)
  uint32_t build(uint32_t begin, uint32_t end, int depth) {
    max_depth = std::max(max_depth, depth);
    const uint32_t n = end - begin;
    if (n <= leaf_size) return BVH_LEAF_FLAG | (n << 24) | begin;
    int axis = 0;
    Box lb, rb;
    const uint32_t mid = split(begin, end, axis, lb, rb);
    const uint32_t me = (uint32_t)nodes.size();
    nodes.emplace_back();
    const uint32_t l = build(begin, mid, depth + 1);
    const uint32_t r = build(mid, end, depth + 1);
    nodes[me] = make_node(lb, rb, l, r, axis);
    return me;
  }
};

}  // namespace

static void bvh_build_mode(const std::vector<BvhInput> &prims, uint32_t leaf_size, std::vector<DNode> &nodes,
                           std::vector<uint32_t> &order, uint32_t &root, bool force_median);

// ---- the top of a large tree on several threads ----
// The subtrees below a split cover disjoint ranges of `order`, so they can be built at the same time; each goes into a node
// array of its own (local indices) and the parent splices them behind itself in the order the sequential build uses
// (node, left subtree, right subtree), shifting the inner-child indices.  The TREE is the one the sequential build makes:
// every split sees the same primitives in the same order.  A 64 000-triangle mesh: 103 ms -> ~10 ms of ctr_scene_create
// (bench.py config.scene_create_ms; main.cu:21-30 pays it once per process).
namespace {
struct SubTree {
  std::vector<DNode> nodes;
  uint32_t root = BVH_LEAF_FLAG;  // leaf descriptor, or the index of the subtree's root in `nodes` (always 0)
  int max_depth = 0;
};

int par_levels(size_t n) {
  static const unsigned hw = [] {
    if (const char *e = getenv("CUTRACE_BUILD_THREADS")) return (unsigned)std::max(1, atoi(e));
    const unsigned h = std::thread::hardware_concurrency();
    return h ? h : 1u;
  }();
  int levels = 0;
  while (levels < 5 && (1u << (levels + 1)) <= hw && (n >> (levels + 1)) >= 2048) levels++;  // tasks of >= 2048 primitives
  return levels;
}

SubTree build_par(std::vector<Item> &items, uint32_t leaf_size, bool median_only, uint32_t begin, uint32_t end, int depth, int levels) {
  SubTree out;
  if (levels <= 0 || end - begin <= leaf_size) {
    Builder b{items, leaf_size, median_only, out.nodes};
    out.root = b.build(begin, end, depth);
    out.max_depth = b.max_depth;
    return out;
  }
  std::vector<DNode> unused;
  Builder b{items, leaf_size, median_only, unused};
  int axis = 0;
  Box lb, rb;
  const uint32_t mid = b.split(begin, end, axis, lb, rb);
  std::future<SubTree> lf = std::async(std::launch::async, [&, begin, mid, depth, levels] {
    return build_par(items, leaf_size, median_only, begin, mid, depth + 1, levels - 1);
  });
  SubTree R = build_par(items, leaf_size, median_only, mid, end, depth + 1, levels - 1);
  SubTree L = lf.get();
  auto shifted = [](uint32_t d, uint32_t off) { return (d & BVH_LEAF_FLAG) ? d : d + off; };
  const uint32_t off_l = 1u, off_r = 1u + (uint32_t)L.nodes.size();
  out.nodes.reserve(1 + L.nodes.size() + R.nodes.size());
  out.nodes.push_back(Builder::make_node(lb, rb, shifted(L.root, off_l), shifted(R.root, off_r), axis));
  for (DNode nd : L.nodes) { nd.left = shifted(nd.left, off_l); nd.right = shifted(nd.right, off_l); out.nodes.push_back(nd); }
  for (DNode nd : R.nodes) { nd.left = shifted(nd.left, off_r); nd.right = shifted(nd.right, off_r); out.nodes.push_back(nd); }
  out.root = 0u;
  out.max_depth = std::max(depth, std::max(L.max_depth, R.max_depth));
  return out;
}
}  // namespace

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
    std::vector<Item> items(prims.size());
    for (uint32_t i = 0; i < prims.size(); i++) { items[i].b = prims[i]; items[i].id = i; }
    int max_depth = 0;
    SubTree t = build_par(items, leaf_size, attempt == 1, 0, (uint32_t)prims.size(), 0, par_levels(prims.size()));
    for (uint32_t i = 0; i < prims.size(); i++) order[i] = items[i].id;
    nodes = std::move(t.nodes);
    root = t.root;
    max_depth = t.max_depth;
    if (max_depth <= BVH_MAX_DEPTH) break;  // else rebuild with balanced (median) splits: depth = log2(n)
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
