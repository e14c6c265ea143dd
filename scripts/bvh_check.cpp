// bvh_check.cpp — the host-side BVH builders (cutrace_amd/csrc/bvh.cpp) on random and degenerate triangle sets, checked for the
// invariants the kernel's walk relies on; meant to run under ASan/UBSan (scripts/cpu_sanitize.sh).
//   every primitive lies in exactly one leaf, a leaf holds 1..leaf_size primitives (a single-leaf mesh: one child),
//   every child box contains the boxes of the primitives below it, depth <= BVH_MAX_DEPTH / BVH4_MAX_DEPTH,
//   child indices are in range, unused four-wide slots are empty leaves.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "bvh.h"

static int fails = 0;
#define CHECK(c, ...) do { if (!(c)) { fails++; printf("FAIL %s:%d: ", __FILE__, __LINE__); printf(__VA_ARGS__); printf("\n"); } } while (0)

static bool contains(const float *mn, const float *mx, const BvhInput &p) {
  for (int a = 0; a < 3; a++) {
    if (std::isnan(p.mn[a]) || std::isnan(p.mx[a])) continue;  // a NaN corner constrains nothing
    if (!(mn[a] <= p.mn[a]) || !(mx[a] >= p.mx[a])) return false;
  }
  return true;
}

struct Walk4 {
  const std::vector<DNode4> &n;
  const std::vector<BvhInput> &prims;
  const std::vector<uint32_t> &order;
  uint32_t leaf_size;
  std::vector<int> seen;
  int max_depth = 0;
  void leaf(uint32_t d, const float *mn, const float *mx, bool only_child) {
    const uint32_t first = d & 0xFFFFFFu, cnt = (d >> 24) & 0x7Fu;
    CHECK(cnt <= leaf_size, "leaf of %u > %u", cnt, leaf_size);
    for (uint32_t k = 0; k < cnt; k++) {
      CHECK(first + k < order.size(), "leaf range");
      if (first + k >= order.size()) return;
      const uint32_t id = order[first + k];
      CHECK(id < prims.size(), "order entry");
      seen[id]++;
      if (mn) CHECK(contains(mn, mx, prims[id]), "leaf box does not contain primitive %u", id);
    }
    (void)only_child;
  }
  void node(uint32_t i, int depth) {
    max_depth = depth > max_depth ? depth : max_depth;
    CHECK(i < n.size(), "node index %u of %zu", i, n.size());
    if (i >= n.size() || depth > 64) return;
    const DNode4 &N = n[i];
    CHECK(N.axis < 3, "axis %u", N.axis);
    for (int c = 0; c < 4; c++) {
      const float mn[3] = {N.lo[0][c], N.lo[1][c], N.lo[2][c]}, mx[3] = {N.hi[0][c], N.hi[1][c], N.hi[2][c]};
      if (N.child[c] & BVH_LEAF_FLAG) leaf(N.child[c], mn, mx, false);
      else {
        node(N.child[c], depth + 1);
        // the child's own children must lie inside this box
        if (N.child[c] < n.size()) {
          const DNode4 &C = n[N.child[c]];
          for (int q = 0; q < 4; q++) {
            if (C.child[q] == BVH_LEAF_FLAG) continue;  // unused slot
            for (int a = 0; a < 3; a++)
              CHECK(!(C.lo[a][q] < mn[a]) && !(C.hi[a][q] > mx[a]), "child box sticks out of its parent's (node %u slot %d axis %d)", N.child[c], q, a);
          }
        }
      }
    }
  }
};

static void run(const char *what, std::vector<BvhInput> prims, uint32_t leaf_size) {
  std::vector<DNode4> n4;
  std::vector<uint32_t> order;
  bvh4_build(prims, leaf_size, n4, order);
  if (prims.empty()) { CHECK(n4.empty(), "%s: nodes for an empty mesh", what); return; }
  CHECK(order.size() == prims.size(), "%s: order size", what);
  CHECK(!n4.empty(), "%s: no root", what);
  Walk4 w{n4, prims, order, leaf_size, std::vector<int>(prims.size(), 0)};
  w.node(0, 0);
  for (size_t i = 0; i < prims.size(); i++) CHECK(w.seen[i] == 1, "%s: primitive %zu in %d leaves", what, i, w.seen[i]);
  CHECK(w.max_depth <= BVH4_MAX_DEPTH, "%s: depth %d", what, w.max_depth);
  // two-wide tree (the top-level tree over meshes)
  std::vector<DNode> n2;
  std::vector<uint32_t> o2;
  uint32_t root = 0;
  bvh_build(prims, 1, n2, o2, root);
  CHECK(o2.size() == prims.size(), "%s: two-wide order", what);
  std::vector<int> seen(prims.size(), 0);
  struct R { static void go(const std::vector<DNode> &n, const std::vector<uint32_t> &o, uint32_t d, int depth, std::vector<int> &seen, int &maxd) {
    maxd = depth > maxd ? depth : maxd;
    if (d & BVH_LEAF_FLAG) { const uint32_t f = d & 0xFFFFFFu, c = (d >> 24) & 0x7Fu; for (uint32_t k = 0; k < c; k++) if (f + k < o.size()) seen[o[f + k]]++; return; }
    if (d >= n.size() || depth > 70) { fails++; printf("FAIL two-wide node index\n"); return; }
    go(n, o, n[d].left, depth + 1, seen, maxd); go(n, o, n[d].right, depth + 1, seen, maxd); } };
  int maxd = 0;
  R::go(n2, o2, root, 0, seen, maxd);
  for (size_t i = 0; i < prims.size(); i++) CHECK(seen[i] == 1, "%s: two-wide: primitive %zu in %d leaves", what, i, seen[i]);
  CHECK(maxd <= BVH_MAX_DEPTH, "%s: two-wide depth %d", what, maxd);
}

static BvhInput box_of(const float *a, const float *b, const float *c) {
  BvhInput p;
  for (int q = 0; q < 3; q++) {
    p.mn[q] = std::fmin(std::fmin(a[q], b[q]), c[q]);
    p.mx[q] = std::fmax(std::fmax(a[q], b[q]), c[q]);
    p.c[q] = (a[q] + b[q] + c[q]) * (1.0f / 3.0f);
  }
  return p;
}

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  auto random_set = [&](size_t n, float spread, float size) {
    std::vector<BvhInput> v(n);
    for (auto &p : v) {
      float a[3], b[3], c[3];
      for (int q = 0; q < 3; q++) { const float o = U(rng) * spread; a[q] = o + U(rng) * size; b[q] = o + U(rng) * size; c[q] = o + U(rng) * size; }
      p = box_of(a, b, c);
    }
    return v;
  };
  for (uint32_t leaf : {1u, 2u, 4u, 8u, 127u}) {
    run("empty", {}, leaf);
    for (size_t n : {1u, 2u, 3u, 4u, 5u, 7u, 8u, 9u, 63u, 64u, 65u, 1000u, 4097u}) run("random", random_set(n, 10.f, 0.5f), leaf);
    run("long thin", random_set(3000, 1000.f, 300.f), leaf);
    { auto v = random_set(2000, 0.f, 0.f); run("all at one point", v, leaf); }                      // coincident centroids: median splits
    { auto v = random_set(2000, 5.f, 0.1f); for (auto &p : v) { p.c[0] = 0; p.mn[0] = p.mx[0] = 0; } run("flat in x", v, leaf); }
    { auto v = random_set(5000, 1.f, 0.01f); for (size_t i = 0; i < v.size(); i++) { const float t = std::ldexp(1.0f, -(int)(i % 120)); for (int q = 0; q < 3; q++) { v[i].mn[q] *= t; v[i].mx[q] *= t; v[i].c[q] *= t; } } run("geometric cluster (deep SAH tree)", v, leaf); }
    { auto v = random_set(500, 5.f, 0.5f); v[17].c[1] = NAN; v[17].mn[1] = NAN; v[99].mx[2] = INFINITY; v[99].c[2] = INFINITY; v[3].mn[0] = -INFINITY; v[3].c[0] = -INFINITY; run("NaN and infinite corners", v, leaf); }
    { auto v = random_set(20000, 1e30f, 1e29f); run("huge coordinates", v, leaf); }
  }
  run("200 000 triangles", random_set(200000, 50.f, 0.2f), 4);
  printf(fails ? "bvh_check: %d FAILURES\n" : "bvh_check: all invariants hold\n", fails);
  return fails ? 1 : 0;
}
