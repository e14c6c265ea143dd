// occl_check.cpp — the occluder-distance maps of cutrace_amd/csrc/occl.cpp are CONSERVATIVE (CPU harness; tests/test_occl_builder.py).
// For random lights and triangle sets — tiny, huge, needle-shaped, degenerate, spanning cube faces, edges and corners, touching the light's
// axes — every sampled point X of every triangle must find  map[cell the kernel would look up for direction X - L] <= |X - L|,
// for the direction itself and for directions perturbed by a few float ulps (the kernel's arithmetic differs from the builder's).
#include "occl.h"
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>
static int failures = 0;
static void check(const float L[3], const std::vector<OcclTri> &tris, const std::vector<float> &map, std::mt19937 &rng, int samples) {
  std::uniform_real_distribution<float> U(0.f, 1.f);
  for (const OcclTri &t : tris)
    for (int s = 0; s < samples; s++) {
      float a = U(rng), b = U(rng);
      if (s < 3) { a = s == 1; b = s == 2; }                    // the corners themselves
      else if (s < 12) { a = U(rng); b = (s % 3 == 0) ? 0.f : (s % 3 == 1 ? 1.f - a : b * 1e-6f); }  // along the edges
      if (a + b > 1.f) { a = 1.f - a; b = 1.f - b; }
      double X[3];
      for (int k = 0; k < 3; k++) X[k] = (double)t.p[0][k] + a * ((double)t.p[1][k] - t.p[0][k]) + b * ((double)t.p[2][k] - t.p[0][k]);
      const double d[3] = {X[0] - L[0], X[1] - L[1], X[2] - L[2]};
      const double dist = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
      if (!(dist > 0)) continue;
      for (int pert = 0; pert < 5; pert++) {
        float v[3];
        for (int k = 0; k < 3; k++) {
          v[k] = (float)(d[k] / dist);
          if (pert) v[k] = std::nextafterf(v[k], (pert + k) % 2 ? 2.f : -2.f);
          if (pert > 2) v[k] = std::nextafterf(v[k], (pert + k) % 2 ? 2.f : -2.f);
        }
        const uint32_t c = occl_cell_of(v);
        if (!(map[c] <= (float)dist * (1.0f + 1e-6f))) {
          if (failures++ < 10) printf("FAIL: cell %u holds %g but a triangle point lies at %g (light %g %g %g)\n", c, map[c], dist, L[0], L[1], L[2]);
        }
      }
    }
}
// the other way round: directions chosen ON cell boundaries (and face edges), a few ulps to either side — where the kernel's float
// arithmetic may pick the neighbouring cell or face — traced against every triangle in double precision
static bool hit_dist(const double d[3], const OcclTri &t, const float L[3], double &dist) {
  double a[3], b[3], p0[3];
  for (int k = 0; k < 3; k++) { p0[k] = (double)t.p[0][k] - L[k]; a[k] = (double)t.p[1][k] - t.p[0][k]; b[k] = (double)t.p[2][k] - t.p[0][k]; }
  const double pv[3] = {d[1] * b[2] - d[2] * b[1], d[2] * b[0] - d[0] * b[2], d[0] * b[1] - d[1] * b[0]};
  const double det = a[0] * pv[0] + a[1] * pv[1] + a[2] * pv[2];
  if (std::fabs(det) < 1e-300) return false;
  const double tv[3] = {-p0[0], -p0[1], -p0[2]};
  const double u = (tv[0] * pv[0] + tv[1] * pv[1] + tv[2] * pv[2]) / det;
  const double qv[3] = {tv[1] * a[2] - tv[2] * a[1], tv[2] * a[0] - tv[0] * a[2], tv[0] * a[1] - tv[1] * a[0]};
  const double v = (d[0] * qv[0] + d[1] * qv[1] + d[2] * qv[2]) / det;
  const double tt = (b[0] * qv[0] + b[1] * qv[1] + b[2] * qv[2]) / det;
  if (u < -1e-9 || v < -1e-9 || u + v > 1 + 1e-9 || tt <= 0) return false;
  dist = tt * std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  return true;
}
static long check_boundaries(const float L[3], const std::vector<OcclTri> &tris, const std::vector<float> &map, std::mt19937 &rng, int n) {
  std::uniform_real_distribution<double> U(-1.0, 1.0);
  const int R = (int)CTR_OCCL_RES;
  long done = 0;
  for (int s = 0; s < n; s++) {
    const int face = (int)(rng() % 6), major = face / 2, sign = face % 2 ? -1 : 1;
    int k = (int)(rng() % (R + 1));
    if (s % 4 == 0) k = (s % 8 == 0) ? 0 : R;                 // the face's own edge
    double u = 2.0 * k / R - 1.0, w = U(rng);
    if (s % 16 == 3) w = (rng() % 2) ? 1.0 : -1.0;            // ... and corner
    const bool swap = rng() % 2;
    double comp3[3];
    const int col = major == 0 ? 1 : 0, row = major == 2 ? 1 : 2;
    comp3[major] = sign; comp3[col] = swap ? w : u; comp3[row] = swap ? u : w;
    for (int pert = -2; pert <= 2; pert++) {
      float v[3] = {(float)comp3[0], (float)comp3[1], (float)comp3[2]};
      const float nrm = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
      for (int q = 0; q < 3; q++) v[q] /= nrm;                // a unit direction, as the kernel has it
      const int which = swap ? row : col;
      for (int e = 0; e < std::abs(pert); e++) v[which] = std::nextafterf(v[which], pert > 0 ? 2.f : -2.f);
      const double d[3] = {v[0], v[1], v[2]};
      const uint32_t c = occl_cell_of(v);
      for (const OcclTri &t : tris) {
        double dist;
        if (hit_dist(d, t, L, dist) && !(map[c] <= (float)dist * (1.0f + 1e-6f))) {
          if (failures++ < 10) printf("FAIL(boundary): cell %u holds %g but the direction meets a triangle at %g\n", c, map[c], dist);
        }
      }
      done++;
    }
  }
  return done;
}

int main() {
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  std::vector<float> map(CTR_OCCL_CELLS);
  long total = 0;
  for (int scene = 0; scene < 400; scene++) {
    float L[3] = {U(rng) * 2, U(rng) * 2, U(rng) * 2};
    if (scene % 7 == 0) L[0] = L[1] = L[2] = 0.f;
    std::vector<OcclTri> tris;
    const int n = 1 + scene % 40;
    const float size = std::pow(10.f, -3.f + 4.f * (scene % 9) / 8.f);  // 1e-3 ... 10: far smaller than a cell to larger than the cube
    for (int k = 0; k < n; k++) {
      OcclTri t;
      float c[3] = {U(rng) * 3, U(rng) * 3, U(rng) * 3};
      if (scene % 5 == 1) c[k % 3] = L[k % 3];                       // centred on one of the light's coordinate planes
      if (scene % 5 == 2) { c[0] = L[0] + 1.5f; c[1] = L[1] + 1.5f * (k % 2 ? 1 : -1); c[2] = L[2] + U(rng) * 0.01f; }  // on a cube edge
      if (scene % 5 == 3) { for (int q = 0; q < 3; q++) c[q] = L[q] + ((k >> q) & 1 ? 1.f : -1.f); }                    // on a cube corner
      for (int v = 0; v < 3; v++)
        for (int q = 0; q < 3; q++) t.p[v][q] = c[q] + U(rng) * size * (scene % 11 == 4 && q == 0 ? 50.f : 1.f);     // needles
      if (scene % 13 == 5 && k == 0) for (int q = 0; q < 3; q++) t.p[2][q] = t.p[1][q];                                 // zero area
      tris.push_back(t);
    }
    if (scene % 3 == 0) {
      // triangles whose extent ENDS within 1e-9 of a cell boundary or a face edge: the kernel's float arithmetic may put the corner's
      // direction into the neighbouring cell / face, which only the builder's one-cell growth and widened frusta cover
      const int R = (int)CTR_OCCL_RES;
      for (int k = 0; k < 12; k++) {
        const int face = (int)(rng() % 6), major = face / 2, sign = face % 2 ? -1 : 1, col = major == 0 ? 1 : 0, row = major == 2 ? 1 : 2;
        const int kb = (k % 3 == 0) ? R : (int)(rng() % (R + 1));        // a face edge, or an inner boundary
        const double ub = 2.0 * kb / R - 1.0, dlt = (k % 2 ? 1e-9 : -1e-9), dep = 0.5 + (rng() % 100) * 0.05;
        OcclTri t;
        const double pts[3][2] = {{ub + dlt, 0.3 * U(rng)}, {ub + dlt - 0.02 * (1 + rng() % 5), 0.3 * U(rng)}, {ub + dlt - 0.01, 0.3 * U(rng) + 0.05}};
        for (int v = 0; v < 3; v++) {
          double c3[3];
          c3[major] = sign * dep; c3[k % 4 == 1 ? row : col] = pts[v][0] * dep; c3[k % 4 == 1 ? col : row] = pts[v][1] * dep;
          for (int q = 0; q < 3; q++) t.p[v][q] = (float)(c3[q] + L[q]);
        }
        tris.push_back(t);
      }
    }
    const bool ok = occl_build_point_light(L, tris.data(), tris.size(), map.data());
    if (!ok) { for (float x : map) if (x != 0.f) { failures++; printf("FAIL: an unusable map must be all zero\n"); break; } continue; }
    check(L, tris, map, rng, 60);
    total += (long)tris.size() * 60 * 5;
    total += check_boundaries(L, tris, map, rng, 600);
  }
  // the light ON a triangle / a corner at infinity -> zero map
  { OcclTri t = {{{-1, -1, 0}, {1, -1, 0}, {0, 1, 0}}}; float L[3] = {0, 0, 0};
    if (occl_build_point_light(L, &t, 1, map.data()) || map[5] != 0.f) { failures++; printf("FAIL: light on a triangle\n"); }
    t.p[0][0] = INFINITY; L[2] = 1;
    if (occl_build_point_light(L, &t, 1, map.data()) || map[77] != 0.f) { failures++; printf("FAIL: infinite corner\n"); } }
  // an empty scene: nothing is ever in the way
  { float L[3] = {1, 2, 3}; occl_build_point_light(L, nullptr, 0, map.data()); if (map[0] < 1e38f) { failures++; printf("FAIL: empty map\n"); } }
  if (failures) { printf("occl_check: %d FAILURES\n", failures); return 1; }
  printf("occl_check: %ld directions checked, every map conservative\n", total);
  return 0;
}
