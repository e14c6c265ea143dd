// occl.cpp — builder of the per-light occluder-distance maps.  See occl.h.
#include "occl.h"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace {

constexpr int R = (int)CTR_OCCL_RES;

struct D3 { double x, y, z; };
inline D3 sub(D3 a, D3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline D3 cross(D3 a, D3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline double comp(const D3 &v, int i) { return i == 0 ? v.x : i == 1 ? v.y : v.z; }

// squared distance from the origin to segment ab
double seg_dist_sq(D3 a, D3 b) {
  const D3 ab = sub(b, a);
  const double l2 = dot(ab, ab);
  double t = l2 > 0.0 ? -dot(a, ab) / l2 : 0.0;
  t = t < 0.0 ? 0.0 : (t > 1.0 ? 1.0 : t);
  const D3 q{a.x + t * ab.x, a.y + t * ab.y, a.z + t * ab.z};
  return dot(q, q);
}

// smallest distance from the origin to triangle abc: the three edges, and the foot of the perpendicular where it falls inside
double tri_dist(D3 a, D3 b, D3 c) {
  double d2 = std::min(seg_dist_sq(a, b), std::min(seg_dist_sq(b, c), seg_dist_sq(c, a)));
  const D3 n = cross(sub(b, a), sub(c, a));
  const double n2 = dot(n, n);
  if (n2 > 0.0) {
    const double k = dot(a, n) / n2;  // foot = k * n
    const D3 q{k * n.x, k * n.y, k * n.z};
    // inside iff the three edge functions agree in sign
    const double e0 = dot(cross(sub(b, a), sub(q, a)), n), e1 = dot(cross(sub(c, b), sub(q, b)), n), e2 = dot(cross(sub(a, c), sub(q, c)), n);
    if (e0 >= 0.0 && e1 >= 0.0 && e2 >= 0.0) d2 = std::min(d2, dot(q, q));
  }
  return std::sqrt(d2);
}

struct FaceAxes { int major, sign, col, row; };
constexpr FaceAxes FACES[6] = {{0, +1, 1, 2}, {0, -1, 1, 2}, {1, +1, 0, 2}, {1, -1, 0, 2}, {2, +1, 0, 1}, {2, -1, 0, 1}};

int face_of(const D3 &v) {
  const double ax = std::fabs(v.x), ay = std::fabs(v.y), az = std::fabs(v.z);
  if (ax >= ay && ax >= az) return v.x < 0 ? 1 : 0;
  if (ay >= az) return v.y < 0 ? 3 : 2;
  return v.z < 0 ? 5 : 4;
}

// Sutherland-Hodgman against the half-space  K * m(v) + s * comp(v, axis) >= 0
int clip(const D3 *in, int n, D3 *out, const FaceAxes &F, double K, int axis, double s) {
  auto f = [&](const D3 &v) { return K * F.sign * comp(v, F.major) + s * comp(v, axis); };
  int m = 0;
  for (int i = 0; i < n; i++) {
    const D3 &p = in[i], &q = in[(i + 1) % n];
    const double fp = f(p), fq = f(q);
    if (fp >= 0.0) out[m++] = p;
    if ((fp >= 0.0) != (fq >= 0.0)) {
      const double t = fp / (fp - fq);
      out[m++] = {p.x + t * (q.x - p.x), p.y + t * (q.y - p.y), p.z + t * (q.z - p.z)};
    }
  }
  return m;
}

inline int cell(double u) {  // [-1, 1] -> [0, R)
  const double c = std::floor((u * 0.5 + 0.5) * R);
  return (int)std::max(0.0, std::min((double)(R - 1), c));
}

void splat(float *map, int face, double u0, double u1, double w0, double w1, float d) {
  const int c0 = std::max(0, cell(u0) - 1), c1 = std::min(R - 1, cell(u1) + 1);
  const int r0 = std::max(0, cell(w0) - 1), r1 = std::min(R - 1, cell(w1) + 1);
  for (int r = r0; r <= r1; r++) {
    float *row = map + ((size_t)face * R + r) * R;
    for (int c = c0; c <= c1; c++) row[c] = std::min(row[c], d);
  }
}

}  // namespace

uint32_t occl_cell_of(const float v[3]) {
  const float ax = std::fabs(v[0]), ay = std::fabs(v[1]), az = std::fabs(v[2]);
  int face;
  float m, u, w;
  if (ax >= ay && ax >= az) { face = v[0] < 0 ? 1 : 0; m = ax; u = v[1]; w = v[2]; }
  else if (ay >= az) { face = v[1] < 0 ? 3 : 2; m = ay; u = v[0]; w = v[2]; }
  else { face = v[2] < 0 ? 5 : 4; m = az; u = v[0]; w = v[1]; }
  const float inv = 1.0f / m;
  const int c = cell((double)(u * inv)), r = cell((double)(w * inv));
  return (uint32_t)(((size_t)face * R + r) * R + c);
}

bool occl_build_point_light(const float light[3], const OcclTri *tris, uint64_t n_tris, float *map) {
  const float far = 3.0e38f;
  for (size_t k = 0; k < CTR_OCCL_CELLS; k++) map[k] = far;
  const D3 L{light[0], light[1], light[2]};
  bool usable = std::isfinite(L.x) && std::isfinite(L.y) && std::isfinite(L.z);
  const double K_wide = 1.0 + 4.0 / R;   // a face's frustum widened by two cells
  const double K_safe = 1.0 - 8.0 / R;   // corners inside this: the triangle touches no other face's widened frustum
  for (uint64_t t = 0; t < n_tris && usable; t++) {
    D3 v[3];
    bool finite = true;
    for (int k = 0; k < 3; k++) {
      v[k] = {(double)tris[t].p[k][0] - L.x, (double)tris[t].p[k][1] - L.y, (double)tris[t].p[k][2] - L.z};
      finite = finite && std::isfinite(v[k].x) && std::isfinite(v[k].y) && std::isfinite(v[k].z);
    }
    if (!finite) { usable = false; break; }  // (a corner at infinity: no bound is claimed for this light)
    const double dist = tri_dist(v[0], v[1], v[2]);
    if (!(dist > 0.0)) { usable = false; break; }  // the light lies on a triangle
    // lowered by 2^-10 and rounded towards zero: never above the true distance of any point of the triangle
    float d = (float)(dist * (1.0 - 1.0 / 1024.0));
    if ((double)d > dist * (1.0 - 1.0 / 1024.0)) d = std::nextafterf(d, 0.0f);
    const int f0 = face_of(v[0]);
    bool fast = f0 == face_of(v[1]) && f0 == face_of(v[2]);
    if (fast) {
      const FaceAxes &F = FACES[f0];
      for (int k = 0; k < 3 && fast; k++) {
        const double m = F.sign * comp(v[k], F.major);
        fast = K_safe * m >= std::fabs(comp(v[k], F.col)) && K_safe * m >= std::fabs(comp(v[k], F.row));
      }
    }
    if (fast) {
      // all three corners well inside ONE face: the projection of the triangle is the triangle of the projected corners
      const FaceAxes &F = FACES[f0];
      double u0 = 2, u1 = -2, w0 = 2, w1 = -2;
      for (int k = 0; k < 3; k++) {
        const double m = F.sign * comp(v[k], F.major), u = comp(v[k], F.col) / m, w = comp(v[k], F.row) / m;
        u0 = std::min(u0, u); u1 = std::max(u1, u); w0 = std::min(w0, w); w1 = std::max(w1, w);
      }
      splat(map, f0, u0, u1, w0, w1, d);
      continue;
    }
    for (int f = 0; f < 6; f++) {
      const FaceAxes &F = FACES[f];
      D3 a[16], b[16];
      int n = 3;
      a[0] = v[0]; a[1] = v[1]; a[2] = v[2];
      n = clip(a, n, b, F, K_wide, F.col, +1.0); if (n == 0) continue;
      n = clip(b, n, a, F, K_wide, F.col, -1.0); if (n == 0) continue;
      n = clip(a, n, b, F, K_wide, F.row, +1.0); if (n == 0) continue;
      n = clip(b, n, a, F, K_wide, F.row, -1.0); if (n == 0) continue;
      double u0 = 2, u1 = -2, w0 = 2, w1 = -2;
      bool any = false;
      for (int k = 0; k < n; k++) {
        const double m = F.sign * comp(a[k], F.major);
        if (!(m > 0.0)) continue;  // (the light itself: dist > 0 excludes it from the triangle)
        const double u = comp(a[k], F.col) / m, w = comp(a[k], F.row) / m;
        u0 = std::min(u0, u); u1 = std::max(u1, u); w0 = std::min(w0, w); w1 = std::max(w1, w);
        any = true;
      }
      if (any) splat(map, f, u0, u1, w0, w1, d);
    }
  }
  if (!usable) memset(map, 0, sizeof(float) * CTR_OCCL_CELLS);
  return usable;
}
