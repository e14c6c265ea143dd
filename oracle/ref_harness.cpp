// ref_harness.cpp — builds the REFERENCE's own hot-path headers for the host.
// TEST INFRASTRUCTURE ONLY (see oracle/ctr_oracle.c header for the usage rule).
//
// Nothing from the reference is copied into this repository: the headers are
// compiled where they lie (-I$REF/inc) and the `cutrace::gpu::schema` block of
// inc/default_schema.hpp (lines 19-399; the rest of that header needs picojson,
// Assimp and the CUDA runtime, none of which exist in this image) is extracted
// at BUILD time into a temporary directory outside the repo that the Makefile
// deletes again.  Output: oracle/_ref/libcutrace_ref.so (git-ignored).
//
// Recipe: CUDA's function-space keywords become empty macros,
// and the unqualified min/max/isfinite/sqrt/pow/atan2/asin the device code
// calls are bound to the std:: float overloads (otherwise they would bind to
// the C double versions and differ from CUDA's float overloads).
//
// This TU restates only the 10-line body of render_kernel (inc/kernel.hpp:44-59)
// as a loop over pixels; ray_cast / ray_color / phong / shadow_intensity and all
// primitives are the reference's code, unmodified.
#define __host__
#define __device__
#define __global__
#define cudaCheck(x)

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

using std::asin;
using std::atan2;
using std::isfinite;
using std::pow;
using std::sqrt;
#ifdef CTR_REF_CUDA_MINMAX
// DIAGNOSTIC flavour (oracle/_ref/libcutrace_ref_cudaminmax.so): nvcc binds the unqualified min/max of device code
// (default_schema.hpp:109-110,241; shading.hpp:87,91) to its float overloads, which behave like fminf/fmaxf
// (a NaN operand is dropped), not like std::min/std::max (a < b selects).  Only used to COUNT how many pixels
// that difference touches (tests/test_oracle_golden.py::test_cuda_minmax_semantics_gap); never the parity target.
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
#else
using std::max;
using std::min;
#endif

#include "vector.hpp"
#include "gpu_array.hpp"
#include "gpu_variant.hpp"
#include "gpu_types.hpp"
#include "ray_cast.hpp"
#include "shading.hpp"
#include "default_schema_gpu_part.hpp"  // generated at build time, never committed

#include "../include/cutrace_amd.h"

namespace {
namespace gs = cutrace::gpu::schema;
using cutrace::vector;

// Counting sentinel: an extra object type that never intersects but counts how
// often ray_cast visits it (= number of ray_cast invocations).  Appended LAST in
// the variant's type list and LAST in the object array so the reference's object
// indices (hit_id) and variant tags 0-3 are unchanged.
thread_local uint64_t g_casts = 0;
thread_local uint64_t g_bbox_tris = 0;
struct counter_obj {
  size_t mat_idx;
  __device__ bool intersect(const cutrace::gpu::ray *, float, vector *, float *, vector *, cutrace::uv *) const {
    g_casts++;
    return false;
  }
  __host__ inline void gpu_clean() {}
};

using object_t = cutrace::gpu::gpu_object_set<gs::triangle, gs::mesh, gs::plane, gs::sphere, counter_obj>;
using light_t = cutrace::gpu::gpu_light_set<gs::sun, gs::point_light>;
using material_t = cutrace::gpu::gpu_material_set<gs::phong_material>;
using scene_t = cutrace::gpu::gpu_scene_<object_t, light_t, material_t, gs::cam>;

vector V(ctr_vec3 v) { return vector{v.x, v.y, v.z}; }

struct built_scene {
  std::vector<object_t> objects;
  std::vector<light_t> lights;
  std::vector<material_t> materials;
  std::vector<std::vector<gs::triangle>> mesh_tris;
  scene_t scene;
};

void build(const ctr_scene_desc *d, built_scene &b, bool with_counter) {
  b.mesh_tris.reserve(d->n_objects);
  for (uint64_t i = 0; i < d->n_objects; i++) {
    const ctr_object &o = d->objects[i];
    switch (o.type) {
      case CTR_OBJ_TRIANGLE:
        b.objects.emplace_back(gs::triangle{V(o.v0), V(o.v1), V(o.v2), (size_t)o.mat_idx});
        break;
      case CTR_OBJ_MESH: {
        b.mesh_tris.emplace_back();
        auto &tv = b.mesh_tris.back();
        tv.reserve(o.tri_count);
        for (uint64_t k = 0; k < o.tri_count; k++) {
          const ctr_triangle &t = d->triangles[o.tri_begin + k];
          tv.push_back(gs::triangle{V(t.p1), V(t.p2), V(t.p3), (size_t)o.mat_idx});
        }
        gs::mesh m{{tv.data(), tv.size()}, (size_t)o.mat_idx, cutrace::bound{V(o.v0), V(o.v1)}};
        b.objects.emplace_back(m);
        break;
      }
      case CTR_OBJ_PLANE:
        b.objects.emplace_back(gs::plane{V(o.v0), V(o.v1), (size_t)o.mat_idx});
        break;
      default:
        b.objects.emplace_back(gs::sphere{V(o.v0), o.f0, (size_t)o.mat_idx});
        break;
    }
  }
  if (with_counter) b.objects.emplace_back(counter_obj{0});
  for (uint64_t i = 0; i < d->n_lights; i++) {
    const ctr_light &l = d->lights[i];
    if (l.type == CTR_LIGHT_SUN) b.lights.emplace_back(gs::sun{V(l.v), V(l.color)});
    else b.lights.emplace_back(gs::point_light{V(l.v), V(l.color)});
  }
  for (uint64_t i = 0; i < d->n_materials; i++) {
    const ctr_material &m = d->materials[i];
    b.materials.emplace_back(gs::phong_material{V(m.color), m.specular, m.reflexivity, m.phong_exp, m.transparency});
  }
  const ctr_camera &c = d->cam;
  gs::cam cam{V(c.pos), V(c.up), V(c.forward), V(c.right), c.near_plane, c.far_plane, c.ambient, (size_t)c.w, (size_t)c.h};
  b.scene = scene_t{{b.objects.data(), b.objects.size()}, {b.lights.data(), b.lights.size()}, {b.materials.data(), b.materials.size()}, cam};
}

template <size_t B>
vector color_of(const scene_t *s, const cutrace::gpu::ray *r, float fudge) {
  return cutrace::gpu::ray_color<scene_t, B>(s, r, fudge, s->cam.get_ambient());
}

vector color_dispatch(const scene_t *s, const cutrace::gpu::ray *r, float fudge, int bounces) {
  switch (bounces) {
    case 0: return color_of<0>(s, r, fudge);
    case 1: return color_of<1>(s, r, fudge);
    case 2: return color_of<2>(s, r, fudge);
    case 3: return color_of<3>(s, r, fudge);
    case 4: return color_of<4>(s, r, fudge);
    case 5: return color_of<5>(s, r, fudge);
    case 6: return color_of<6>(s, r, fudge);
    case 7: return color_of<7>(s, r, fudge);
    case 8: return color_of<8>(s, r, fudge);
    case 9: return color_of<9>(s, r, fudge);
    default: return color_of<10>(s, r, fudge);
  }
}
}  // namespace

extern "C" {

// Same contract as orc_render (oracle/ctr_oracle.c).  counters[0] = ray_cast
// invocations; counters[1] is left 0 (only the restatement counts bytes).
// ... plus uv2 (optional): ray_cast's tex_coords of the primary cast (kernel.hpp:51-52), 2 floats per pixel
int ref_render_uv(const ctr_scene_desc *d, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2);
int ref_render(const ctr_scene_desc *d, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
               float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters) {
  return ref_render_uv(d, fudge, bounces, rows_in, n_threads, depth, color3, normal3, hit_ids, counters, nullptr);
}

int ref_render_ex(const ctr_scene_desc *d, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2, int ignore_transparent_primary);
int ref_render_uv(const ctr_scene_desc *d, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2) {
  return ref_render_ex(d, fudge, bounces, rows_in, n_threads, depth, color3, normal3, hit_ids, counters, uv2, 0);
}

// ... plus ignore_transparent_primary: the ninth argument of the reference's ray_cast (inc/ray_cast.hpp:30,39-40) for the cast
// of kernel.hpp:52; no caller of the reference passes true, this harness can
int ref_render_ex(const ctr_scene_desc *d, float fudge, int bounces, const ctr_rows *rows_in, int n_threads,
                  float *depth, float *color3, float *normal3, int64_t *hit_ids, uint64_t *counters, float *uv2, int ignore_transparent_primary) {
  built_scene b;
  build(d, b, counters != nullptr);
  const scene_t *scene = &b.scene;
  const size_t w = d->cam.w, h = d->cam.h;
  const size_t n_real_objects = d->n_objects;

  ctr_rows rr{0, h, h ? h : 1, 0, 1};
  if (rows_in && rows_in->row_end > rows_in->row_begin) {
    rr = *rows_in;
    if (rr.block_rows == 0) rr.block_rows = h ? h : 1;
    if (rr.n_parts == 0) { rr.n_parts = 1; rr.part = 0; }
    if (rr.row_end > h) rr.row_end = h;
  }
  std::vector<size_t> sel;
  for (size_t y = 0; y < h; y++)
    if (y >= rr.row_begin && y < rr.row_end && ((y / rr.block_rows) % rr.n_parts) == rr.part) sel.push_back(y);

  if (n_threads < 1) n_threads = 1;
  std::atomic<size_t> next{0};
  std::atomic<uint64_t> casts{0};
  auto work = [&]() {
    g_casts = 0;
    for (;;) {
      size_t k = next.fetch_add(1);
      if (k >= sel.size()) break;
      size_t y_id = sel[k];
      for (size_t x_id = 0; x_id < w; x_id++) {
        // ---- body of render_kernel, inc/kernel.hpp:47-59 ----
        float dist = INFINITY;
        cutrace::gpu::ray r = scene->cam.get_ray(x_id, y_id);
        size_t hit_id = scene->objects.size;
        vector hit_point{}, normal{0, 0, 0};
        cutrace::uv tc{};
        bool did_hit = cutrace::gpu::ray_cast(scene, &r, fudge, &dist, &hit_id, &hit_point, &normal, &tc, ignore_transparent_primary != 0);
        size_t px = k * w + x_id;
        depth[px] = dist;
        normal3[3 * px + 0] = normal.x; normal3[3 * px + 1] = normal.y; normal3[3 * px + 2] = normal.z;
        if (hit_ids) hit_ids[px] = (did_hit && hit_id < n_real_objects) ? (int64_t)hit_id : -1;
        if (uv2) { uv2[2 * px + 0] = tc.u; uv2[2 * px + 1] = tc.v; }
        vector c = color_dispatch(scene, &r, fudge, bounces);
        color3[3 * px + 0] = c.x; color3[3 * px + 1] = c.y; color3[3 * px + 2] = c.z;
      }
    }
    casts += g_casts;
  };
  if (n_threads == 1) work();
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(work);
    for (auto &t : th) t.join();
  }
  if (counters) { counters[0] = casts.load(); counters[1] = 0; }
  return 0;
}

// cam::look_at from the reference (default_schema.hpp:370-374), for pinning
// ctr_camera_look_at / orc_look_at.
void ref_look_at(ctr_camera *cam, ctr_vec3 eye, ctr_vec3 up_hint, ctr_vec3 look) {
  gs::cam c{V(eye), V(up_hint), {}, {}, 0.1f, 100.0f, 0.1f, 1, 1};
  c.look_at(V(look));
  cam->pos = {c.pos.x, c.pos.y, c.pos.z};
  cam->up = {c.up.x, c.up.y, c.up.z};
  cam->forward = {c.forward.x, c.forward.y, c.forward.z};
  cam->right = {c.right.x, c.right.y, c.right.z};
}

}  // extern "C"
