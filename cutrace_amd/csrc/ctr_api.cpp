// ctr_api.cpp — the C-ABI of include/cutrace_amd.h on top of the gfx950 kernel.
//
// Replaces the host half of the reference's hot path:
//   cpu_to_gpu::convert          inc/cpu_to_gpu.hpp:188-198  → ctr_scene_create (one flat upload,
//                                                               hipMalloc + hipMemcpy, no managed memory)
//   gpu::render<S,bounces,tpb>   inc/kernel.hpp:86-130       → ctr_render / ctr_render_device
//
// Compile with -ffp-contract=off: the ray-independent triangle quantities computed here
// (a, b, geometric normal) must have the reference's bits.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "bvh.h"
#include "cutrace_amd.h"
#include "scene_device.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
  g_err = msg;
  fprintf(stderr, "cutrace_amd: %s\n", msg.c_str());  // print-and-continue, like cudaCheck (inc/cuda.hpp:12-22)
  return code;
}
int hip_fail(hipError_t e, const char *what) {
  return fail(CTR_E_HIP_BASE + (int)e, std::string(what) + ": " + hipGetErrorName(e) + " (" + hipGetErrorString(e) + ")");
}
#define HIP_TRY(expr)                                     \
  do {                                                    \
    hipError_t _e = (expr);                               \
    if (_e != hipSuccess) return hip_fail(_e, #expr);     \
  } while (0)

constexpr float KAPPA = 1.0f / 16384.0f;  // prefilter slack factor 2^-14 (≈1000 ulp), see DESIGN.md

struct f3 { float x, y, z; };
inline f3 sub(ctr_vec3 a, ctr_vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 cross(f3 a, f3 o) { return {a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x}; }

#ifndef CTR_VMEM_THRESHOLD
#define CTR_VMEM_THRESHOLD (128 * 1024)
#endif
#ifndef CTR_BVH_LEAF
#define CTR_BVH_LEAF 4
#endif
#ifndef CTR_OCC6_MIN_TRIS
#define CTR_OCC6_MIN_TRIS 1000  // scenes with at least this many mesh triangles use the 6-waves-per-SIMD build
#endif
#ifndef CTR_FIRST_ORDER_MIN_TILES
#define CTR_FIRST_ORDER_MIN_TILES 8192  // smaller launches start in image order: one round of waves, no tail to shape
#endif
#ifndef CTR_ORDER_PERIOD
#define CTR_ORDER_PERIOD 8  // launches between rebuilds of the tile order
#endif
constexpr uint32_t BVH_LEAF = CTR_BVH_LEAF;  // triangles per BVH leaf

DCam to_dcam(const ctr_camera &c) {
  DCam cam{};
  cam.pos[0] = c.pos.x; cam.pos[1] = c.pos.y; cam.pos[2] = c.pos.z;
  cam.up[0] = c.up.x; cam.up[1] = c.up.y; cam.up[2] = c.up.z;
  cam.forward[0] = c.forward.x; cam.forward[1] = c.forward.y; cam.forward[2] = c.forward.z;
  cam.right[0] = c.right.x; cam.right[1] = c.right.y; cam.right[2] = c.right.z;
  cam.ambient = c.ambient;
  cam.w = (uint32_t)c.w;
  cam.h = (uint32_t)c.h;
  return cam;
}

void make_tri(const ctr_vec3 &p1, const ctr_vec3 &p2, const ctr_vec3 &p3, uint32_t orig, DTri &T, float *gn) {
  f3 a = sub(p2, p1), b = sub(p2, p3);  // default_schema.hpp:58
  T.ab[0][0] = a.x; T.ab[1][0] = a.y; T.ab[2][0] = a.z;
  T.ab[0][1] = b.x; T.ab[1][1] = b.y; T.ab[2][1] = b.z;
  T.px = p2.x; T.py = p2.y; T.pz = p2.z;
  f3 n = cross(a, b);
  T.nx = n.x; T.ny = n.y; T.nz = n.z;
  float emax = 0.f;
  for (float v : {a.x, a.y, a.z, b.x, b.y, b.z}) emax = fmaxf(emax, fabsf(v));
  T.ke = KAPPA * emax;
  T.ke2 = KAPPA * emax * emax;
  T.orig = orig;
  T.pad1 = 0.f;
  // default_schema.hpp:72: -1.0f * (p2 - p3).cross(p1 - p3).normalized()
  f3 c = cross(sub(p2, p3), sub(p1, p3));
  float nrm = sqrtf(c.x * c.x + c.y * c.y + c.z * c.z);
  float f = 1.0f / nrm;
  gn[0] = -1.0f * (f * c.x);
  gn[1] = -1.0f * (f * c.y);
  gn[2] = -1.0f * (f * c.z);
  gn[3] = 0.f;
}

}  // namespace

void ctr_internal_set_error(const char *msg) { g_err = msg ? msg : ""; }

struct ctr_scene {
  int device = 0;
  DObj *d_objs = nullptr;
  DObj *d_oloop = nullptr;
  DObj *d_meshes = nullptr;
  uint32_t n_mesh = 0, tlas_root = BVH_LEAF_FLAG, tlas_begin = 0;
  float tl_mn[3] = {0, 0, 0}, tl_mx[3] = {0, 0, 0};
  DPlanePair *d_planes = nullptr;
  uint32_t n_oloop = 0, n_plane_recs = 0, n_axis_recs = 0;
  DTri *d_tris = nullptr;
  DNode *d_nodes = nullptr;
  DNode4 *d_nodes4 = nullptr;
  float *d_gnorm = nullptr;
  DLight *d_lights = nullptr;
  DMat *d_mats = nullptr;
  uint32_t n_obj = 0, n_tri = 0, n_light = 0, n_mat = 0;
  bool has_mesh = false;
  bool all_opaque = true;
  bool need_cold = false;
  bool any_bounce = false;    // some material reflects or transmits (>= 1e-6): the recursion can go below depth 0
  size_t mesh_bytes = 0;      // triangles + BVH nodes
  uint64_t mesh_tris = 0;     // triangles in meshes
  static uint64_t occ6_min_tris() {
    static const uint64_t v = [] { const char *e = getenv("CUTRACE_OCC6_MIN_TRIS"); return e ? (uint64_t)atoll(e) : (uint64_t)CTR_OCC6_MIN_TRIS; }();
    return v;
  }
  DCam cam{};                 // camera 0 (image size of every camera)
  DCam *d_cams = nullptr;     // device camera array (>= 1 entry)
  uint32_t n_cams = 0;
  uint32_t user_variant = CTR_VAR_AUTO;
  // cached device outputs for the host-buffer form: ONE allocation, a call's buffers are its consecutive
  // parts [depth px | color 3 px | normal 3 px] so that a frame can leave in a single D2H transfer
  float *d_out = nullptr;
  float *d_uv = nullptr;      // ctr_render_uv: 2 floats per pixel, allocated on first use
  size_t uv_px = 0;
  // Guard of the BVH culling (see refresh_linear_meshes): host copies of what it needs
  struct MeshGuard {
    uint32_t node_begin = 0, node_count = 0;
    uint32_t tri_begin = 0, tri_count = 0;  // the mesh's DTri range (leaf order); CTR_GUARD_SLOTS spare records follow it
    uint32_t obj_index = 0;                 // position in d_objs
    int mesh_pos = -1;                      // position in d_meshes (-1: an empty mesh, never walked)
    std::vector<double> planes;  // per triangle (leaf order): unit normal (3), a point of the plane (3), extent
    std::vector<uint32_t> guarded;          // triangles currently copied into the spare records
    bool linear = false;         // its nodes currently carry unbounded boxes
  };
  std::vector<MeshGuard> guards;
  // ONE four-wide tree over the triangles of ALL meshes (scenes with 2..255 non-empty meshes; render_kernel.hip "merged
  // walk"): its records are appended to d_tris / d_nodes4, a pseudo mesh record at d_meshes[n_mesh] leads to them, and
  // d_meshes[n_mesh + 1 + r] is mesh r in SCENE order (a merged triangle's key names r in its upper 8 bits).
  struct Merged {
    bool reserved = false;   // the scene qualifies (2..255 non-empty meshes) and the device arrays have room for the tree
    bool built = false;      // the structures exist (build_merged_tree: at the first ctr_set_variant with CTR_VAR_MERGE)
    uint32_t node_cap = 0;   // room for the tree's nodes in d_nodes4 (the spare nodes follow)
    std::vector<ctr_triangle> src;  // the meshes' triangles, scene order then file order (kept for the build)
    bool usable = false;     // ... and may be walked (refresh_linear_meshes: no mesh went linear, the guard records fit)
    uint32_t tri_begin = 0, tri_count = 0, node_begin = 0, node_count = 0;
    std::vector<uint32_t> slot_of;  // per mesh rank: first index of its triangles in a (rank, file index) numbering (+ one past the last)
    std::vector<uint32_t> where;    // merged position of triangle (rank, file index) -> tri_begin-relative record index
    std::vector<uint32_t> guarded;  // keys currently in the guard records
  } merged;
  uint32_t min_merge_meshes() const {
    static const uint32_t v = [] { const char *e = getenv("CUTRACE_MERGE_MIN_MESHES"); return e ? (uint32_t)atol(e) : 2u; }();
    return v;
  }
  std::vector<DNode4> h_nodes4;  // the real boxes
  std::vector<DTri> h_tris;
  std::vector<float> h_gn;
  std::vector<DObj> h_objs, h_meshes;
  std::vector<DCam> h_cams;
  std::vector<DLight> h_lights;
  std::vector<DMat> h_mats;
  unsigned long long *h_counters = nullptr;  // pinned landing zone of the 16 counter words
  unsigned long long last_cnt[16] = {0};     // the counter words of the last host-form render
  unsigned long long *d_counters = nullptr;
  unsigned long long *d_shards = nullptr;  // CTR_SHARDS x CTR_SHARD_WORDS, zero between launches
  size_t out_px = 0;
  uint32_t *d_groups = nullptr;  // host delivery: one completion counter per group of tiles (render_kernel.hip)
  uint32_t *h_groups = nullptr;  // page-locked landing zone of the counters (checked after every direct launch)
  size_t groups_cap = 0;
  bool poison_next_order = false;  // test hook (ctr_debug_poison_next_order)
  // tile scheduling feedback (include/cutrace_amd.h "Tile scheduling")
  uint32_t *d_cost = nullptr, *d_order = nullptr;
  uint32_t order_age = 0;  // launches of the current shape
  uint64_t order_view = 0; // camera set + first frame of the previous launch
  uint32_t cams_epoch = 0; // bumped by ctr_scene_set_cameras / ctr_scene_set_size
  uint64_t order_cap = 0;
  uint64_t order_key[6] = {0, 0, 0, 0, 0, 0};
  bool order_valid = false;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  std::mutex mtx;

  uint32_t kernel_variant(bool count) const {
    uint32_t kv = 0;
    if (!(user_variant & CTR_VAR_NO_PREFILTER)) kv |= KV_PREFILTER;
    // shadow any-hit is result-identical only when every material is exactly opaque
    // (SURVEY §8(a) row a9); with any transparency the ordered nearest-hit loop is kept
    if (all_opaque && !(user_variant & CTR_VAR_NO_ANYHIT) && !count) kv |= KV_ANYHIT;
    if (!(user_variant & CTR_VAR_NO_CLUSTER) && !count) kv |= KV_BVH;
    if (!(user_variant & CTR_VAR_EXACT_POW)) kv |= KV_FASTPOW;
    // meshes of CTR_OCC6_MIN_TRIS triangles and more: the build for 6 waves per SIMD (render_kernel.hip KV_OCC6)
    if (mesh_tris >= occ6_min_tris() && !(user_variant & CTR_VAR_NO_OCC6)) kv |= KV_OCC6;
    if (user_variant & CTR_VAR_STATS) kv = KV_STATS | (all_opaque && !(user_variant & CTR_VAR_NO_ANYHIT) ? KV_ANYHIT : 0u);
    if (count) kv = KV_PREFILTER | KV_COUNT;  // the counting launch walks like the reference (and wins over STATS)
    return kv;
  }
};

namespace {

int make_rows(const ctr_scene *s, const ctr_rows *rin, DRows &R) {
  const uint64_t h = s->cam.h;
  ctr_rows r{0, h, h ? h : 1, 0, 1};
  if (rin && rin->row_end > rin->row_begin) {
    r = *rin;
    if (r.block_rows == 0) r.block_rows = h ? h : 1;
    if (r.n_parts == 0) { r.n_parts = 1; r.part = 0; }
    if (r.row_end > h) r.row_end = h;
  }
  if (r.part >= r.n_parts) return fail(CTR_E_INVALID, "ctr_rows: part >= n_parts");
  if (r.n_parts > 1 && (r.row_begin % r.block_rows) != 0)
    return fail(CTR_E_INVALID, "ctr_rows: row_begin must be a multiple of block_rows when n_parts > 1");
  R.row_begin = (uint32_t)r.row_begin;
  R.row_end = (uint32_t)r.row_end;
  R.part_stride = 0;
  R.block_rows = (uint32_t)r.block_rows;
  R.part = r.part;
  R.n_parts = r.n_parts;
  if (r.n_parts <= 1) {
    R.n_rows = (uint32_t)(r.row_end > r.row_begin ? r.row_end - r.row_begin : 0);
    R.first_block = 0;
    R.n_parts = 1;
    R.part = 0;
  } else {
    uint64_t b0 = r.row_begin / r.block_rows;
    uint64_t first = b0 + ((r.part + r.n_parts - (b0 % r.n_parts)) % r.n_parts);
    R.first_block = (uint32_t)first;
    uint64_t n = 0;
    for (uint64_t b = first; b * r.block_rows < r.row_end; b += r.n_parts) {
      uint64_t lo = b * r.block_rows, hi = lo + r.block_rows;
      if (hi > r.row_end) hi = r.row_end;
      n += hi - lo;
    }
    R.n_rows = (uint32_t)n;
  }
  return CTR_OK;
}

void fill_launch(const ctr_scene *s, RenderLaunch &L) {
  L.objs = s->d_objs;
  L.oloop = s->d_oloop;
  L.meshes = s->d_meshes;
  L.n_mesh = s->n_mesh;
  L.tlas_root = s->tlas_root;          // (use_merged_tree, once the launch's variant is known, may put the merged tree here)
  L.tlas_root_regular = s->tlas_root;
  L.tlas_begin = s->tlas_begin;
  for (int q = 0; q < 3; q++) { L.tl_mn[q] = s->tl_mn[q]; L.tl_mx[q] = s->tl_mx[q]; }
  L.planes = s->d_planes;
  L.n_oloop = s->n_oloop;
  L.n_plane_recs = s->n_plane_recs;
  L.n_axis_recs = s->n_axis_recs;
  L.tris = s->d_tris;
  L.nodes = s->d_nodes;
  L.nodes4 = s->d_nodes4;
  L.gnorm = s->d_gnorm;
  L.lights = s->d_lights;
  L.mats = s->d_mats;
  L.n_obj = s->n_obj;
  L.n_light = s->n_light;
  L.n_mat = s->n_mat;
  L.has_mesh = s->has_mesh ? 1u : 0u;
  L.need_cold_frames = s->need_cold ? 1u : 0u;
  L.any_bounce = s->any_bounce ? 1u : 0u;
  L.cams = s->d_cams;
  L.shards = s->d_shards;
  L.w = s->cam.w;
  L.h = s->cam.h;
  L.first_frame = 0;
  L.n_frames = 1;
  L.frame_stride_px = 0;
}

// The merged tree (CTR_VAR_MERGE) when the scene has one and nothing speaks against it (ctr_scene::Merged): a BVH walk of
// the shipped kind, frame leaving through device buffers.  The top-level tree over the meshes stays the fallback the kernel
// itself takes for a cast the merged walk cannot decide (render_kernel.hip "merged walk").
void use_merged_tree(const ctr_scene *s, RenderLaunch &L) {
  const bool bvh_walk = (L.variant & (KV_BVH | KV_STATS)) && !(L.variant & (KV_COUNT | KV_UV));
  if ((s->user_variant & CTR_VAR_MERGE) && s->merged.built && s->merged.usable && bvh_walk && !L.group_done &&
      (L.variant & KV_PREFILTER || (L.variant & KV_STATS))) {
    L.variant |= KV_MERGE;
    L.tlas_root = BVH_LEAF_FLAG | s->n_mesh;
  }
}

// Attach the tile-order buffers to a launch: use the stored order when the launch has the shape the
// order was measured on, and have the launch record costs + sort them for the next one.
int attach_order(ctr_scene *s, RenderLaunch &L, bool count) {
  L.order = nullptr;
  L.cost = nullptr;
  L.order_next = nullptr;
  if ((s->user_variant & (CTR_VAR_NO_REORDER | CTR_VAR_STATS)) || count) return CTR_OK;
  const uint64_t n = ctr_launch_waves(L);
  if (n == 0 || n > 0x7FFFFFFFull) return CTR_OK;
  if (n > s->order_cap) {
    if (s->d_cost) (void)hipFree(s->d_cost);
    if (s->d_order) (void)hipFree(s->d_order);
    s->d_cost = s->d_order = nullptr;
    s->order_cap = 0;
    s->order_valid = false;
    HIP_TRY(hipMalloc((void **)&s->d_cost, n * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&s->d_order, n * sizeof(uint32_t)));
    s->order_cap = n;
  }
  const uint64_t key[6] = {n, ((uint64_t)L.w << 32) | L.h, ((uint64_t)L.rows.row_begin << 32) | L.rows.row_end,
                           ((uint64_t)L.rows.block_rows << 32) | L.rows.n_parts,
                           ((uint64_t)L.rows.part << 32) | L.rows.part_stride,
                           ((uint64_t)(L.group_done ? 1u : 0u) << 32) | L.n_frames};  // (host delivery orders tiles by group)
  const bool same = s->order_valid && memcmp(key, s->order_key, sizeof(key)) == 0;
  // (a shape's first launch runs in image order: an a-priori estimate — tiles whose primary rays meet the box
  //  of a mesh / sphere, computed and sorted by a pre-pass — was built and measured in round 2: the expensive
  //  tiles of these scenes are speckle along shadow edges and reflections, not the tiles that look at an
  //  object, and the pre-pass cost more than it won; DESIGN.md "First launch")
  if (same) L.order = s->d_order;
  else {
    s->order_age = 0;
    // a shape nothing is known about: centre-out instead of image order (render_kernel.hip first_order) — for
    // launches large enough to have a tail and scenes heavy enough (the triangle count that also picks the 6-wave
    // build) for the ~8 us of the order kernel to pay: bunny -5.5 %, 64k bunny -2 %, C4 -2 %, but mirror.json (924
    // triangles, 0.2 ms) +4 % (profiles/r02/first_launch_centre_out.txt)
    if (n >= CTR_FIRST_ORDER_MIN_TILES && s->mesh_tris >= ctr_scene::occ6_min_tris() && !(s->user_variant & CTR_VAR_IMAGE_ORDER_FIRST)) {
      L.order = s->d_order;
      L.order_init = 1;
    }
  }
  memcpy(s->order_key, key, sizeof(key));
  s->order_valid = true;  // after this launch d_order holds an order measured on this shape
  L.cost = s->d_cost;
  // The order is rebuilt after the first two launches of a shape (the second one measured under the
  // new order); after that every launch while the view keeps changing (a camera path: 90-frame
  // orbit 1.46 ms/frame rebuilt every frame vs 1.59 every 8th, 1.77 without scheduling), and only
  // every CTR_ORDER_PERIOD-th launch while the same view is rendered again and again.
  const uint64_t view = ((uint64_t)s->cams_epoch << 32) | L.first_frame;
  const bool same_view = same && view == s->order_view;
  s->order_view = view;
  if (s->order_age < 2 || !same_view || s->order_age % CTR_ORDER_PERIOD == 0) L.order_next = s->d_order;
  s->order_age++;
  if (s->poison_next_order) {
    // test hook: this launch gets an order whose second half names no tile — those waves leave at once, their tiles
    // are never rendered, and (host delivery) their groups never complete
    s->poison_next_order = false;
    std::vector<uint32_t> o(n);
    for (uint64_t k = 0; k < n; k++) o[k] = k < n / 2 ? (uint32_t)k : 0xFFFFFFFFu;
    HIP_TRY(hipMemcpy(s->d_order, o.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    L.order = s->d_order;
    L.order_init = 0;
    L.order_next = nullptr;
    s->order_valid = false;  // the next launch starts over
  }
  return CTR_OK;
}

// The per-mesh BVH lets a cast skip triangles whose (widened) box its ray cannot touch.  That is the
// reference's result except in ONE regime: a ray that lies IN the plane of a triangle to within rounding
// has alpha = det[a b c] (default_schema.hpp:59) and all three numerators at noise level, and the
// reference's float test may then accept the triangle for a ray that passes far from it — a hit the
// culling would drop (tests/test_gpu_parity.py::test_rays_coplanar_with_triangles).  Such a ray has its
// ORIGIN in the triangle's plane and its direction parallel to it, both to ~1e-6 relative (a direction
// that leaves the plane by more makes t0 = noise/alpha < min_t): for primary rays the eye must lie in
// the plane, for shadow rays the light must (or a sun must be parallel to it).  So at upload, and
// whenever the cameras change, every triangle plane of every mesh is checked against the eyes, the point
// lights and the sun directions (tolerance 2^-17, an order of magnitude above the rounding that matters), and the
// triangles that qualify are copied into the mesh's GUARD records, which the walk tests for every lane
// that passes the mesh's own AABB test, whatever their box (a duplicate test cannot change the
// lexicographic minimum of (t, file index)).  More than CTR_GUARD_SLOTS of them: every node of the mesh
// gets unbounded boxes instead — the reference's linear walk through the same code.
// Secondary rays (round 3; tests/test_gpu_parity.py::test_secondary_rays_coplanar_with_triangles showed the hole is real:
// one pixel of a purpose-built scene differed).  A ray reflected by a PLANAR mirror lies on the line through the
// mirror image of its parent's origin, so the rays a flat mirror makes of the primary rays all pass through the
// mirror image of the eye — a VIRTUAL eye — and fall into a triangle's plane only if that point lies in it.  The
// same check therefore runs for the virtual eyes too: every eye mirrored in every reflective plane, stand-alone
// triangle and triangle of a small mesh (<= CTR_MIRROR_MESH_TRIS: mirrors built from a few triangles, like
// scene/mirror.stl), and those images mirrored once more (reflections of reflections) while the list stays short.
// Pass-through rays continue their parent's line and need no entry; shadow rays run from a surface to a light, whose
// position is checked already; a sphere keeps a pencil of rays planar only in a plane through its centre and the
// pencil's apex, which the apex's own entry covers.  NOT covered: chains of more than two reflections, mirrors that
// are large meshes, and single rays (not families) that meet a triangle's plane by numerical coincidence — per (ray,
// triangle) pair a ~1e-9 event that no full-size comparison or fuzz run has shown yet.
#define CTR_GUARD_SLOTS 64u
#define CTR_MERGED_SPARE_NODES ((CTR_GUARD_SLOTS + 2u) / 3u)  // the merged tree: three meshes' guard leaves per spare node
#define CTR_MIRROR_MESH_TRIS 16u
#define CTR_SECOND_ORDER_MAX_EYES 8u
#define CTR_VIRTUAL_EYES_MAX 4096u  // (round 3: 96 — a 90-camera path through a room of five reflecting walls got images for its first 16 cameras only)
int refresh_linear_meshes(ctr_scene *s) {
  constexpr double TOL = 1.0 / 131072.0;
  // ---- the points a family of rays can emanate from: eyes, and their images in the scene's flat mirrors ----
  struct P3 { double x, y, z; };
  struct Mirror { P3 p, n; };  // a point of the plane, its unit normal
  std::vector<Mirror> mirrors;
  auto add_mirror = [&](double px, double py, double pz, double nx, double ny, double nz) {
    const double len = sqrt(nx * nx + ny * ny + nz * nz);
    if (!(len > 0.0)) return;
    nx /= len; ny /= len; nz /= len;
    const double c = px * nx + py * ny + pz * nz;
    for (const Mirror &m : mirrors) {  // one entry per plane (a mirror made of coplanar triangles)
      const double dot = m.n.x * nx + m.n.y * ny + m.n.z * nz, cm = m.p.x * m.n.x + m.p.y * m.n.y + m.p.z * m.n.z;
      if ((fabs(dot - 1.0) < 1e-9 && fabs(cm - c) < 1e-9 * (1.0 + fabs(c))) || (fabs(dot + 1.0) < 1e-9 && fabs(cm + c) < 1e-9 * (1.0 + fabs(c)))) return;
    }
    mirrors.push_back({{px, py, pz}, {nx, ny, nz}});
  };
  auto tri_plane = [&](const DTri &T) {
    const double ax = T.ab[0][0], ay = T.ab[1][0], az = T.ab[2][0], bx = T.ab[0][1], by = T.ab[1][1], bz = T.ab[2][1];
    add_mirror(T.px, T.py, T.pz, ay * bz - az * by, az * bx - ax * bz, ax * by - ay * bx);
  };
  for (const DObj &O : s->h_objs) {
    if (O.mat >= s->h_mats.size() || !((double)s->h_mats[O.mat].reflexivity >= 1e-6)) continue;
    if (O.type == CTR_OBJ_PLANE) add_mirror(O.f[0], O.f[1], O.f[2], O.f[3], O.f[4], O.f[5]);
    else if (O.type == CTR_OBJ_TRIANGLE) tri_plane(s->h_tris[O.tri_begin]);
    else if (O.type == CTR_OBJ_MESH && O.tri_count <= CTR_MIRROR_MESH_TRIS)
      for (uint32_t k = 0; k < O.tri_count; k++) tri_plane(s->h_tris[O.tri_begin + k]);
  }
  std::vector<P3> origins;
  for (const DCam &c : s->h_cams) origins.push_back({c.pos[0], c.pos[1], c.pos[2]});
  {
    auto image = [](const P3 &e, const Mirror &m) {
      const double d = (e.x - m.p.x) * m.n.x + (e.y - m.p.y) * m.n.y + (e.z - m.p.z) * m.n.z;
      return P3{e.x - 2.0 * d * m.n.x, e.y - 2.0 * d * m.n.y, e.z - 2.0 * d * m.n.z};
    };
    const size_t n_eyes = origins.size();
    std::vector<std::pair<P3, size_t>> first;  // image, the mirror that made it
    for (size_t e = 0; e < n_eyes && first.size() < CTR_VIRTUAL_EYES_MAX; e++)
      for (size_t m = 0; m < mirrors.size() && first.size() < CTR_VIRTUAL_EYES_MAX; m++) first.push_back({image(origins[e], mirrors[m]), m});
    for (const auto &f : first) origins.push_back(f.first);
    if (n_eyes * mirrors.size() > first.size()) {
      static bool warned = false;
      if (!warned) fprintf(stderr, "cutrace_amd: %zu cameras x %zu flat mirrors exceed %u mirror images: the in-plane guard of reflected rays "
                                   "(DESIGN.md section 2) covers the first %zu only\n", n_eyes, mirrors.size(), CTR_VIRTUAL_EYES_MAX, first.size());
      warned = true;
    }
    // Images of images (two reflections in a row): for up to CTR_SECOND_ORDER_MAX_EYES cameras.  The guard records serve every
    // launch on the handle, whichever of its cameras the launch renders, so each camera's images cost every frame: with all
    // second-order images of a 90-camera path (2 340 points) the bunny room's frames ran 14 % slower for a handful of guard
    // triangles (bench.py config.campath_ms 1.17 -> 1.34 ms); first-order images of every camera stay.
    if (n_eyes <= CTR_SECOND_ORDER_MAX_EYES &&
        first.size() * (mirrors.size() ? mirrors.size() - 1 : 0) + origins.size() <= CTR_VIRTUAL_EYES_MAX)
      for (const auto &f : first)
        for (size_t m = 0; m < mirrors.size(); m++)
          if (m != f.second) origins.push_back(image(f.first, mirrors[m]));
  }
  std::vector<uint32_t> m_keys;  // merged tree: the keys (mesh rank << 24 | file index) of every mesh's risky triangles
  bool m_any_linear = false;
  uint32_t rank = 0;             // of the current mesh among the non-empty meshes, scene order (guards are in scene order)
  for (ctr_scene::MeshGuard &g : s->guards) {
    if (g.mesh_pos < 0) continue;
    const uint32_t g_rank = rank++;
    std::vector<uint32_t> risky;
    const size_t nt = g.planes.size() / 7;
    for (size_t t = 0; t < nt; t++) {
      const double *q = &g.planes[7 * t];
      if (q[0] == 0.0 && q[1] == 0.0 && q[2] == 0.0) continue;  // zero-area triangle: alpha is exactly 0, never a hit
      auto point_in_plane = [&](double x, double y, double z) {
        const double dx = x - q[3], dy = y - q[4], dz = z - q[5];
        const double dist = fabs(dx * q[0] + dy * q[1] + dz * q[2]);
        const double scale = fmax(fmax(fabs(dx), fabs(dy)), fmax(fabs(dz), q[6]));
        return dist <= TOL * scale;
      };
      bool hit = false;
      for (const P3 &o : origins)
        if (point_in_plane(o.x, o.y, o.z)) hit = true;
      for (const DLight &l : s->h_lights) {
        if (l.type == CTR_LIGHT_POINT) {
          if (point_in_plane(l.vx, l.vy, l.vz)) hit = true;
        } else {
          const double len = sqrt((double)l.vx * l.vx + (double)l.vy * l.vy + (double)l.vz * l.vz);
          if (len > 0.0 && fabs(l.vx * q[0] + l.vy * q[1] + l.vz * q[2]) <= TOL * len) hit = true;
        }
      }
      if (hit) risky.push_back((uint32_t)t);
    }
    const bool want_linear = risky.size() > CTR_GUARD_SLOTS;
    if (want_linear) m_any_linear = true;
    else
      for (uint32_t t : risky) m_keys.push_back((g_rank << 24) | s->h_tris[g.tri_begin + t].orig);
    if (want_linear) risky.clear();
    if (want_linear != g.linear && g.node_count) {
      std::vector<DNode4> nn(s->h_nodes4.begin() + g.node_begin, s->h_nodes4.begin() + g.node_begin + g.node_count);
      if (want_linear)
        for (DNode4 &n : nn)
          for (int c = 0; c < 4; c++)
            if (n.child[c] != BVH_LEAF_FLAG)  // (an unused slot stays what it is)
              for (int a = 0; a < 3; a++) { n.lo[a][c] = -3.0e38f; n.hi[a][c] = 3.0e38f; }
      HIP_TRY(hipMemcpy(s->d_nodes4 + g.node_begin, nn.data(), nn.size() * sizeof(DNode4), hipMemcpyHostToDevice));
      g.linear = want_linear;
    }
    if (risky != g.guarded) {
      const uint32_t slot0 = g.tri_begin + g.tri_count;
      for (size_t k = 0; k < risky.size(); k++) {
        s->h_tris[slot0 + k] = s->h_tris[g.tri_begin + risky[k]];
        for (int q = 0; q < 4; q++) s->h_gn[4 * (slot0 + k) + q] = s->h_gn[4 * (g.tri_begin + risky[k]) + q];
      }
      if (!risky.empty()) {
        HIP_TRY(hipMemcpy(s->d_tris + slot0, &s->h_tris[slot0], risky.size() * sizeof(DTri), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(s->d_gnorm + 4 * (size_t)slot0, &s->h_gn[4 * (size_t)slot0], risky.size() * 4 * sizeof(float), hipMemcpyHostToDevice));
      }
      // the walk starts at node `bvh_root`: the root (0), or — with guard records — the mesh's extra node, whose
      // children are the guard leaf (relative to the mesh's first triangle) and the root, both with unbounded boxes
      const uint32_t start = risky.empty() ? 0u : g.node_count;
      if (!risky.empty()) {
        DNode4 gn4;
        memset(&gn4, 0, sizeof(gn4));
        for (int c = 0; c < 4; c++) {
          const bool used = c < 2;
          for (int a = 0; a < 3; a++) { gn4.lo[a][c] = used ? -3.0e38f : 3.4028235e38f; gn4.hi[a][c] = used ? 3.0e38f : 3.4028235e38f; }
          gn4.child[c] = BVH_LEAF_FLAG;
        }
        gn4.child[0] = BVH_LEAF_FLAG | ((uint32_t)risky.size() << 24) | g.tri_count;
        gn4.child[1] = 0u;  // the root
        HIP_TRY(hipMemcpy(s->d_nodes4 + g.node_begin + g.node_count, &gn4, sizeof(DNode4), hipMemcpyHostToDevice));
      }
      s->h_meshes[g.mesh_pos].bvh_root = start;
      s->h_objs[g.obj_index].bvh_root = start;
      HIP_TRY(hipMemcpy(s->d_meshes + g.mesh_pos, &s->h_meshes[g.mesh_pos], sizeof(DObj), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(s->d_objs + g.obj_index, &s->h_objs[g.obj_index], sizeof(DObj), hipMemcpyHostToDevice));
      g.guarded = risky;
    }
  }
  // ---- the merged tree: the same guard, one set of spare records for the triangles of all meshes; a mesh that went
  //      linear or more risky triangles than records -> the merged tree is not walked (fill_launch) ----
  if (getenv("CUTRACE_DEBUG_GUARDS")) {
    size_t per_mesh = 0;
    for (const ctr_scene::MeshGuard &g : s->guards) per_mesh += g.guarded.size();
    fprintf(stderr, "cutrace_amd guards: %zu origins (eyes + mirror images), %zu mirrors, %zu guard records over %zu meshes, merged keys %zu, linear %d\n",
            origins.size(), mirrors.size(), per_mesh, s->guards.size(), m_keys.size(), (int)m_any_linear);
  }
  if (s->merged.built) {
    ctr_scene::Merged &M = s->merged;
    M.usable = !m_any_linear && m_keys.size() <= CTR_GUARD_SLOTS;
    if (M.usable && m_keys != M.guarded) {
      const uint32_t slot0 = M.tri_begin + M.tri_count;
      for (size_t k = 0; k < m_keys.size(); k++)
        s->h_tris[slot0 + k] = s->h_tris[M.tri_begin + M.where[M.slot_of[m_keys[k] >> 24] + (m_keys[k] & 0xFFFFFFu)]];
      if (!m_keys.empty()) {
        HIP_TRY(hipMemcpy(s->d_tris + slot0, &s->h_tris[slot0], m_keys.size() * sizeof(DTri), hipMemcpyHostToDevice));
        // The walk starts at a chain of spare nodes before the root.  A spare node holds the guard leaves of up to three
        // meshes, each behind the box of ITS MESH — the reference shows a mesh's triangles only to rays that pass that
        // box (default_schema.hpp:126), so the box is exactly as far as a guard record has to reach (behind an unbounded
        // box every cast of the frame would test every guard record of every mesh: a handful of them doubled the 16-mesh
        // frame's triangle tests) — and, as its fourth child, the next spare node or the root, unbounded.
        struct Group { uint32_t rank, first, count; };
        std::vector<Group> groups;
        for (size_t k = 0; k < m_keys.size(); k++) {
          const uint32_t r = m_keys[k] >> 24;
          if (groups.empty() || groups.back().rank != r) groups.push_back({r, (uint32_t)k, 0u});
          groups.back().count++;
        }
        const uint32_t n_spare = (uint32_t)((groups.size() + 2) / 3);
        std::vector<DNode4> chain(n_spare);
        for (uint32_t j = 0; j < n_spare; j++) {
          DNode4 &g4 = chain[j];
          memset(&g4, 0, sizeof(g4));
          for (int c = 0; c < 4; c++) {
            for (int a = 0; a < 3; a++) { g4.lo[a][c] = 3.4028235e38f; g4.hi[a][c] = 3.4028235e38f; }
            g4.child[c] = BVH_LEAF_FLAG;
          }
          // (slot 0: the next spare node or the root; slots 1..3: guard leaves — unused slots last, the walk skips an empty second pair)
          for (int a = 0; a < 3; a++) { g4.lo[a][0] = -3.0e38f; g4.hi[a][0] = 3.0e38f; }
          g4.child[0] = (j + 1 < n_spare) ? (M.node_count + j + 1) : 0u;
          for (int c = 0; c < 3; c++) {
            const size_t gi = (size_t)3 * j + c;
            if (gi >= groups.size()) break;
            const DObj &Rm = s->h_meshes[s->n_mesh + 1u + groups[gi].rank];
            for (int a = 0; a < 3; a++) { g4.lo[a][1 + c] = Rm.f[a]; g4.hi[a][1 + c] = Rm.f[3 + a]; }
            g4.child[1 + c] = BVH_LEAF_FLAG | (groups[gi].count << 24) | (M.tri_count + groups[gi].first);
          }
        }
        HIP_TRY(hipMemcpy(s->d_nodes4 + M.node_begin + M.node_count, chain.data(), chain.size() * sizeof(DNode4), hipMemcpyHostToDevice));
      }
      DObj &P = s->h_meshes[s->n_mesh];
      P.bvh_root = m_keys.empty() ? 0u : M.node_count;
      HIP_TRY(hipMemcpy(s->d_meshes + s->n_mesh, &P, sizeof(DObj), hipMemcpyHostToDevice));
      M.guarded = m_keys;
    }
  }
  return CTR_OK;
}

// The merged tree of a scene that has room for it (ctr_scene::Merged, ctr_scene_create): built, uploaded, guarded.
int build_merged_tree(ctr_scene *s) {
  ctr_scene::Merged &M = s->merged;
  if (!M.reserved || M.built) return CTR_OK;
  const uint32_t total = M.tri_count, n_rank = (uint32_t)M.slot_of.size() - 1;
  std::vector<BvhInput> prims(total);
  std::vector<uint32_t> rank_of(total);
  for (uint32_t r = 0; r < n_rank; r++)
    for (uint32_t k = M.slot_of[r]; k < M.slot_of[r + 1]; k++) {
      const ctr_vec3 *v[3] = {&M.src[k].p1, &M.src[k].p2, &M.src[k].p3};
      BvhInput &b = prims[k];
      for (int a = 0; a < 3; a++) {
        const float c0 = (&v[0]->x)[a], c1 = (&v[1]->x)[a], c2 = (&v[2]->x)[a];
        b.mn[a] = fminf(c0, fminf(c1, c2));
        b.mx[a] = fmaxf(c0, fmaxf(c1, c2));
        b.c[a] = 0.5f * (b.mn[a] + b.mx[a]);
      }
      rank_of[k] = r;
    }
  std::vector<DNode4> mnodes;
  std::vector<uint32_t> order;
  bvh4_build(prims, BVH_LEAF, mnodes, order);
  if (mnodes.size() > M.node_cap) return CTR_OK;  // (cannot happen for leaves of up to four triangles; then simply never used)
  M.node_count = (uint32_t)mnodes.size();
  std::copy(mnodes.begin(), mnodes.end(), s->h_nodes4.begin() + M.node_begin);
  M.where.assign(total, 0u);
  float dummy_gn[4];
  for (uint32_t k = 0; k < total; k++) {
    const uint32_t g = order[k], r = rank_of[g], f = g - M.slot_of[r];
    const ctr_triangle &t = M.src[g];
    make_tri(t.p1, t.p2, t.p3, (r << 24) | f, s->h_tris[M.tri_begin + k], dummy_gn);
    M.where[g] = k;
  }
  for (uint32_t k = 0; k < CTR_GUARD_SLOTS; k++) s->h_tris[M.tri_begin + total + k] = s->h_tris[M.tri_begin];
  DObj &P = s->h_meshes[s->n_mesh];
  P.node_count = M.node_count;
  P.bvh_root = 0;
  HIP_TRY(hipMemcpy(s->d_tris + M.tri_begin, &s->h_tris[M.tri_begin], ((size_t)total + CTR_GUARD_SLOTS) * sizeof(DTri), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_nodes4 + M.node_begin, &s->h_nodes4[M.node_begin], (size_t)M.node_count * sizeof(DNode4), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(s->d_meshes + s->n_mesh, &P, sizeof(DObj), hipMemcpyHostToDevice));
  M.built = true;
  M.guarded.assign(1, 0xFFFFFFFFu);  // (no key: the first refresh writes the guard state, empty or not)
  return refresh_linear_meshes(s);
}

int check_args(const ctr_scene *s, int bounces) {
  if (!s) return fail(CTR_E_INVALID, "null scene");
  if (bounces < 0 || bounces > CTR_MAX_BOUNCES)
    return fail(CTR_E_INVALID, "bounces must be in [0," + std::to_string(CTR_MAX_BOUNCES) + "]");
  if (s->cam.w == 0 || s->cam.h == 0) return fail(CTR_E_INVALID, "camera has zero width or height");
  return CTR_OK;
}

int ensure_outputs(ctr_scene *s, size_t px) {
  if (!s->h_counters) HIP_TRY(hipHostMalloc((void **)&s->h_counters, 16 * sizeof(unsigned long long), hipHostMallocDefault));
  if (px <= s->out_px && s->d_out) return CTR_OK;
  if (s->d_out) (void)hipFree(s->d_out);
  s->d_out = nullptr;
  s->out_px = 0;
  HIP_TRY(hipMalloc((void **)&s->d_out, sizeof(float) * 7 * px));
  s->out_px = px;
  return CTR_OK;
}

// is `p` page-locked host memory (hipHostMalloc / hipHostRegister)?  Then a D2H copy is one direct DMA.
bool is_pinned(const void *p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'ed memory: "invalid value", not an error of ours
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

// the device's address of page-locked host memory (false: not mapped for this device)
bool device_view(float *host, float **dev) {
  void *d = nullptr;
  if (hipHostGetDevicePointer(&d, host, 0) != hipSuccess || !d) {
    (void)hipGetLastError();
    return false;
  }
  *dev = (float *)d;
  return true;
}

}  // namespace

extern "C" {

int ctr_abi_version(void) { return CTR_ABI_VERSION; }
const char *ctr_last_error(void) { return g_err.c_str(); }

int ctr_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) return -hip_fail(e, "hipGetDeviceCount");
  return n;
}

int ctr_scene_create(const ctr_scene_desc *d, int device, ctr_scene **out) {
  if (!d || !out) return fail(CTR_E_INVALID, "ctr_scene_create: null argument");
  *out = nullptr;
  // ---- validate the description: the kernel trusts these indices ----
  if (d->n_objects > 0xFFFFFFFFull || d->n_triangles > 0x7FFFFFFFull || d->cam.w > 0x7FFFFFFFull ||
      d->cam.h > 0x7FFFFFFFull)
    return fail(CTR_E_INVALID, "scene too large for 32-bit device indices");
  if (d->cam.w * d->cam.h > 0x7FFFFFFFull * 2) return fail(CTR_E_INVALID, "image too large");
  for (uint64_t i = 0; i < d->n_objects; i++) {
    const ctr_object &o = d->objects[i];
    if (o.type > CTR_OBJ_SPHERE) return fail(CTR_E_INVALID, "object #" + std::to_string(i) + ": bad type");
    if (o.mat_idx >= d->n_materials)
      return fail(CTR_E_INVALID, "object #" + std::to_string(i) + ": material index out of range");
    if (o.type == CTR_OBJ_MESH && (o.tri_begin > d->n_triangles || o.tri_count > d->n_triangles - o.tri_begin))
      return fail(CTR_E_INVALID, "object #" + std::to_string(i) + ": triangle range out of bounds");
  }
  for (uint64_t i = 0; i < d->n_lights; i++)
    if (d->lights[i].type > CTR_LIGHT_POINT) return fail(CTR_E_INVALID, "light #" + std::to_string(i) + ": bad type");
  for (uint64_t i = 0; i < d->n_materials; i++)
    if (d->materials[i].type != CTR_MAT_PHONG)
      return fail(CTR_E_INVALID, "material #" + std::to_string(i) + ": bad type");

  // CUTRACE_DEBUG_CREATE=1: where the call spends its time (stderr)
  const bool dbg_t = getenv("CUTRACE_DEBUG_CREATE") != nullptr;
  auto dbg_t0 = std::chrono::high_resolution_clock::now();
  auto dbg_stamp = [&](const char *what) {
    if (!dbg_t) return;
    const auto now = std::chrono::high_resolution_clock::now();
    fprintf(stderr, "cutrace_amd scene_create: %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - dbg_t0).count());
    dbg_t0 = now;
  };
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0) return fail(CTR_E_NO_DEVICE, "no HIP device available (and there is no CPU fallback)");
  if (device < 0 || device >= ndev) return fail(CTR_E_INVALID, "device index out of range");
  HIP_TRY(hipSetDevice(device));

  dbg_stamp("device check");
  // ---- flatten: per mesh a BVH + its triangles in leaf order (each keeps its file index for
  //      tie-breaks), stand-alone triangles appended ----
  std::vector<DObj> objs(d->n_objects);
  std::vector<DTri> tris;
  std::vector<DNode> nodes;    // top-level tree
  std::vector<DNode4> nodes4;  // per-mesh trees
  std::vector<float> gn;
  std::vector<ctr_scene::MeshGuard> guards;
  tris.reserve(d->n_triangles + d->n_objects);
  bool has_mesh = false;
  // the meshes' trees first, several meshes at a time (a scene of 16 meshes of 1000 triangles: 2.5 -> 0.5 ms of this call; a mesh
  // large enough to matter is built on several threads by the builder itself)
  struct MeshTree { std::vector<DNode4> nodes; std::vector<uint32_t> order; };
  std::vector<MeshTree> trees(d->n_objects);
  {
    std::vector<uint64_t> mesh_ids;
    for (uint64_t i = 0; i < d->n_objects; i++)
      if (d->objects[i].type == CTR_OBJ_MESH && d->objects[i].tri_count <= 0xFFFFFFull) mesh_ids.push_back(i);
    std::atomic<size_t> next{0};
    auto work = [&] {
      for (size_t k; (k = next.fetch_add(1)) < mesh_ids.size();) {
        const ctr_object &o = d->objects[mesh_ids[k]];
        const ctr_triangle *src = d->triangles + o.tri_begin;
        const uint32_t n = (uint32_t)o.tri_count;
        std::vector<BvhInput> prims(n);
        for (uint32_t q = 0; q < n; q++) {
          const ctr_vec3 *v[3] = {&src[q].p1, &src[q].p2, &src[q].p3};
          BvhInput &b = prims[q];
          for (int a = 0; a < 3; a++) {
            const float c0 = (&v[0]->x)[a], c1 = (&v[1]->x)[a], c2 = (&v[2]->x)[a];
            b.mn[a] = fminf(c0, fminf(c1, c2));
            b.mx[a] = fmaxf(c0, fmaxf(c1, c2));
            b.c[a] = 0.5f * (b.mn[a] + b.mx[a]);
          }
        }
        bvh4_build(prims, BVH_LEAF, trees[mesh_ids[k]].nodes, trees[mesh_ids[k]].order);
      }
    };
    const size_t n_thr = std::min<size_t>(mesh_ids.size() > 1 ? mesh_ids.size() : 1, std::max(1u, std::min(8u, std::thread::hardware_concurrency())));
    std::vector<std::thread> pool;
    for (size_t t = 1; t < n_thr; t++) pool.emplace_back(work);
    work();
    for (std::thread &t : pool) t.join();
  }
  for (uint64_t i = 0; i < d->n_objects; i++) {
    const ctr_object &o = d->objects[i];
    DObj &O = objs[i];
    memset(&O, 0, sizeof(O));
    O.type = o.type;
    O.mat = (uint32_t)o.mat_idx;
    O.index = (uint32_t)i;
    {  // material::is_transparent (default_schema.hpp:334: `transparency >= 1e-6`, a double comparison): ray_cast's ignore_transparent
      const uint32_t tr = ((double)d->materials[o.mat_idx].transparency >= 1e-6) ? 1u : 0u;
      memcpy(&O.f[7], &tr, sizeof(tr));
    }
    switch (o.type) {
      case CTR_OBJ_TRIANGLE: {
        O.tri_begin = (uint32_t)tris.size();
        O.tri_count = 1;
        tris.emplace_back();
        gn.resize(4 * tris.size());
        make_tri(o.v0, o.v1, o.v2, 0, tris.back(), &gn[4 * (tris.size() - 1)]);
        O.f[0] = o.v0.x; O.f[1] = o.v0.y; O.f[2] = o.v0.z;  // p1, p3: triangle::uv_for (KV_UV)
        O.f[3] = o.v2.x; O.f[4] = o.v2.y; O.f[5] = o.v2.z;
        break;
      }
      case CTR_OBJ_MESH: {
        has_mesh = true;
        const ctr_triangle *src = d->triangles + o.tri_begin;
        const uint32_t n = (uint32_t)o.tri_count;
        if (o.tri_count > 0xFFFFFFull) return fail(CTR_E_INVALID, "object #" + std::to_string(i) + ": mesh has more than 2^24 triangles");
        std::vector<DNode4> &mnodes = trees[i].nodes;
        std::vector<uint32_t> &order = trees[i].order;
        O.tri_begin = (uint32_t)tris.size();
        O.tri_count = n;
        O.node_begin = (uint32_t)nodes4.size();
        O.node_count = (uint32_t)mnodes.size();
        O.bvh_root = 0;  // node 0 of the mesh; child descriptors stay relative to the mesh's first node / first triangle
        nodes4.insert(nodes4.end(), mnodes.begin(), mnodes.end());
        nodes4.emplace_back();  // the mesh's spare node (guard records, refresh_linear_meshes); unused = all zero
        memset(&nodes4.back(), 0, sizeof(DNode4));
        {
          ctr_scene::MeshGuard g;
          g.node_begin = O.node_begin;
          g.node_count = O.node_count;
          g.tri_begin = O.tri_begin;
          g.tri_count = n;
          g.obj_index = (uint32_t)i;
          g.planes.resize(7 * (size_t)n);
          for (uint32_t k = 0; k < n; k++) {
            const ctr_triangle &t = src[order[k]];  // leaf order, like the DTri records
            const double ax = (double)t.p2.x - t.p1.x, ay = (double)t.p2.y - t.p1.y, az = (double)t.p2.z - t.p1.z;
            const double bx = (double)t.p2.x - t.p3.x, by = (double)t.p2.y - t.p3.y, bz = (double)t.p2.z - t.p3.z;
            double nx = ay * bz - az * by, ny = az * bx - ax * bz, nz = ax * by - ay * bx;
            const double len = sqrt(nx * nx + ny * ny + nz * nz);
            double *q = &g.planes[7 * (size_t)k];
            if (len > 0.0) { q[0] = nx / len; q[1] = ny / len; q[2] = nz / len; } else { q[0] = q[1] = q[2] = 0.0; }
            q[3] = t.p2.x; q[4] = t.p2.y; q[5] = t.p2.z;
            q[6] = fmax(fmax(fmax(fabs(ax), fabs(ay)), fabs(az)), fmax(fmax(fabs(bx), fabs(by)), fabs(bz)));
          }
          guards.push_back(std::move(g));
        }
        tris.resize(tris.size() + n + (n ? CTR_GUARD_SLOTS : 0u));  // + the mesh's guard records (refresh_linear_meshes)
        gn.resize(4 * tris.size());
        for (uint32_t k = 0; k < n; k++) {
          const ctr_triangle &t = src[order[k]];
          make_tri(t.p1, t.p2, t.p3, order[k], tris[O.tri_begin + k], &gn[4 * (O.tri_begin + order[k])]);  // normals in FILE order: the kernel keeps only the winner's original index
        }
        for (uint32_t k = 0; n && k < CTR_GUARD_SLOTS; k++) {  // unused guard records: copies of the first triangle
          tris[O.tri_begin + n + k] = tris[O.tri_begin];
          for (int q = 0; q < 4; q++) gn[4 * (O.tri_begin + n + k) + q] = gn[4 * O.tri_begin + q];
        }
        O.f[0] = o.v0.x; O.f[1] = o.v0.y; O.f[2] = o.v0.z;
        O.f[3] = o.v1.x; O.f[4] = o.v1.y; O.f[5] = o.v1.z;
        break;
      }
      case CTR_OBJ_PLANE:
        O.f[0] = o.v0.x; O.f[1] = o.v0.y; O.f[2] = o.v0.z;
        O.f[3] = o.v1.x; O.f[4] = o.v1.y; O.f[5] = o.v1.z;
        break;
      default:
        O.f[0] = o.v0.x; O.f[1] = o.v0.y; O.f[2] = o.v0.z;
        O.f[3] = o.f0;
        O.f[4] = o.f0 * o.f0;
        break;
    }
  }
  dbg_stamp("meshes: BVH + records");
  std::vector<DObj> oloop, meshes_in, meshes;
  uint32_t n_axis_recs = 0;
  std::vector<DPlanePair> planes;
  std::vector<const DObj *> axis_planes[3], general_planes;
  for (const DObj &O : objs) {
    if (O.type == CTR_OBJ_PLANE) {
      // axis-aligned: exactly one normal component is not zero (+0 and -0 both count as zero), everything finite
      int nz = 0, axis = -1;
      bool finite = true;
      for (int a = 0; a < 3; a++) {
        if (!std::isfinite(O.f[a]) || !std::isfinite(O.f[3 + a])) finite = false;
        if (O.f[3 + a] != 0.0f) { nz++; axis = a; }
      }
      // (and the point within 1e37: with the kernel's |origin| < 8.5e37 gate no component of point - origin overflows to
      //  inf, which the reference would multiply by the normal's zero into a NaN — ADVICE r03)
      for (int a = 0; a < 3; a++)
        if (!(fabsf(O.f[a]) <= 1e37f)) finite = false;
      if (finite && nz == 1) axis_planes[axis].push_back(&O);
      else general_planes.push_back(&O);
    } else if (O.type == CTR_OBJ_MESH) { if (O.tri_count) meshes_in.push_back(O); }  // an empty mesh is never hit
    else oloop.push_back(O);
  }
  {
    auto empty_pair = [] {
      DPlanePair pr;
      for (int a = 0; a < 3; a++) { pr.p[a][0] = pr.p[a][1] = 0.0f; pr.n[a][0] = pr.n[a][1] = 1.0f; }
      pr.index[0] = pr.index[1] = CTR_PLANE_PAD;
      pr.transparent[0] = pr.transparent[1] = 0;
      return pr;
    };
    auto put = [](DPlanePair &pr, int slot, const DObj &O) {
      for (int a = 0; a < 3; a++) { pr.p[a][slot] = O.f[a]; pr.n[a][slot] = O.f[3 + a]; }
      pr.index[slot] = O.index;
      uint32_t tr;
      memcpy(&tr, &O.f[7], sizeof(tr));
      pr.transparent[slot] = tr;
    };
    // fewer than three axis-aligned planes (a lone floor): one general record is less to fetch than a triple
    if (axis_planes[0].size() + axis_planes[1].size() + axis_planes[2].size() < 3) {
      general_planes.clear();
      for (int a = 0; a < 3; a++) axis_planes[a].clear();
      for (const DObj &O : objs)
        if (O.type == CTR_OBJ_PLANE) general_planes.push_back(&O);
    }
    // axis triples: as many as the busiest axis needs (a box room: one)
    size_t triples = 0;
    for (int a = 0; a < 3; a++) triples = std::max(triples, (axis_planes[a].size() + 1) / 2);
    for (size_t t = 0; t < triples; t++)
      for (int a = 0; a < 3; a++) {
        DPlanePair pr = empty_pair();
        for (int slot = 0; slot < 2; slot++)
          if (2 * t + slot < axis_planes[a].size()) put(pr, slot, *axis_planes[a][2 * t + slot]);
        // an empty slot copies its neighbour (same numbers, never tested)
        if (pr.index[0] != CTR_PLANE_PAD && pr.index[1] == CTR_PLANE_PAD)
          for (int q = 0; q < 3; q++) { pr.p[q][1] = pr.p[q][0]; pr.n[q][1] = pr.n[q][0]; }
        planes.push_back(pr);
      }
    n_axis_recs = (uint32_t)planes.size();
    for (size_t k = 0; k < general_planes.size(); k++) {
      if (k % 2 == 0) {
        DPlanePair pr = empty_pair();
        put(pr, 0, *general_planes[k]);
        for (int q = 0; q < 3; q++) { pr.p[q][1] = pr.p[q][0]; pr.n[q][1] = pr.n[q][0]; }
        planes.push_back(pr);
      } else {
        put(planes.back(), 1, *general_planes[k]);
      }
    }
  }
  // top-level BVH over the mesh boxes, one mesh per leaf (same node layout as the per-mesh trees)
  uint32_t tlas_root = BVH_LEAF_FLAG, tlas_begin = (uint32_t)nodes.size();
  float tl_mn[3] = {0, 0, 0}, tl_mx[3] = {0, 0, 0};
  if (!meshes_in.empty()) {
    std::vector<BvhInput> prims(meshes_in.size());
    for (int a = 0; a < 3; a++) { tl_mn[a] = INFINITY; tl_mx[a] = -INFINITY; }
    for (size_t k = 0; k < meshes_in.size(); k++) {
      for (int a = 0; a < 3; a++) {
        prims[k].mn[a] = meshes_in[k].f[a];
        prims[k].mx[a] = meshes_in[k].f[3 + a];
        prims[k].c[a] = 0.5f * (prims[k].mn[a] + prims[k].mx[a]);
        tl_mn[a] = fminf(tl_mn[a], prims[k].mn[a]);
        tl_mx[a] = fmaxf(tl_mx[a], prims[k].mx[a]);
      }
    }
    std::vector<DNode> tnodes;
    std::vector<uint32_t> order;
    bvh_build(prims, 1, tnodes, order, tlas_root);
    nodes.insert(nodes.end(), tnodes.begin(), tnodes.end());
    for (uint32_t k : order) meshes.push_back(meshes_in[k]);
    for (size_t pos = 0; pos < meshes.size(); pos++)
      for (ctr_scene::MeshGuard &g : guards)
        if (g.obj_index == meshes[pos].index) g.mesh_pos = (int)pos;
  }
  // ---- ONE four-wide tree over the triangles of all meshes (render_kernel.hip "merged walk") ----
  // With several meshes a cast otherwise walks the top-level tree, runs the reference's AABB test per mesh it reaches and
  // sets up a walk per mesh it enters (a fifth of the vector and the densest scalar code of the 16-mesh frame).  The merged
  // tree's records carry (mesh rank << 24 | file index) as their tie-break key — the reference's (object, triangle) order
  // as one integer — and the mesh's own AABB test (default_schema.hpp:99-114: a ray that fails it misses the mesh whatever
  // its triangles say) is applied afterwards, to the lanes the walk found something for.
  // Built on demand (build_merged_tree, at the first ctr_set_variant with CTR_VAR_MERGE): here only its room in the device
  // arrays is set aside and the triangles are kept.
  ctr_scene::Merged merged;
  std::vector<DObj> by_rank;  // non-empty meshes in scene order
  for (const DObj &O : objs)
    if (O.type == CTR_OBJ_MESH && O.tri_count) by_rank.push_back(O);
  {
    uint64_t total = 0;
    for (const DObj &O : by_rank) total += O.tri_count;
    if (by_rank.size() >= 2 && by_rank.size() <= CTR_MERGE_MAX_MESHES && total <= 0xFFFFFFull) {
      merged.reserved = true;
      merged.src.reserve(total);
      for (uint32_t r = 0; r < by_rank.size(); r++) {
        merged.slot_of.push_back((uint32_t)merged.src.size());
        const ctr_object &o = d->objects[by_rank[r].index];
        merged.src.insert(merged.src.end(), d->triangles + o.tri_begin, d->triangles + o.tri_begin + o.tri_count);
      }
      merged.slot_of.push_back((uint32_t)total);
      merged.tri_begin = (uint32_t)tris.size();
      merged.tri_count = (uint32_t)total;
      merged.node_begin = (uint32_t)nodes4.size();
      merged.node_cap = (uint32_t)(total / 2 + 64);  // (a four-wide node has at least two children: far fewer in practice)
      DNode4 zero4;
      memset(&zero4, 0, sizeof(zero4));
      nodes4.resize(nodes4.size() + merged.node_cap + CTR_MERGED_SPARE_NODES, zero4);
      DTri zero_t;
      memset(&zero_t, 0, sizeof(zero_t));
      tris.resize(tris.size() + total + CTR_GUARD_SLOTS, zero_t);
      gn.resize(4 * tris.size());  // (unused for the merged records: normals are looked up in the mesh's own range)
      // the pseudo mesh record: the kernel's mesh code walks it like a mesh whose AABB every lane passes
      DObj P;
      memset(&P, 0, sizeof(P));
      P.type = CTR_OBJ_MERGED;
      P.tri_begin = merged.tri_begin;
      P.tri_count = merged.tri_count;
      P.node_begin = merged.node_begin;
      P.node_count = 0;
      P.bvh_root = 0;
      P.index = 0xFFFFFFFFu;
      for (int a = 0; a < 3; a++) { P.f[a] = tl_mn[a]; P.f[3 + a] = tl_mx[a]; }  // (margin of the box tests: the box of all meshes)
      meshes.push_back(P);
      for (const DObj &O : by_rank) meshes.push_back(O);
    }
  }
  std::vector<DLight> lights(d->n_lights);
  for (uint64_t i = 0; i < d->n_lights; i++) {
    const ctr_light &l = d->lights[i];
    lights[i] = DLight{l.type, l.v.x, l.v.y, l.v.z, l.color.x, l.color.y, l.color.z, 0.f};
  }
  std::vector<DMat> mats(d->n_materials);
  bool all_opaque = true, need_cold = false, any_bounce = false;
  for (uint64_t i = 0; i < d->n_materials; i++) {
    const ctr_material &m = d->materials[i];
    mats[i] = DMat{m.color.x, m.color.y, m.color.z, m.specular, m.reflexivity, m.phong_exp, m.transparency, 0.f};
    if (!(m.transparency == 0.0f)) all_opaque = false;
    if ((double)m.transparency >= 1e-6 && (double)m.reflexivity >= 1e-6) need_cold = true;
    if ((double)m.transparency >= 1e-6 || (double)m.reflexivity >= 1e-6) any_bounce = true;
  }

  dbg_stamp("planes, top level, materials");
  auto *s = new ctr_scene();
  s->device = device;
  s->n_obj = (uint32_t)objs.size();
  s->n_tri = (uint32_t)tris.size();
  s->n_light = (uint32_t)lights.size();
  s->n_mat = (uint32_t)mats.size();
  s->has_mesh = has_mesh;
  for (const ctr_scene::MeshGuard &g : guards) s->mesh_tris += g.tri_count;
  s->mesh_bytes = tris.size() * sizeof(DTri) + nodes.size() * sizeof(DNode) + nodes4.size() * sizeof(DNode4);
  s->all_opaque = all_opaque;
  s->need_cold = need_cold;
  s->any_bounce = any_bounce;
  DCam cam = to_dcam(d->cam);
  s->cam = cam;

  auto upload = [&](void **dst, const void *src, size_t bytes) -> hipError_t {
    // never hand the kernel a null base pointer: allocate at least one element's worth
    // (and 256 bytes beyond the end: the leaf loop requests the three cache lines after a leaf's first triangle ahead
    //  of their use, whether the leaf has that many triangles or not — render_kernel.hip, "touch")
    hipError_t er = hipMalloc(dst, (bytes ? bytes : 64) + 256);
    if (er != hipSuccess) return er;
    if (bytes) er = hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice);
    return er;
  };
  hipError_t er;
  s->n_oloop = (uint32_t)oloop.size();
  s->n_mesh = (uint32_t)meshes_in.size();  // (meshes[] may continue with the merged pseudo mesh and the meshes in scene order)
  s->tlas_root = tlas_root;
  s->tlas_begin = tlas_begin;
  for (int q = 0; q < 3; q++) { s->tl_mn[q] = tl_mn[q]; s->tl_mx[q] = tl_mx[q]; }
  s->n_plane_recs = (uint32_t)planes.size();
  s->n_axis_recs = n_axis_recs;
  if ((er = upload((void **)&s->d_objs, objs.data(), objs.size() * sizeof(DObj))) != hipSuccess ||
      (er = upload((void **)&s->d_oloop, oloop.data(), oloop.size() * sizeof(DObj))) != hipSuccess ||
      (er = upload((void **)&s->d_meshes, meshes.data(), meshes.size() * sizeof(DObj))) != hipSuccess ||
      (er = upload((void **)&s->d_planes, planes.data(), planes.size() * sizeof(DPlanePair))) != hipSuccess ||
      (er = upload((void **)&s->d_tris, tris.data(), tris.size() * sizeof(DTri))) != hipSuccess ||
      (er = upload((void **)&s->d_nodes, nodes.data(), nodes.size() * sizeof(DNode))) != hipSuccess ||
      (er = upload((void **)&s->d_nodes4, nodes4.data(), nodes4.size() * sizeof(DNode4))) != hipSuccess ||
      (er = upload((void **)&s->d_gnorm, gn.data(), gn.size() * sizeof(float))) != hipSuccess ||
      (er = upload((void **)&s->d_lights, lights.data(), lights.size() * sizeof(DLight))) != hipSuccess ||
      (er = upload((void **)&s->d_mats, mats.data(), mats.size() * sizeof(DMat))) != hipSuccess ||
      (er = upload((void **)&s->d_cams, &s->cam, sizeof(DCam))) != hipSuccess ||
      (er = hipMalloc((void **)&s->d_counters, 16 * sizeof(unsigned long long))) != hipSuccess ||
      (er = hipMalloc((void **)&s->d_shards, (size_t)CTR_SHARDS * CTR_SHARD_WORDS * sizeof(unsigned long long))) != hipSuccess ||
      (er = hipMemset(s->d_shards, 0, (size_t)CTR_SHARDS * CTR_SHARD_WORDS * sizeof(unsigned long long))) != hipSuccess ||
      (er = hipEventCreate(&s->ev0)) != hipSuccess || (er = hipEventCreate(&s->ev1)) != hipSuccess) {
    ctr_scene_destroy(s);
    return hip_fail(er, "scene upload");
  }
  dbg_stamp("device allocation + upload");
  s->n_cams = 1;
  s->guards = std::move(guards);
  s->merged = std::move(merged);
  s->h_nodes4 = nodes4;
  s->h_tris = tris;
  s->h_gn = gn;
  s->h_objs = objs;
  s->h_meshes = meshes;
  s->h_cams.assign(1, s->cam);
  s->h_lights = lights;
  s->h_mats = mats;
  dbg_stamp("host copies");
  if (int st = refresh_linear_meshes(s)) {
    ctr_scene_destroy(s);
    return st;
  }
  dbg_stamp("guard records");
  *out = s;
  return CTR_OK;
}

int ctr_scene_set_cameras(ctr_scene *s, const ctr_camera *cams, uint32_t n) {
  if (!s || !cams || n == 0) return fail(CTR_E_INVALID, "ctr_scene_set_cameras: bad argument");
  std::vector<DCam> dc(n);
  for (uint32_t i = 0; i < n; i++) {
    if (cams[i].w != cams[0].w || cams[i].h != cams[0].h)
      return fail(CTR_E_INVALID, "ctr_scene_set_cameras: all cameras of a batch must share width and height");
    dc[i] = to_dcam(cams[i]);
  }
  if (dc[0].w == 0 || dc[0].h == 0) return fail(CTR_E_INVALID, "camera has zero width or height");
  std::lock_guard<std::mutex> lk(s->mtx);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());  // no launch may still be reading the old array
  DCam *nd = nullptr;
  HIP_TRY(hipMalloc((void **)&nd, sizeof(DCam) * n));
  hipError_t e = hipMemcpy(nd, dc.data(), sizeof(DCam) * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(nd); return hip_fail(e, "camera upload"); }
  if (s->d_cams) (void)hipFree(s->d_cams);
  s->d_cams = nd;
  s->n_cams = n;
  s->cams_epoch++;
  s->cam = dc[0];
  s->h_cams = dc;
  return refresh_linear_meshes(s);
}

void ctr_scene_destroy(ctr_scene *s) {
  if (!s) return;
  (void)hipSetDevice(s->device);
  for (void *p : {(void *)s->d_objs, (void *)s->d_oloop, (void *)s->d_meshes, (void *)s->d_planes, (void *)s->d_tris, (void *)s->d_nodes, (void *)s->d_nodes4, (void *)s->d_gnorm, (void *)s->d_lights, (void *)s->d_mats, (void *)s->d_cams,
                  (void *)s->d_out, (void *)s->d_uv, (void *)s->d_groups, (void *)s->d_counters, (void *)s->d_shards, (void *)s->d_cost, (void *)s->d_order})
    if (p) (void)hipFree(p);
  if (s->h_counters) (void)hipHostFree(s->h_counters);
  if (s->h_groups) (void)hipHostFree(s->h_groups);
  if (s->ev0) (void)hipEventDestroy(s->ev0);
  if (s->ev1) (void)hipEventDestroy(s->ev1);
  delete s;
}

int ctr_scene_size(const ctr_scene *s, uint64_t *w, uint64_t *h) {
  if (!s) return fail(CTR_E_INVALID, "null scene");
  if (w) *w = s->cam.w;
  if (h) *h = s->cam.h;
  return CTR_OK;
}

int ctr_scene_set_size(ctr_scene *s, uint64_t w, uint64_t h) {
  if (!s || w == 0 || h == 0 || w > 0x7FFFFFFFull || h > 0x7FFFFFFFull) return fail(CTR_E_INVALID, "bad size");
  std::lock_guard<std::mutex> lk(s->mtx);
  HIP_TRY(hipSetDevice(s->device));
  HIP_TRY(hipDeviceSynchronize());
  std::vector<DCam> dc(s->n_cams);
  HIP_TRY(hipMemcpy(dc.data(), s->d_cams, sizeof(DCam) * s->n_cams, hipMemcpyDeviceToHost));
  for (DCam &c : dc) { c.w = (uint32_t)w; c.h = (uint32_t)h; }
  HIP_TRY(hipMemcpy(s->d_cams, dc.data(), sizeof(DCam) * s->n_cams, hipMemcpyHostToDevice));
  s->cam.w = (uint32_t)w;
  s->cam.h = (uint32_t)h;
  s->cams_epoch++;
  return CTR_OK;
}

int ctr_set_variant(ctr_scene *s, uint32_t bits) {
  if (!s) return fail(CTR_E_INVALID, "null scene");
  constexpr uint32_t KNOWN = CTR_VAR_NO_PREFILTER | CTR_VAR_NO_ANYHIT | CTR_VAR_NO_CLUSTER | CTR_VAR_STATS | CTR_VAR_EXACT_POW |
                             CTR_VAR_NO_REORDER | CTR_VAR_NO_OCC6 | CTR_VAR_NO_DIRECT | CTR_VAR_IMAGE_ORDER_FIRST | CTR_VAR_MERGE | CTR_VAR_IGNORE_TRANSPARENT;
  if (bits & ~KNOWN) return fail(CTR_E_INVALID, "ctr_set_variant: unknown variant bits " + std::to_string(bits & ~KNOWN));
  s->user_variant = bits;
  if ((bits & CTR_VAR_MERGE) && s->merged.reserved && !s->merged.built) {
    std::lock_guard<std::mutex> lk(s->mtx);
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    return build_merged_tree(s);
  }
  return CTR_OK;
}

int ctr_render_device_batch(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, uint32_t first_frame,
                            uint32_t n_frames, uint64_t frame_stride_px, uint32_t part_stride, void *d_depth,
                            void *d_color3, void *d_normal3, void *d_counters, void *hip_stream) {
  int st = check_args(s, bounces);
  if (st) return st;
  if (!d_depth || !d_color3 || !d_normal3) return fail(CTR_E_INVALID, "null output buffer");
  if (s->user_variant & CTR_VAR_IGNORE_TRANSPARENT) return fail(CTR_E_INVALID, "CTR_VAR_IGNORE_TRANSPARENT: host-buffer calls only (ctr_render, ctr_render_uv)");
  if (n_frames == 0 || first_frame >= s->n_cams || n_frames > s->n_cams - first_frame)
    return fail(CTR_E_INVALID, "frame range exceeds the cameras set with ctr_scene_set_cameras");
  int cur = -1;
  if (hipGetDevice(&cur) == hipSuccess && cur != s->device) HIP_TRY(hipSetDevice(s->device));
  RenderLaunch L{};
  fill_launch(s, L);
  if ((st = make_rows(s, rows, L.rows))) return st;
  if (part_stride && L.rows.n_parts > 1) {
    // parts rotate: size the per-frame tile grid for the largest part
    L.rows.part_stride = part_stride % L.rows.n_parts;
    uint32_t cap = 0;
    for (uint32_t p = 0; p < L.rows.n_parts; p++) {
      ctr_rows rp = *rows;
      rp.part = p;
      DRows tmp{};
      if ((st = make_rows(s, &rp, tmp))) return st;
      cap = tmp.n_rows > cap ? tmp.n_rows : cap;
    }
    L.rows.n_rows = cap;
  }
  if (n_frames > 1 && frame_stride_px < (uint64_t)L.rows.n_rows * s->cam.w)
    return fail(CTR_E_INVALID, "frame_stride_px smaller than one frame's rows");
  L.first_frame = first_frame;
  L.n_frames = n_frames;
  L.frame_stride_px = frame_stride_px;
  L.fudge = fudge;
  L.bounces = bounces;
  L.depth = (float *)d_depth;
  L.color = (float *)d_color3;
  L.normal = (float *)d_normal3;
  L.counters = (unsigned long long *)d_counters;
  L.variant = s->kernel_variant(false);
  use_merged_tree(s, L);
  {
    std::lock_guard<std::mutex> lk(s->mtx);
    if ((st = attach_order(s, L, false))) return st;
  }
  int e = ctr_launch_render(L, hip_stream);
  if (e) return hip_fail((hipError_t)e, "render kernel launch");
  return CTR_OK;
}

int ctr_render_device(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, void *d_depth, void *d_color3,
                      void *d_normal3, void *d_counters, void *hip_stream) {
  return ctr_render_device_batch(s, fudge, bounces, rows, 0, 1, 0, 0, d_depth, d_color3, d_normal3, d_counters, hip_stream);
}

static int render_host(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, float *depth, float *color3,
                       float *normal3, ctr_render_stats *stats, bool count, unsigned long long *aabb_tris, float *uv2 = nullptr) {
  auto t0 = std::chrono::high_resolution_clock::now();
  int st = check_args(s, bounces);
  if (st) return st;
  std::lock_guard<std::mutex> lk(s->mtx);
  HIP_TRY(hipSetDevice(s->device));
  RenderLaunch L{};
  fill_launch(s, L);
  if ((st = make_rows(s, rows, L.rows))) return st;
  const size_t px = (size_t)L.rows.n_rows * s->cam.w;
  // Page-locked destinations (ctr_frame_alloc, hipHostMalloc, mapped hipHostRegister) are visible to the device:
  // the kernel then delivers the frame ITSELF, group of tiles by group of tiles while it renders (render_kernel.hip
  // "Host delivery"), so the 28 bytes per pixel cross PCIe underneath the rendering instead of in a DMA after it.
  // Any other destination: device buffers + copies, below.
  float *zd = nullptr, *zc = nullptr, *zn = nullptr;
  const bool merge_wanted = (s->user_variant & CTR_VAR_MERGE) && s->merged.built && s->merged.usable;  // (no delivering build of it)
  const bool direct = px && depth && color3 && normal3 && !count && !uv2 && !(s->user_variant & (CTR_VAR_NO_DIRECT | CTR_VAR_STATS | CTR_VAR_IGNORE_TRANSPARENT)) && !merge_wanted &&
                      ctr_host_delivery_available(s->kernel_variant(false)) &&
                      is_pinned(depth) && is_pinned(depth + px - 1) && is_pinned(color3) && is_pinned(color3 + 3 * px - 1) &&
                      is_pinned(normal3) && is_pinned(normal3 + 3 * px - 1) && device_view(depth, &zd) &&
                      device_view(color3, &zc) && device_view(normal3, &zn);
  const size_t spx = direct ? (size_t)ctr_staging_pixels(L) : (px ? px : 1);  // pixels per output buffer on the device
  if ((st = ensure_outputs(s, spx))) return st;
  L.fudge = fudge;
  L.bounces = bounces;
  L.depth = s->d_out;
  L.color = s->d_out + spx;
  L.normal = s->d_out + 4 * spx;
  if (direct) {
    const size_t groups = (size_t)ctr_staging_groups(L);
    if (groups > s->groups_cap) {
      if (s->d_groups) (void)hipFree(s->d_groups);
      s->d_groups = nullptr;
      s->groups_cap = 0;
      if (s->h_groups) (void)hipHostFree(s->h_groups);
      s->h_groups = nullptr;
      HIP_TRY(hipMalloc((void **)&s->d_groups, groups * sizeof(uint32_t)));
      HIP_TRY(hipHostMalloc((void **)&s->h_groups, groups * sizeof(uint32_t), hipHostMallocDefault));
      s->groups_cap = groups;
    }
    // cleared at the head of EVERY direct launch (a few microseconds): whatever an earlier launch left behind — one
    // that faulted or was cut short included — this one starts from zero
    HIP_TRY(hipMemsetAsync(s->d_groups, 0, groups * sizeof(uint32_t), nullptr));
    L.host_depth = zd;
    L.host_color = zc;
    L.host_normal = zn;
    L.group_done = s->d_groups;
  }
  L.counters = s->d_counters;
  L.variant = s->kernel_variant(count);
  const bool igntr = (s->user_variant & CTR_VAR_IGNORE_TRANSPARENT) != 0 && !count;
  if ((uv2 || igntr) && px) {
    if (count || (s->user_variant & CTR_VAR_STATS)) return fail(CTR_E_INVALID, "ctr_render_uv / CTR_VAR_IGNORE_TRANSPARENT: not with the counting / statistics variants");
    if (px > s->uv_px) {
      if (s->d_uv) (void)hipFree(s->d_uv);
      s->d_uv = nullptr;
      s->uv_px = 0;
      HIP_TRY(hipMalloc((void **)&s->d_uv, sizeof(float) * 2 * px));
      s->uv_px = px;
    }
    L.uv = s->d_uv;
    L.variant = (L.variant & (KV_ANYHIT | KV_FASTPOW)) | KV_PREFILTER | KV_BVH | KV_UV | (igntr ? KV_IGNTR : 0u);
  }
  use_merged_tree(s, L);
  if ((st = attach_order(s, L, count))) return st;
  HIP_TRY(hipMemsetAsync(s->d_counters, 0, 16 * sizeof(unsigned long long), nullptr));
  HIP_TRY(hipEventRecord(s->ev0, nullptr));
  int e = ctr_launch_render(L, nullptr);
  if (e) return hip_fail((hipError_t)e, "render kernel launch");
  HIP_TRY(hipEventRecord(s->ev1, nullptr));
  // Copy-out (the reference does 3·h row-wise copies, kernel.hpp:110-114).  Page-locked destinations
  // (ctr_frame_alloc, hipHostMalloc, hipHostRegister) are written by direct DMA queued behind the kernel:
  // ONE transfer when the three buffers are the consecutive parts of one block, else one per buffer.
  // Pageable destinations go through the runtime's staged copy, one call per buffer.
  if (px && !direct) {
    const bool packed = depth && color3 == depth + px && normal3 == color3 + 3 * px;
    if (packed && is_pinned(depth) && is_pinned(normal3 + 3 * px - 1)) {
      HIP_TRY(hipMemcpyAsync(depth, s->d_out, sizeof(float) * 7 * px, hipMemcpyDeviceToHost, nullptr));
    } else {
      auto out = [&](float *dst, const float *src, size_t n) -> hipError_t {
        if (!dst) return hipSuccess;
        if (is_pinned(dst) && is_pinned(dst + n - 1)) return hipMemcpyAsync(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost, nullptr);
        return hipMemcpy(dst, src, sizeof(float) * n, hipMemcpyDeviceToHost);
      };
      HIP_TRY(out(depth, L.depth, px));
      HIP_TRY(out(color3, L.color, 3 * px));
      HIP_TRY(out(normal3, L.normal, 3 * px));
    }
    if (uv2) HIP_TRY(hipMemcpy(uv2, s->d_uv, sizeof(float) * 2 * px, hipMemcpyDeviceToHost));
  }
  HIP_TRY(hipMemcpyAsync(s->h_counters, s->d_counters, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, nullptr));
  const size_t n_groups = direct ? (size_t)ctr_staging_groups(L) : 0;
  if (direct) HIP_TRY(hipMemcpyAsync(s->h_groups, s->d_groups, n_groups * sizeof(uint32_t), hipMemcpyDeviceToHost, nullptr));
  HIP_TRY(hipStreamSynchronize(nullptr));
  if (direct) {
    // every group of tiles must have counted all its tiles, or its pixels never left for the caller's buffers
    // (how many tiles a group has is the kernel's business — its tile shape is a build option: ctr_group_tile_count)
    size_t missing = 0;
    for (size_t g = 0; g < n_groups; g++)
      if (s->h_groups[g] != ctr_group_tile_count(L, g)) missing++;
    if (missing) {
      s->order_valid = false;  // whatever order that launch ran in is not to be trusted
      return fail(CTR_E_DELIVERY, std::to_string(missing) + " of " + std::to_string(n_groups) +
                  " tile groups were not delivered to the caller's buffers (incomplete launch); render again");
    }
  }
  if (direct && getenv("CUTRACE_VERIFY_DELIVERY")) {
    // Debug aid for the kernel's own delivery (render_kernel.hip "Host delivery" relies on write-through stores and
    // scoped loads instead of fences): the tile-major staging copy of the frame is still on the device — fetch it
    // and compare every pixel with what arrived in the caller's buffers.
    std::vector<float> stg(7 * spx);
    HIP_TRY(hipMemcpy(stg.data(), s->d_out, sizeof(float) * 7 * spx, hipMemcpyDeviceToHost));
    const uint32_t w = s->cam.w;
    uint64_t bad = 0;
    for (uint32_t y = 0; y < L.rows.n_rows; y++)
      for (uint32_t x = 0; x < w; x++) {
        const size_t at = (size_t)y * w + x, sp = (size_t)ctr_staging_index(L, x, y);
        bool ok = memcmp(&depth[at], &stg[sp], 4) == 0;
        ok = ok && memcmp(&color3[3 * at], &stg[spx + 3 * sp], 12) == 0 && memcmp(&normal3[3 * at], &stg[4 * spx + 3 * sp], 12) == 0;
        bad += ok ? 0 : 1;
      }
    if (bad) return fail(CTR_E_INVALID, "CUTRACE_VERIFY_DELIVERY: " + std::to_string(bad) + " delivered pixels differ from the staged frame");
  }
  float ms = 0.f;
  HIP_TRY(hipEventElapsedTime(&ms, s->ev0, s->ev1));
  unsigned long long cnt[16];
  memcpy(cnt, s->h_counters, sizeof(cnt));
  memcpy(s->last_cnt, cnt, sizeof(cnt));
  if (s->user_variant & CTR_VAR_STATS)
    fprintf(stderr, "cutrace_amd stats: wave_casts=%llu nodes=%llu tri_prefilter=%llu tri_exact=%llu mesh_entries=%llu "
                    "active_lanes=%llu node_lanes=%llu prefilter_lanes=%llu exact_lanes=%llu lane_max_nodes=%llu "
                    "lane_max_tris=%llu kernel_ms=%.3f\n", cnt[4], cnt[5],
            cnt[6], cnt[7], cnt[8], cnt[9], cnt[10], cnt[11], cnt[12], cnt[13], cnt[14], ms);
  if (aabb_tris) *aabb_tris = cnt[2];
  if (cnt[13] && !(s->user_variant & CTR_VAR_STATS))  // CTR_TIMING diagnostic build: share of the waves' lifetime
    fprintf(stderr, "cutrace_amd timing (%% of wave cycles): cast_setup=%.1f planes=%.1f object_loop=%.1f tlas+aabb=%.1f "
                    "mesh_setup=%.1f bvh_nodes=%.1f leaves=%.1f cont_mode=%.1f cont_rest=%.1f | wave_total=%llu kernel_ms=%.3f\n",
            100.0 * cnt[4] / cnt[13], 100.0 * cnt[5] / cnt[13], 100.0 * cnt[6] / cnt[13], 100.0 * cnt[7] / cnt[13],
            100.0 * cnt[8] / cnt[13], 100.0 * cnt[9] / cnt[13], 100.0 * cnt[10] / cnt[13], 100.0 * cnt[11] / cnt[13],
            100.0 * cnt[12] / cnt[13], cnt[13], ms);
  auto t1 = std::chrono::high_resolution_clock::now();
  if (stats) {
    stats->kernel_ms = ms;
    stats->total_ms = std::chrono::duration<double, std::milli>(t1 - t0).count();
    stats->ray_count = cnt[0];
    stats->rows = L.rows.n_rows;
    uint32_t bits = (uint32_t)cnt[1];
    float md;
    memcpy(&md, &bits, 4);
    stats->max_depth = md;  // largest finite depth, 0 if none (kernel.hpp:120-125)
    stats->reserved = 0;
  }
  return CTR_OK;
}

int ctr_render(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, float *depth, float *color3,
               float *normal3, ctr_render_stats *stats) {
  return render_host(s, fudge, bounces, rows, depth, color3, normal3, stats, false, nullptr);
}

int ctr_render_uv(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, float *depth, float *color3,
                  float *normal3, float *uv2, ctr_render_stats *stats) {
  if (!uv2) return fail(CTR_E_INVALID, "ctr_render_uv: null uv buffer");
  return render_host(s, fudge, bounces, rows, depth, color3, normal3, stats, false, nullptr, uv2);
}

int ctr_debug_poison_next_order(ctr_scene *s) {
  if (!s) return fail(CTR_E_INVALID, "null scene");
  std::lock_guard<std::mutex> lk(s->mtx);
  s->poison_next_order = true;
  return CTR_OK;
}

int ctr_last_counters(ctr_scene *s, uint64_t *out16) {
  if (!s || !out16) return fail(CTR_E_INVALID, "ctr_last_counters: null argument");
  std::lock_guard<std::mutex> lk(s->mtx);
  for (int q = 0; q < 16; q++) out16[q] = s->last_cnt[q];
  return CTR_OK;
}

int ctr_tile_costs(ctr_scene *s, uint32_t *out, uint64_t capacity, uint64_t *n_tiles) {
  if (!s) return fail(CTR_E_INVALID, "null scene");
  std::lock_guard<std::mutex> lk(s->mtx);
  const uint64_t n = s->order_valid ? s->order_key[0] : 0;
  if (n_tiles) *n_tiles = n;
  if (out && n) {
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out, s->d_cost, sizeof(uint32_t) * (n < capacity ? n : capacity), hipMemcpyDeviceToHost));
  }
  return CTR_OK;
}

int ctr_frame_alloc(uint64_t n_pixels, float **depth, float **color3, float **normal3) {
  if (!depth || !color3 || !normal3 || n_pixels == 0) return fail(CTR_E_INVALID, "ctr_frame_alloc: bad argument");
  float *p = nullptr;
  // portable + mapped: page-locked for, and visible to, every device of the process (whichever one is current now)
  HIP_TRY(hipHostMalloc((void **)&p, sizeof(float) * 7 * n_pixels, hipHostMallocPortable | hipHostMallocMapped));
  *depth = p;
  *color3 = p + n_pixels;
  *normal3 = p + 4 * n_pixels;
  return CTR_OK;
}

void ctr_frame_free(float *depth) {
  if (depth) (void)hipHostFree(depth);
}

int ctr_algorithmic_bytes(ctr_scene *s, float fudge, int bounces, const ctr_rows *rows, uint64_t *bytes,
                          uint64_t *ray_count) {
  ctr_render_stats stt{};
  unsigned long long aabb = 0;
  int st = render_host(s, fudge, bounces, rows, nullptr, nullptr, nullptr, &stt, true, &aabb);
  if (st) return st;
  // 56·N_obj per ray_cast + 48·N_tri per AABB-hit mesh + 28 B per pixel written
  if (bytes) *bytes = 56ull * s->n_obj * stt.ray_count + 48ull * aabb + 28ull * stt.rows * s->cam.w;
  if (ray_count) *ray_count = stt.ray_count;
  return CTR_OK;
}

}  // extern "C"
