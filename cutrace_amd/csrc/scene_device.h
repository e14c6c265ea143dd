// scene_device.h — HBM layout of the flat scene the render kernel consumes.
//
// The reference keeps a 2-level pointer graph in managed memory
// (scene.objects[i] → gpu_variant → mesh.triangles.buffer, inc/gpu_types.hpp:263-274,
// inc/cpu_to_gpu.hpp:69-198).  Here the scene is five dense arrays uploaded once:
//
//   DObj   objs[n_obj]      64 B records, read wave-uniformly (scalar loads)
//   DTri   tris[n_tri]      64 B records = one cache line = one s_load_dwordx16;
//                           per mesh in BVH-leaf order, each carrying its ORIGINAL file
//                           index (ties on t are broken by it, so results do not depend
//                           on the order); then stand-alone triangles
//   DNode4 nodes4[n_node4]  128 B records, per-mesh four-wide BVH, nodes holding their children's boxes (bvh.h)
//   DNode  nodes[n_node]    64 B records, two-wide tree over the meshes' boxes
//   float4 gnorm[n_tri]     geometric normal of each DTri as the reference computes it on
//                           a hit; only the winning triangle's is ever fetched
//   DLight lights[n_light]  32 B
//   DMat   mats[n_mat]      32 B
//
// Ray-independent arithmetic the reference redoes per ray-triangle test is
// hoisted to upload time WITH THE SAME OPERATIONS (so bits are identical):
//   a = p2-p1, b = p2-p3            (default_schema.hpp:58)
//   normal = -1 * normalize((p2-p3) x (p1-p3))   (default_schema.hpp:72)
// plus prefilter-only data (n = a x b, error scale) whose rounding is irrelevant
// because the prefilter is conservative and every survivor is re-tested exactly.
#ifndef CUTRACE_AMD_SCENE_DEVICE_H
#define CUTRACE_AMD_SCENE_DEVICE_H

#include <stdint.h>

struct DTri {
  float ab[3][2];       // ab[axis][0] = a = p2 - p1, ab[axis][1] = b = p2 - p3: (a, b) of one axis is an aligned SGPR
                        // pair, so that one v_pk_fma_f32 advances a.q and b.q together (prefilter)
  float px, py, pz;     // p2; (px, py) is an aligned pair as well
  uint32_t orig;        // index of this triangle in the ORIGINAL (file-order) array: tie-break key.  In the merged tree:
                        // (rank of its mesh among the scene's meshes << 24) | file index — the reference's order
                        // (ray_cast.hpp:43 first object wins, default_schema.hpp:134 first triangle wins) as ONE integer
  float nx, ny, nz;     // a x b (prefilter only)
  float ke;             // kappa * emax,   emax = max|component of a,b| (prefilter error scale)
  float ke2;            // kappa * emax^2
  float pad1;
};
static_assert(sizeof(DTri) == 64, "DTri must be one 64-byte line");

#define CTR_OBJ_MERGED 4u  /* device-only: the pseudo mesh that stands for ONE tree over the triangles of all meshes */
#define CTR_MERGE_MAX_MESHES 255u
struct DObj {
  uint32_t type;        // CTR_OBJ_*
  uint32_t mat;         // material index
  uint32_t tri_begin;   // mesh: first DTri; triangle: its DTri
  uint32_t tri_count;   // mesh: number of triangles; triangle: 1
  uint32_t node_begin;  // mesh: first DNode4 of its BVH (= its root)
  uint32_t node_count;  // mesh: number of BVH nodes
  uint32_t bvh_root;    // mesh: node the walk starts at: 0 = the root, or node_count = the mesh's spare node, which leads
                        // to its GUARD records (triangles every lane that enters the mesh must meet, whatever their
                        // box; ctr_api.cpp refresh_linear_meshes) and then to the root
  uint32_t index;       // position in the scene's object list (hit_id; ties on t go to the lower index)
  // triangle: f[0..2] p1, f[3..5] p3 (p2 is in its DTri): triangle::uv_for needs the vertices themselves (KV_UV only)
  // mesh    : f[0..2] bbox.min, f[3..5] bbox.max
  // plane   : f[0..2] point,    f[3..5] normal
  // sphere  : f[0..2] center,   f[3] radius, f[4] radius*radius
  // all     : f[7] (as bits) 1 when the object's material is transparent (ray_cast's ignore_transparent, KV_IGNTR only)
  float f[8];
};
static_assert(sizeof(DObj) == 64, "DObj must be 64 bytes");

// planes get their own dense array and loop (no type dispatch): box scenes are mostly planes, and every
// cast visits every one of them.  Two planes share one 64-byte record with their coordinates
// interleaved, so that (plane 0, plane 1) of one coordinate is an aligned SGPR pair and the two
// numerators / denominators are evaluated with packed f32 instructions — same multiplies and adds in
// the same order as the scalar form, two at a time.  A slot without a plane holds harmless numbers and
// index = CTR_PLANE_PAD (never tested).
// Axis-aligned planes (exactly one non-zero normal component: the walls of a box room) come first, in whole TRIPLES
// of records: record 0 of a triple holds up to two planes normal to x, record 1 to y, record 2 to z (n_axis_recs, a
// multiple of 3).  For those the kernel forms (point - origin).normal and dir.normal from the one component that
// is not multiplied by zero — three packed instructions per record instead of thirteen, the same bits whenever the
// value can matter (render_kernel.hip, "axis-aligned planes").  The records are complete, so the general code gives
// the reference's arithmetic on them as well.
struct DPlanePair {
  float p[3][2];        // point:  p[axis][which plane]
  float n[3][2];        // normal
  uint32_t index[2];    // position in the scene's object list
  uint32_t transparent[2];  // 1: the plane's material is transparent (material::is_transparent, default_schema.hpp:334): what
                            // ray_cast's ignore_transparent skips (ray_cast.hpp:39-40; KV_IGNTR only)
};
static_assert(sizeof(DPlanePair) == 64, "DPlanePair must be 64 bytes");
#define CTR_PLANE_PAD 0xFFFFFFFFu

struct DLight {
  uint32_t type;        // CTR_LIGHT_*
  float vx, vy, vz;     // sun: direction; point: position
  float cx, cy, cz;     // color
  float pad;
};
static_assert(sizeof(DLight) == 32, "DLight must be 32 bytes");

struct DMat {
  float cx, cy, cz;     // color
  float specular, reflexivity, phong_exp, transparency;
  float pad;
};
static_assert(sizeof(DMat) == 32, "DMat must be 32 bytes");

struct DCam {
  float pos[3], up[3], forward[3], right[3];
  float ambient;
  uint32_t w, h;
};

// kernel variant bits (internal; selected from CTR_VAR_* + scene properties)
enum : uint32_t {
  KV_PREFILTER = 1u,   // conservative FMA prefilter before the exact Cramer test
  KV_ANYHIT = 2u,      // shadow casts stop at the first occluder (all-opaque scenes only)
  KV_COUNT = 4u,       // also accumulate algorithmic-byte counters
  KV_BVH = 8u,         // walk each mesh through its BVH instead of linearly
  KV_STATS = 16u,      // diagnostic: wave-level work counters into counters[4..9]
  KV_FASTPOW = 32u,    // specular pow() as exp2(e*log2(x)) in f32 instead of f64 pow
  KV_OCC6 = 64u,       // compiled for 6 waves per SIMD instead of 5 (large meshes: latency-bound on scalar-cache misses)
  KV_HOSTOUT = 128u,   // the launch delivers the frame to page-locked host memory itself (RenderLaunch::group_done)
  KV_UV = 256u,        // also write the texture coordinates of the primary hit (ray_cast's tex_coords), RenderLaunch::uv
  KV_IGNTR = 1024u,    // the cast of kernel.hpp:52 (depth, normal, uv) is made with ray_cast's ignore_transparent = true (CTR_VAR_IGNORE_TRANSPARENT)
  KV_MERGE = 512u,     // the walk may meet the merged pseudo mesh: ONE tree over the triangles of all meshes (CTR_VAR_MERGE)
};

struct DRows {
  uint32_t row_begin;   // first global row of this call's selection
  uint32_t row_end;     // one past the last global row
  uint32_t part_stride; // batch: frame f renders part (part + f*part_stride) % n_parts
  uint32_t n_rows;      // local rows per frame (the largest part's row count when parts rotate)
  uint32_t block_rows;  // interleave block height
  uint32_t part, n_parts;
  uint32_t first_block; // index of the first selected block
};

struct RenderLaunch {
  const DObj *objs;        // every object, scene order (hit records)
  const DObj *oloop;       // spheres and stand-alone triangles, scene order (sequential loop)
  const DObj *meshes;      // non-empty meshes in top-level-BVH leaf order; [n_mesh] the merged pseudo mesh, [n_mesh + 1 + r] mesh r in scene order
  uint32_t n_mesh, tlas_root, tlas_begin;
  uint32_t tlas_root_regular;  // the top-level tree over the meshes when tlas_root names the merged tree (else the same)
  float tl_mn[3], tl_mx[3];
  const DPlanePair *planes;  // n_plane_recs records, the first n_axis_recs of them axis triples
  uint32_t n_oloop, n_plane_recs, n_axis_recs;
  const DTri *tris;
  const void *nodes;    // DNode[] (bvh.h): top-level tree over the meshes
  const void *nodes4;   // DNode4[] (bvh.h): per-mesh trees
  const float *gnorm;   // 4 floats per triangle
  const DLight *lights;
  const DMat *mats;
  uint32_t n_obj, n_light, n_mat;
  uint32_t has_mesh;
  uint32_t need_cold_frames;  // some material both reflects and transmits (>= 1e-6 each)
  uint32_t any_bounce;        // some material reflects or transmits: only then does ray_color recurse at all
  const DCam *cams;           // device array, one camera per frame
  uint32_t w, h;
  uint32_t first_frame, n_frames;
  uint64_t frame_stride_px;
  DRows rows;
  float fudge;
  int bounces;
  float *depth;
  float *color;
  float *normal;
  float *uv;                     // KV_UV: 2 floats per pixel (same indexing as depth), else null
  unsigned long long *shards;    // scene-owned CTR_SHARDS x CTR_SHARD_WORDS scratch the kernel adds into
  unsigned long long *counters;  // [0] ray_count, [1] max-depth bits, [2] AABB-hit triangle count (KV_COUNT), [4..9] KV_STATS
  uint32_t variant;
  // tile scheduling (render_kernel.hip "Dispatch order"): all three may be null
  const uint32_t *order;  // dispatch slot -> wave index for THIS launch (a permutation of 0..waves-1)
  uint32_t *cost;         // out: per-wave cost of this launch
  uint32_t *order_next;   // out (or null = keep the old order): waves sorted by descending cost
                          // (may alias `order`: written after the render)
  uint32_t order_init;    // 1: no costs are known for this shape — fill `order` (writable then) with the centre-out
                          // order before the render (render_kernel.hip first_order)
  // Host delivery (render_kernel.hip "Host delivery"; single frame only): when group_done is set, depth / color /
  // normal above are a TILE-MAJOR staging area of ctr_staging_pixels() pixels each (x1, x3, x3 floats), and the
  // frame is written into host_* — device-visible page-locked host memory, compact row-major as in ctr_render — by
  // the wave that completes each group of tiles.  group_done: ctr_staging_groups() zeroed words (left zeroed).
  float *host_depth, *host_color, *host_normal;
  uint32_t *group_done;
};
uint64_t ctr_staging_pixels(const RenderLaunch &L);
uint64_t ctr_staging_groups(const RenderLaunch &L);
uint32_t ctr_group_tile_count(const RenderLaunch &L, uint64_t group);          // tiles group `group` counts when it is complete
uint64_t ctr_staging_index(const RenderLaunch &L, uint32_t x, uint32_t k_row); // staging pixel of compact pixel (x, k_row)
bool ctr_host_delivery_available(uint32_t kernel_variant);

// keeps `msg` for ctr_last_error() (ctr_api.cpp); used by the other translation units of the library
void ctr_internal_set_error(const char *msg);
// host-callable launcher implemented in render_kernel.hip; returns a hipError_t as int
int ctr_launch_render(const RenderLaunch &L, void *stream);
// number of waves (tiles x frames) the launch will dispatch
uint64_t ctr_launch_waves(const RenderLaunch &L);
// counters are accumulated in CTR_SHARDS 128-byte shards (see render_kernel.hip) and folded afterwards
#define CTR_SHARDS 1024
#define CTR_SHARD_WORDS 16
// counters of the tile scheduler's counting sort (64 cost classes x 16 sub-bins)
#define CTR_COST_BINS 1024u
// maximum `bounces` the kernel supports (explicit per-lane stack depth - 1)
#define CTR_MAX_BOUNCES 15

#endif
