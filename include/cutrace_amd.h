/*
 * cutrace_amd.h — C-ABI of the MI355X-native ray-cast + shading path.
 *
 * This is the drop-in boundary for the ONE hot path of jay-tux/cutrace:
 *
 *   template <S, size_t bounces = 10, size_t tpb = 256>
 *   void cutrace::gpu::render(const S &scene, float fudge, float &max,
 *                             grid<float> &depth_map, grid<vector> &color_map,
 *                             grid<vector> &normal_map,
 *                             size_t &render_ms, size_t &total_ms);
 *                                              (reference inc/kernel.hpp:86-130,
 *                                               sole caller main.cu:30)
 *
 * Everything crossing the boundary is a plain pointer, a size or a POD struct:
 * no C++ types, no torch types.  The scene is handed over as FLAT host arrays
 * (ctr_scene_desc) whose elements keep the field order and meaning of the
 * reference's gpu::schema structs (inc/default_schema.hpp:26-396) and the
 * reference's variant tag order (inc/default_schema.hpp:920-922):
 *   objects   triangle=0, mesh=1, plane=2, sphere=3
 *   lights    sun=0, point=1
 *   materials phong(solid)=0
 *
 * Error convention: the reference's render() returns void and its cudaCheck
 * macro prints to stderr and continues (inc/cuda.hpp:12-22).  Here every entry
 * point returns an int status (0 = ok, otherwise a CTR_E_* code or a HIP error
 * number offset by CTR_E_HIP_BASE); the message is also kept for
 * ctr_last_error().  Nothing aborts, nothing throws across the ABI.
 *
 * The library has NO CPU fallback: if no HIP device / code object is available
 * the calls fail with a non-zero status.
 */
#ifndef CUTRACE_AMD_H
#define CUTRACE_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTR_ABI_VERSION 3   /* 3 (round 4): CTR_VAR_IGNORE_TRANSPARENT, CTR_VAR_MERGE, ctr_debug_lane_stats
                             * 2 (round 3): CTR_E_DELIVERY, ctr_multi_submit / ctr_multi_wait, ctr_render_uv, ctr_set_variant rejects unknown bits
                             * (1 silently ignored the bits CTR_VAR_TRI_LDS = 1, CTR_VAR_VMEM = 64, CTR_VAR_SMEM = 128 that round 2 removed) */

/* ---- status codes ------------------------------------------------------- */
#define CTR_OK 0
#define CTR_E_INVALID 1      /* bad argument / malformed scene description   */
#define CTR_E_NO_DEVICE 2    /* no usable HIP device                          */
#define CTR_E_IO 3           /* file could not be read / written              */
#define CTR_E_PARSE 4        /* scene JSON rejected (loader semantics)        */
#define CTR_E_DELIVERY 5     /* a launch that delivers the frame itself ended with tile groups undelivered */
#define CTR_E_HIP_BASE 1000  /* 1000 + hipError_t                             */

/* ---- POD vocabulary (mirrors cutrace::vector, inc/vector.hpp:25-28) ----- */
typedef struct ctr_vec3 { float x, y, z; } ctr_vec3;

/* object tags — reference variant order, inc/default_schema.hpp:920 */
enum { CTR_OBJ_TRIANGLE = 0, CTR_OBJ_MESH = 1, CTR_OBJ_PLANE = 2, CTR_OBJ_SPHERE = 3 };
/* light tags — inc/default_schema.hpp:921 */
enum { CTR_LIGHT_SUN = 0, CTR_LIGHT_POINT = 1 };
/* material tags — inc/default_schema.hpp:922 */
enum { CTR_MAT_PHONG = 0 };

/* One mesh triangle: p1,p2,p3 in file order (gpu::schema::triangle,
 * inc/default_schema.hpp:26-30; mesh triangles all carry the mesh's mat_idx,
 * inc/default_schema.hpp:536-541, so it is not repeated here). */
typedef struct ctr_triangle { ctr_vec3 p1, p2, p3; } ctr_triangle;

/* One scene object (the reference's 56-byte gpu_variant of
 * triangle|mesh|plane|sphere, flattened).
 *   triangle: v0=p1 v1=p2 v2=p3                    (default_schema.hpp:26-30)
 *   mesh    : tri_begin/tri_count index ctr_scene_desc.triangles,
 *             v0=bbox.min v1=bbox.max              (default_schema.hpp:89-92)
 *   plane   : v0=point v1=normal                   (default_schema.hpp:159-162)
 *   sphere  : v0=center f0=radius                  (default_schema.hpp:212-215) */
typedef struct ctr_object {
  uint32_t type;
  uint32_t reserved;
  uint64_t mat_idx;
  ctr_vec3 v0, v1, v2;
  float f0;
  uint64_t tri_begin, tri_count;
} ctr_object;

/* sun: v=direction; point: v=point            (default_schema.hpp:267-311) */
typedef struct ctr_light {
  uint32_t type;
  ctr_vec3 v;
  ctr_vec3 color;
} ctr_light;

/* phong_material                              (default_schema.hpp:319-324) */
typedef struct ctr_material {
  uint32_t type;
  ctr_vec3 color;
  float specular, reflexivity, phong_exp, transparency;
} ctr_material;

/* gpu::schema::cam with the basis ALREADY computed on the host, exactly as the
 * reference does in default_cam::to_gpu → cam::look_at
 * (default_schema.hpp:370-374, 870-874).  ctr_camera_look_at() (cutrace_host.h) does it. */
typedef struct ctr_camera {
  ctr_vec3 pos, up, forward, right;
  float near_plane, far_plane, ambient;
  uint32_t reserved;
  uint64_t w, h;
} ctr_camera;

/* The flat scene = gpu_scene_{objects,lights,materials,cam}
 * (inc/gpu_types.hpp:263-274).  All pointers are HOST memory owned by the
 * caller; ctr_scene_create copies what it needs. */
typedef struct ctr_scene_desc {
  const ctr_object *objects;     uint64_t n_objects;
  const ctr_triangle *triangles; uint64_t n_triangles;
  const ctr_light *lights;       uint64_t n_lights;
  const ctr_material *materials; uint64_t n_materials;
  ctr_camera cam;
} ctr_scene_desc;

/* Which image rows one call renders.  Rows y with
 *     row_begin <= y < row_end  and  ((y / block_rows) % n_parts) == part
 * are rendered, in increasing y, into a COMPACT row-major buffer (the k-th
 * selected row is local row k).  {0,h,h,0,1} (or all zeros) = whole frame.
 * Interleaved row blocks are how the frame is tiled over the GPUs of a node:
 * rank r of n passes {0,h,block_rows,r,n}. */
typedef struct ctr_rows {
  uint64_t row_begin, row_end;
  uint64_t block_rows;
  uint32_t part, n_parts;
} ctr_rows;

typedef struct ctr_scene ctr_scene; /* opaque device-resident scene */

/* Per-call statistics (all optional: pass NULL to skip). */
typedef struct ctr_render_stats {
  double kernel_ms;     /* HIP-event time of the render kernel (≈ render_ms, kernel.hpp:105-108) */
  double total_ms;      /* wall time of the call incl. alloc + D2H (≈ total_ms, kernel.hpp:88,126) */
  uint64_t ray_count;   /* ray_cast invocations of the REFERENCE algorithm for these pixels
                           (incl. the duplicated primary cast kernel.hpp:52 + shading.hpp:123) */
  uint64_t rows;        /* rows rendered by this call */
  float max_depth;      /* largest finite depth, 0 if none (kernel.hpp:120-125) */
  uint32_t reserved;
} ctr_render_stats;

/* ---- library ------------------------------------------------------------- */
int ctr_abi_version(void);
const char *ctr_last_error(void);
/* number of visible HIP devices, or a negative status */
int ctr_device_count(void);

/* ---- scene upload (replaces cpu_to_gpu::convert, inc/cpu_to_gpu.hpp:188-198) -- */
int ctr_scene_create(const ctr_scene_desc *desc, int device, ctr_scene **out);
void ctr_scene_destroy(ctr_scene *scene);
/* image size of the uploaded scene's camera */
int ctr_scene_size(const ctr_scene *scene, uint64_t *w, uint64_t *h);
/* change the camera resolution of an uploaded scene (bench/test override; the
 * reference has no CLI flag for this, main.cu:8-12) */
int ctr_scene_set_size(ctr_scene *scene, uint64_t w, uint64_t h);

/* ---- THE hot path: replaces gpu::render<S,bounces,tpb> (kernel.hpp:86-130) --
 * Host-buffer form.  depth: rows*w floats; color3/normal3: rows*w*3 floats in
 * the AoS layout of grid<vector> (inc/grid.hpp, inc/vector.hpp:25-28), pixel
 * index local_row*w + x (kernel.hpp:54).  Misses: depth=+inf, normal=0,
 * color=0 (kernel.hpp:47,50; shading.hpp:119).  Synchronous. */
int ctr_render(ctr_scene *scene, float fudge, int bounces, const ctr_rows *rows,
               float *depth, float *color3, float *normal3, ctr_render_stats *stats);

/* Page-locked host memory for a frame's three buffers, as the consecutive parts of ONE block
 * [depth n | color 3n | normal 3n floats].  Page-locked destinations (this block, or any hipHostMalloc /
 * mapped hipHostRegister memory, together or one by one) are visible to the device, and ctr_render then has
 * the render kernel deliver the frame ITSELF: finished groups of tiles are copied into the host buffers while
 * the rest of the frame is still being rendered, so the 28 bytes per pixel cross PCIe underneath the kernel
 * instead of in a transfer after it (1920x1080 bunny: 1.4 ms per call against 2.15 ms for kernel + one DMA;
 * first frame 1.66 against 2.6 ms).  Ordinary (pageable) destinations get three hipMemcpy after the kernel: 2.2 ms
 * per call into buffers that are reused, 4-5 ms when the copy is the first to touch the destination's pages.
 * Bits are the same on every path.
 * CTR_VAR_NO_DIRECT falls back to device buffers + DMA (one transfer when the buffers are one block).
 * Replaces the reference's cudaMallocManaged outputs + 3*h row copies (inc/kernel.hpp:99-118).
 * Free with ctr_frame_free(depth). */
int ctr_frame_alloc(uint64_t n_pixels, float **depth, float **color3, float **normal3);
void ctr_frame_free(float *depth);

/* Device-buffer form: outputs are DEVICE pointers on the scene's device (e.g.
 * torch tensors' data_ptr()), the launch is asynchronous on `hip_stream`
 * (a hipStream_t, NULL = default stream).  No sync inside; no allocation
 * except on the first launch of a shape (tile-order buffers, see "Tile scheduling" below): make that launch
 * before capturing the call into a HIP graph.  d_counters: optional device pointer to 16×uint64
 * ([0] ray_count, [1] max_depth bits, rest reserved); the call ACCUMULATES into it
 * (add / max), so zero it before the first launch of a frame.  A scene handle keeps one
 * internal reduction scratch: launches of the SAME handle that use counters must be on one
 * stream at a time. */
int ctr_render_device(ctr_scene *scene, float fudge, int bounces, const ctr_rows *rows,
                      void *d_depth, void *d_color3, void *d_normal3,
                      void *d_counters, void *hip_stream);

/* Batch form: a sequence of frames of the same scene (a camera path) in ONE launch, so that a
 * rank that owns only 1/N of each frame's rows still fills the chip.  Cameras are uploaded once
 * with ctr_scene_set_cameras (all must share width/height; camera 0 replaces the scene's own);
 * frame f of the batch uses camera first_frame+f and writes its rows at pixel offset
 * f*frame_stride_px of the three buffers (same compact row layout as ctr_render_device).
 * part_stride != 0 rotates the row part from frame to frame: frame f renders part
 * (rows->part + f*part_stride) % rows->n_parts — over a batch every rank then sees every row
 * block, which evens out ranks whose blocks differ in cost.  frame_stride_px must then hold the
 * largest part. */
int ctr_scene_set_cameras(ctr_scene *scene, const ctr_camera *cams, uint32_t n_cams);
int ctr_render_device_batch(ctr_scene *scene, float fudge, int bounces, const ctr_rows *rows,
                            uint32_t first_frame, uint32_t n_frames, uint64_t frame_stride_px,
                            uint32_t part_stride, void *d_depth, void *d_color3, void *d_normal3,
                            void *d_counters, void *hip_stream);

/* ---- one frame over several GPUs of a node ---------------------------
 * The reference renders on one device (inc/kernel.hpp:86-130); this is the same boundary for N devices of
 * ONE process: the scene is replicated (ctr_multi_create uploads it to every listed device), device d
 * renders the interleaved row blocks {b : b mod N == d} of `block_rows` rows (0 = 8) into a compact buffer,
 * the N-1 compact buffers reach device 0 (= devices[0]) in ONE grouped RCCL send/recv over xGMI, a HIP
 * kernel re-interleaves them into the row-major frame — straight into the caller's buffers when they are
 * page-locked (whole rows, no D2H afterwards), else into a device frame that one D2H delivers (same layout and
 * semantics as ctr_render).  N = 1 needs no RCCL and is ctr_render's own host delivery.
 * librccl.so is loaded on demand; a group that lists the same device more than once (rehearsal on a
 * one-GPU box), or a box without RCCL, moves the parts with hipMemcpyPeerAsync — ctr_multi_transport tells
 * which: "single", "rccl" or "peer-copy".  Results are bitwise those of ctr_render on one device.
 * stats: kernel_ms = the slowest device's kernel, ray_count = sum, max_depth = max, total_ms = wall time. */
#define CTR_MULTI_MAX_DEVICES 16
typedef struct ctr_multi ctr_multi;
int ctr_multi_create(const ctr_scene_desc *desc, const int *devices, int n_devices, ctr_multi **out);
void ctr_multi_destroy(ctr_multi *group);
int ctr_multi_devices(const ctr_multi *group);
const char *ctr_multi_transport(const ctr_multi *group);
int ctr_multi_size(const ctr_multi *group, uint64_t *w, uint64_t *h);
int ctr_multi_set_size(ctr_multi *group, uint64_t w, uint64_t h);
int ctr_multi_set_variant(ctr_multi *group, uint32_t variant_bits);
int ctr_render_multi(ctr_multi *group, float fudge, int bounces, uint64_t block_rows,
                     float *depth, float *color3, float *normal3, ctr_render_stats *stats);
/* The re-interleave step on its own, for callers that gather the parts themselves (one process per GPU with
 * torch.distributed / RCCL: bench.py): part p's compact buffers (device pointers on the current device) hold
 * the rows {y : (y / block_rows) % n_parts == p} of a w x h frame in increasing y; they are copied into the
 * row-major frame buffers.  Asynchronous on hip_stream. */
typedef struct ctr_reint_part { const void *d_depth, *d_color3, *d_normal3; } ctr_reint_part;
int ctr_reinterleave_device(const ctr_reint_part *parts, uint32_t n_parts, uint64_t block_rows, uint64_t w, uint64_t h,
                            void *d_depth, void *d_color3, void *d_normal3, void *hip_stream);
/* The same frame-per-call work as a two-deep pipeline (round 3): ctr_multi_submit queues one frame into the caller's
 * buffers and returns at once; ctr_multi_wait blocks until the OLDEST queued frame is complete and reports its
 * statistics.  At most two frames may be in flight (a third submit fails with CTR_E_INVALID until one has been waited
 * for); each device starts the kernel of frame k+1 as soon as its kernel of frame k is done, while frame k's compact
 * buffers are gathered on device 0, re-interleaved there and copied out — the gather (58.7 MB per device at 4096 x 4096
 * over 8 devices) and the 470 MB re-interleave hide under the next frame's rendering instead of following it.
 * Destinations must stay valid and untouched until the matching wait; page-locked ones (ctr_frame_alloc) make the final
 * copy asynchronous — a pageable destination still works, but its copy blocks the submitting thread.  A group of ONE
 * device whose destination is page-locked (delivered by the render kernel) or pageable renders inside submit
 * (synchronously, through ctr_render).  ctr_render_multi(...) == ctr_multi_submit(...) + ctr_multi_wait(...). */
int ctr_multi_submit(ctr_multi *group, float fudge, int bounces, uint64_t block_rows, float *depth, float *color3,
                     float *normal3);
int ctr_multi_wait(ctr_multi *group, ctr_render_stats *stats);
/* kernel ms of every device in the frame waited for last (load balance) */
int ctr_multi_kernel_ms(ctr_multi *group, double *ms_per_device, int capacity);

/* Kernel variant selection (tuning / ablation; default picks the fastest
 * variant that is exact for the scene).  Bits: */
#define CTR_VAR_AUTO 0u
#define CTR_VAR_NO_PREFILTER 2u   /* run the exact Cramer test on every triangle */
#define CTR_VAR_NO_ANYHIT 4u      /* never use the any-hit shadow early-out */
#define CTR_VAR_NO_CLUSTER 8u     /* walk meshes linearly instead of through their BVH */
#define CTR_VAR_EXACT_POW 32u     /* exact specular term: pow() in f64 (<=1 ulp of glibc powf), IEEE half-vector normalisation */
#define CTR_VAR_STATS 16u         /* diagnostic build: print wave-level work counters to stderr */
#define CTR_VAR_NO_REORDER 256u   /* always dispatch tiles in image order (see below) */
#define CTR_VAR_IMAGE_ORDER_FIRST 2048u /* the first launch of a shape dispatches its tiles in image order instead of centre-out */
#define CTR_VAR_NO_DIRECT 1024u   /* ctr_render / ctr_render_multi: page-locked destinations get device buffers + DMA instead of delivery by the kernels */
#define CTR_VAR_MERGE 4096u       /* scenes with 2..255 meshes: walk ONE tree over the triangles of all meshes instead of a tree over the meshes' boxes and
                                   * then each mesh's own tree.  Same results; measured SLOWER on every shipped config (profiles/r04/exp_merged_tree_ab.txt), so opt-in */
#define CTR_VAR_IGNORE_TRANSPARENT 8192u /* ctr_render / ctr_render_uv: the cast of kernel.hpp:52 — the one depth, normal (and uv) come from — is made with
                                   * ray_cast's ignore_transparent = true (inc/ray_cast.hpp:30,39-40): objects whose material is transparent
                                   * (transparency >= 1e-6, default_schema.hpp:334) do not exist for it.  Colour is unchanged: ray_color's own casts
                                   * pass false (shading.hpp:32,123).  No caller of the reference passes true; this is the branch, restated. */
#define CTR_VAR_NO_OCC6 512u      /* never pick the build compiled for 6 waves per SIMD (chosen for scenes with >= 1000 mesh triangles) */
/* Tile scheduling: every launch records what each 8x8 tile cost, and the next launch of the same
 * shape (image size, rows, frame count) on the same scene handle dispatches the expensive tiles
 * first, which removes the tail of slow waves at the end of a frame.  Results do not depend on the
 * order.  The first launch of a shape has no costs to go by: image order, or — large frames of scenes with >= 1000 mesh
 * triangles — blocks of tiles from the image centre outwards (CTR_VAR_IMAGE_ORDER_FIRST: always image order).  Every
 * launch under CTR_VAR_NO_REORDER uses image order.
 * Consequence: launches on ONE scene handle must be ordered by the caller (one stream at a time).
 * The first launch of a shape allocates the (small) cost/order buffers with hipMalloc: make that
 * launch before capturing ctr_render_device into a HIP graph, or capture under CTR_VAR_NO_REORDER. */
int ctr_set_variant(ctr_scene *scene, uint32_t variant_bits);
/* Self-test: the kernel's short-cut for n = sqrtf(x), 1.0f / n (both IEEE, correctly rounded; used where a vector
 * is normalised) against the library versions, over every float mantissa and both exponent parities at the ends and
 * the middle of the range the short-cut accepts.  n_mismatch must come back 0. */
int ctr_selftest_exact_math(uint64_t *n_mismatch);
/* Diagnostic: the 16 counter words of the last ctr_render on this handle.  [0] ray_count, [1] max-depth
 * bits; under CTR_VAR_STATS wave-level work: [4] casts (wave trips) [5] BVH nodes visited [6] triangle
 * prefilters [7] exact triangle tests [8] mesh entries [9] lanes active per cast (sum) [10] lanes whose ray
 * meets a child box of a visited node (sum) [11] lanes inside the leaf's box per prefilter (sum) [12] lanes
 * per exact test (sum): [10]/(64*[5]) etc. say how many of a wave's 64 lanes the wave-level work served;
 * [13] / [14]: per mesh entry the largest number of node visits / triangle tests any ONE lane had a use for, summed —
 * what a walk by every lane for itself would take in wave steps (the bound on a per-lane walk). */
int ctr_last_counters(ctr_scene *scene, uint64_t *out16);
/* Diagnostic: how many of a wave's 64 lanes are alive, trip by trip, in the CTR_VAR_STATS launches since the last reset
 * (one lane = one pixel for the pixel's whole life; the reference's recursion, inc/shading.hpp:126-150, is what makes
 * lanes finish at different times).  96 words: [0..15] lanes casting a radiance ray at recursion depth d (0 = primary),
 * summed over the trips; [16..31] lanes casting a shadow ray for a hit at depth d; [32..47] / [48..63] trips in which at
 * least one lane casts such a ray; [64..71] trips by number of live lanes (1-8, 9-16, ... 57-64); [72] trips; [73] live
 * lanes summed over the trips; [74] waves; [75] lanes inside the image summed over the waves; [76] trips whose live lanes
 * cast for more than one (kind, depth); [77] wave casts the merged walk (one tree over all meshes' triangles) handed back
 * to the two-level walk, [78] wave casts that went through the merged walk; [80] shadow wave casts that reach the mesh phase, [81] those whose
 * live lanes all start on a plane or sphere, [82] those of [81] with no occluder found among the meshes, [83] those of [80] with none.  Process-wide (one table per device code object), not per scene handle. */
int ctr_debug_lane_stats(uint64_t *out96, int reset);
/* ctr_render plus a FOURTH output: the texture coordinates ray_cast hands back for the primary hit (its tex_coords,
 * inc/ray_cast.hpp:47 — triangle::uv_for, plane::uv_for, the sphere's atan2 / asin pair, a mesh's (hit.x, hit.y);
 * inc/default_schema.hpp:37-46,138-139,169-178,246-249), uv2 = 2 floats per pixel, row-major like depth, (0, 0) on a
 * miss.  The reference computes them for every cast and uses them nowhere (its only material ignores them,
 * :326-340); a material that reads them would start from here.  The frame leaves through device
 * buffers and copies (no delivery by the kernel); depth / colour / normal are bit-identical to ctr_render's. */
int ctr_render_uv(ctr_scene *scene, float fudge, int bounces, const ctr_rows *rows, float *depth, float *color3,
                  float *normal3, float *uv2, ctr_render_stats *stats);
/* Test hook: the next launch on this handle gets a dispatch order whose second half names no tile, i.e. it renders
 * half the tiles and leaves the rest untouched.  A ctr_render that delivers the frame itself then fails with
 * CTR_E_DELIVERY; the handle stays usable (tests/test_gpu_parity.py). */
int ctr_debug_poison_next_order(ctr_scene *scene);
/* Diagnostic: what each 8x8 tile of the last launch cost (shader-clock ticks / 64; tile index = frame-major,
 * then row-major over the launch's tile grid).  n_tiles receives the count; out may be NULL. */
int ctr_tile_costs(ctr_scene *scene, uint32_t *out, uint64_t capacity, uint64_t *n_tiles);

/* Algorithmic bytes: 56·n_objects per ray_cast + 48·n_tri for
 * every mesh whose AABB the ray hits + 28 B per pixel, as the reference's flat
 * traversal streams them; measured by a counting launch of the same kernel. */
int ctr_algorithmic_bytes(ctr_scene *scene, float fudge, int bounces, const ctr_rows *rows,
                          uint64_t *bytes, uint64_t *ray_count);

#ifdef __cplusplus
}
#endif
#endif /* CUTRACE_AMD_H */
