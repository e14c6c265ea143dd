// render_kernel.hip — the per-pixel ray-cast + shading kernel for gfx950 (MI355X).
//
// What it replaces (reference, file:line):
//   render_kernel            inc/kernel.hpp:35-60
//   ray_cast                 inc/ray_cast.hpp:29-55
//   shadow_intensity/phong/ray_color   inc/shading.hpp:22-154
//   triangle/mesh/plane/sphere::intersect, lights, material, cam::get_ray
//                            inc/default_schema.hpp:22-399
//
// Design (MI355X-first, not a translation; the measurements behind each choice are in DESIGN.md):
//  * one lane = one pixel, one wave = an 8x8 pixel tile, one wave per workgroup.  The 64 rays of
//    a wave are spatially coherent and the scene is tiny and read-only, so EVERY scene read in the
//    hot loops has a wave-uniform address: planes, objects, BVH nodes and triangles are fetched
//    with scalar loads (one 64-byte record = one s_load_dwordx16) into SGPRs and feed the VALU as
//    scalar operands — no VGPRs, no LDS bandwidth, no bank conflicts.  (Measured, scripts/valu_issue.hip:
//    a VALU instruction with an SGPR operand issues at half rate, 4.1 cycles per wave64 instruction
//    against 2.2 — still the cheapest way to get a wave-uniform record to the lanes; a packed
//    instruction with an SGPR-PAIR operand does two such operations in the same 4.1 cycles, which is
//    why nodes, planes and triangles store their coordinates as aligned pairs.)
//  * the recursion of ray_color (template depth `bounces`) becomes an explicit per-lane state
//    machine: every trip of the outer loop performs exactly ONE nearest-hit cast for every live
//    lane, whatever that lane needs it for (primary, reflection, pass-through or a shadow-loop
//    iteration).  The expensive part (plane / object / BVH / triangle loops) is always executed by
//    a converged wave; only the cheap continuation logic diverges.  Suspended activations live in
//    an LDS stack [frame][field][lane] (4 dwords per frame); no scratch memory in the default build.
//  * hot-loop conditions are 64-bit lane masks (v_cmp into SGPR pairs, combined on the scalar
//    unit), not per-lane booleans.
//  * per mesh: the reference's AABB test decided with 1-ulp reciprocals except on borderline lanes;
//    a four-wide BVH (a node holds its four children's boxes, wave-uniform stack in the lanes of one
//    VGPR) walked by the whole wave, two child boxes per packed FMA; triangles whose plane contains an
//    eye or a light are additionally handed to every lane (guard records: the one regime where culling
//    could differ from the linear walk, ctr_api.cpp refresh_linear_meshes); per triangle a
//    conservative FMA prefilter for the whole wave (barycentrics, then — only if a lane survives —
//    the ray parameter), then the reference's exact Cramer/determinant test
//    (default_schema.hpp:57-78) in the reference's operation order for surviving lanes, its three
//    IEEE divisions performed only where a 1-ulp reciprocal cannot decide.  Every shortcut acts
//    only on "certainly" (margins above the rounding of both evaluations), so results are identical
//    to the reference's linear walk (ties broken by file order).
//  * shadow rays stop at the first occluder when no material is transparent.
//  * the duplicated primary cast (kernel.hpp:52 + shading.hpp:123) is done once.
//  * tiles are dispatched expensive-first from the costs the previous launch recorded (after_render),
//    which removes the tail of slow waves at the end of a frame.
//  * arguments that only the continuation or the start of a cast needs are re-read from the kernarg
//    segment instead of living in SGPRs across the inner loops (no spill reloads by v_readlane); the
//    build avoids SLP vectorisation (compiler-made v_pk_* cost more in shuffles than they save).
//  * two builds of the default variant: 85 VGPRs = 5 waves per SIMD, and (KV_OCC6) 80 VGPRs = 6 waves, picked for scenes
//    with >= 1000 mesh triangles: five cold dwords of shading state live in one 1280-byte LDS granule behind the wave's
//    stack instead (PARK), mode and stack depth share a register, lane-derived addresses are recomputed from v_mbcnt.
//    No instantiation uses scratch memory.
//
// Numerics: compiled with -ffp-contract=off; +,-,*,/ and sqrt are IEEE correctly rounded on gfx950,
// so every geometric quantity (depth, hit, normal, which object is hit) is bit-identical to the
// reference's headers COMPILED FOR THE HOST — std::min/max select semantics for the unqualified min/max
// of the device code, no FMA contraction.  That, not a CUDA build (fminf/fmaxf-like overloads, --fmad),
// is the parity target; profiles/r02/cuda_minmax_gap.txt counts what the difference touches.
// The specular term — the half vector's normalisation and pow() — uses
// v_rsq_f32 and exp2(e*log2(x)) in f32 by default (colour within 3e-6 of the reference) or IEEE
// sqrt/division and f64 pow rounded once (CTR_VAR_EXACT_POW, bit-identical to glibc powf on every
// tested pixel); it only feeds the colour.  Texture coordinates (atan2/asin, uv_for)
// are never produced: the only material type ignores them (default_schema.hpp:326-340).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "cutrace_amd.h"
#include "scene_device.h"
#include "bvh.h"

#define CADDR __attribute__((address_space(4)))
#define BALLOT(p) __builtin_amdgcn_ballot_w64(p)
#define HIDE_UNIFORM(x) asm volatile("" : "+v"(x))  /* make a wave-uniform index look per-lane: vector load */
#define PIN3(a, b, c) asm volatile("" : "+v"(a), "+v"(b), "+v"(c))  /* opaque to LICM / speculation */
#define INVB(m) __builtin_amdgcn_inverse_ballot_w64(m)   /* wave-uniform lane mask -> per-lane predicate */
#define FCMP(a, b, pred) __builtin_amdgcn_fcmpf((a), (b), (pred))  /* v_cmp straight into a lane mask */
enum { FC_OEQ = 1, FC_OGT = 2, FC_OGE = 3, FC_OLT = 4, FC_OLE = 5 };  /* LLVM FCmp predicate numbers */
#define TL_NONE 0x7FFFFFFFu                                        /* "no more top-level items" */

// v_writelane_b32 with a wave-uniform value and lane select: this clang has no __builtin for it, so the
// LLVM intrinsic is bound by name (the compiler then routes the lane select through M0 by itself)
extern "C" __device__ uint32_t ctr_writelane(uint32_t value, uint32_t lane, uint32_t old) __asm("llvm.amdgcn.writelane.i32");

namespace {

#if defined(CTR_PROFILE)
__device__ unsigned int g_prof[128];  // executions of the CTR_MARK segments, summed over the waves of every launch
#endif

// ---- how many lanes are alive (KV_STATS builds only; read by ctr_debug_lane_stats) ----
// A lane is one pixel for the pixel's whole life, and every trip of the kernel's loop casts one ray for every lane that
// still needs one — so lanes whose pixel is finished, or whose recursion is shallower than their neighbours', idle while
// the wave goes on (shading.hpp:126-150 is what diverges).  Counted per trip, split by what the lane casts for and how
// deep its recursion is:
//   [0..15]  lanes casting a radiance ray at recursion depth d (d = 0: the primary ray), summed over the trips
//   [16..31] lanes casting a shadow ray for a hit at depth d
//   [32..47] trips in which at least one lane casts a radiance ray at depth d;   [48..63] the same for shadow rays
//   [64..71] trips by the number of live lanes: 1-8, 9-16, ..., 57-64
//   [72] trips, [73] live lanes summed over the trips, [74] waves, [75] lanes inside the image summed over the waves,
//   [76] trips in which the live lanes cast for more than one (kind, depth)
//   [80] shadow wave casts that reach the mesh phase with live lanes, [81] those whose live lanes all start on a plane / sphere (not on a
//        mesh or triangle), [82] those of [81] in which no lane met an occluder in the mesh phase — what a perfect per-light occluder-distance
//        map could skip (built, measured and removed in round 4: those are the cheap casts, profiles/r04/exp_occluder_map.txt) —, [83] those of [80] in which no lane met an occluder in the mesh phase, whatever the receivers
//   [77] casts the merged walk handed back to the two-level walk (a mesh's AABB test failed for a lane the walk had decided,
//        or a mesh's nearest valid t equalled min_t), [78] casts that went through the merged walk
__device__ unsigned long long g_lane_stats[96];

// ---- execution profile of the source's straight-line segments (scripts/dynamic_mix.py) ----
// CTR_MARK(n) opens segment n.  -DCTR_MARKS: a comment in the ISA, from which the segment's instructions are counted;
// -DCTR_PROFILE: a counter of how often a wave runs the segment — one global atomic add by lane 0 with EXEC forced
// to that lane and restored, so a segment inside divergent code counts once per wave that reaches it, whichever
// lanes are live; otherwise nothing.
#if defined(CTR_PROFILE)
#define CTR_MARK(n)                                                                                                    \
do {                                                                                                                 \
  unsigned long long pm_x;                                                                                           \
  uint32_t pm_a, pm_b;                                                                                               \
  asm volatile("s_mov_b64 %0, exec\n\ts_mov_b64 exec, 1\n\tv_mov_b32 %1, 0\n\tv_mov_b32 %2, 1\n\t"                    \
               "global_atomic_add %1, %2, %3 offset:%4\n\ts_mov_b64 exec, %0"                                         \
               : "=&s"(pm_x), "=&v"(pm_a), "=&v"(pm_b)                                                               \
               : "s"(g_prof), "n"(4 * (n))                                                                        \
               : "memory");                                                                                          \
} while (0)
#elif defined(CTR_MARKS)
#define CTR_MARK(n) asm volatile("; CTR_MARK %0" ::"n"(n))
#else
#define CTR_MARK(n)
#endif

#ifndef CTR_TW
#define CTR_TW 8
#endif
#ifndef CTR_TH
#define CTR_TH 8
#endif
#ifndef CTR_WAVES_PER_WG
#define CTR_WAVES_PER_WG 1
#endif
#ifndef CTR_MIN_WAVES_EU
#define CTR_MIN_WAVES_EU 4
#endif
constexpr int TW = CTR_TW, TH = CTR_TH;  // pixel tile of one wave (TW*TH == 64)
static_assert(TW * TH == 64, "one wave = one TW x TH tile");
constexpr int WAVES_PER_WG = CTR_WAVES_PER_WG;
constexpr int WG_THREADS = 64 * WAVES_PER_WG;

struct V3 { float x, y, z; };
typedef float float2_ __attribute__((ext_vector_type(2)));
template <int N> struct SiteTag { static constexpr int value = N; };  // which inlined copy of a lambda (CTR_MARK ids)
#define SITE(n) SiteTag<n>()
// d = s * v.{lo|hi} - k.{lo|hi} for both halves of the SGPR pair s: v_pk_fma_f32 with op_sel choosing
// which half of the VGPR pairs v and k is broadcast (vsel/ksel: 0 = low, 1 = high), k negated
// d = s * v.{lo|hi} (both halves of the SGPR pair s times ONE broadcast half of the VGPR pair v)
#define PKMULB(d, s, v, vsel)                                                                                   \
  asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0," #vsel "] op_sel_hi:[1," #vsel "]" : "=v"(d) : "s"(s), "v"(v))
// d = s * v.{lo|hi} + acc (acc: a full pair)
#define PKFMAB(d, s, v, vsel, acc)                                                                              \
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #vsel ",0] op_sel_hi:[1," #vsel ",1]" : "=v"(d) : "s"(s), "v"(v), "v"(acc))
#define PKFMA(d, s, v, vsel, k, ksel)                                                                          \
  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0," #vsel "," #ksel "] op_sel_hi:[1," #vsel "," #ksel "] "          \
      "neg_lo:[0,0,1] neg_hi:[0,0,1]"                                                                          \
      : "=v"(d)                                                                                                \
      : "s"(s), "v"(v), "v"(k))
// max over the axes of min(t1, t2) / min over the axes of max(t1, t2); a NaN operand is ignored by
// v_min/v_max (the other operand wins), i.e. that axis' constraint drops out as in the scalar version
__device__ __forceinline__ float slab_lo(float ax, float bx, float ay, float by, float az, float bz) {
  float x, y, z, r;
  asm("v_min_f32 %0, %1, %2" : "=v"(x) : "v"(ax), "v"(bx));
  asm("v_min_f32 %0, %1, %2" : "=v"(y) : "v"(ay), "v"(by));
  asm("v_min_f32 %0, %1, %2" : "=v"(z) : "v"(az), "v"(bz));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
__device__ __forceinline__ float slab_hi(float ax, float bx, float ay, float by, float az, float bz) {
  float x, y, z, r;
  asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(ax), "v"(bx));
  asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(ay), "v"(by));
  asm("v_max_f32 %0, %1, %2" : "=v"(z) : "v"(az), "v"(bz));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}
// the same with one more bound folded in: max(entry distances, mn) / min(exit distances, mx)
__device__ __forceinline__ float slab_lo4(float ax, float bx, float ay, float by, float az, float bz, float mn) {
  float x, y, z, r;
  asm("v_min_f32 %0, %1, %2" : "=v"(x) : "v"(ax), "v"(bx));
  asm("v_min_f32 %0, %1, %2" : "=v"(y) : "v"(ay), "v"(by));
  asm("v_min_f32 %0, %1, %2" : "=v"(z) : "v"(az), "v"(bz));
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(mn));
  return r;
}
__device__ __forceinline__ float slab_hi4(float ax, float bx, float ay, float by, float az, float bz, float mx) {
  float x, y, z, r;
  asm("v_max_f32 %0, %1, %2" : "=v"(x) : "v"(ax), "v"(bx));
  asm("v_max_f32 %0, %1, %2" : "=v"(y) : "v"(ay), "v"(by));
  asm("v_max_f32 %0, %1, %2" : "=v"(z) : "v"(az), "v"(bz));
  asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(r), "v"(mx));
  return r;
}
// two adjacent floats of a wave-uniform record as one aligned SGPR pair
__device__ __forceinline__ float2_ ldpair(const CADDR float (&p)[2]) { return *(const CADDR float2_ *)p; }
__device__ __forceinline__ float2_ ldpair2(const CADDR float *p) { return *(const CADDR float2_ *)p; }

// one triangle record as register values (SGPRs: the record is wave-uniform)
struct TriR {
  float2_ ab0, ab1, ab2, pxy;
  float pz;
  uint32_t orig;
  float nx, ny, nz, ke, ke2;
};
__device__ __forceinline__ TriR load_tri(const CADDR DTri &T) {
  TriR r;
  r.ab0 = ldpair(T.ab[0]); r.ab1 = ldpair(T.ab[1]); r.ab2 = ldpair(T.ab[2]);
  r.pxy = ldpair2(&T.px); r.pz = T.pz; r.orig = T.orig;
  r.nx = T.nx; r.ny = T.ny; r.nz = T.nz; r.ke = T.ke; r.ke2 = T.ke2;
  return r;
}

// ---- inc/vector.hpp, same operation order -------------------------------------
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 vadd(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 vsub(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 vscale(V3 a, float f) { return mk(f * a.x, f * a.y, f * a.z); }
__device__ __forceinline__ V3 vmul(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ float vdot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(V3 a, V3 o) {
  return mk(a.y * o.z - a.z * o.y, a.z * o.x - a.x * o.z, a.x * o.y - a.y * o.x);
}
__device__ __forceinline__ float vnorm(V3 a) { return sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
__device__ __forceinline__ V3 vnormalized(V3 a) { return vscale(a, 1.0f / vnorm(a)); }
// The same two IEEE results — n = sqrtf(x) and 1.0f / n, correctly rounded — for the whole wave at about half
// the instructions when every active lane's x lies in [2^-80, 2^80].  hipcc's correctly rounded sqrtf / division
// wrap the arithmetic below in denormal scaling (v_div_scale, a pre-multiply by 2^32) and special-case fix-ups
// (v_div_fixup, v_cmp_class) that cannot trigger in that range; what remains is
//   sqrt: s = v_sqrt_f32(x) (1 ulp), then the neighbour whose residual x - s'.s changes sign
//   1/n:  y = v_rcp_f32(n), one Newton step, then two residual corrections q += (1 - n q) y
// in the same operations and order, hence the same bits (checked exhaustively over all 2^24 mantissa / exponent-
// parity patterns by ctr_selftest_exact_math, tests/test_gpu_parity.py).  Any lane outside the range (a zero
// vector, inf, NaN): the whole wave takes the library path.
__device__ __forceinline__ void norm_and_inverse(float x, float &n, float &inv) {
  if (BALLOT(!(x >= 0x1p-80f && x <= 0x1p80f)) != 0ull) {
    CTR_MARK(90);  // library square root and division
    n = sqrtf(x);
    inv = 1.0f / n;
    return;
  }
  CTR_MARK(91);
  const float s0 = __builtin_amdgcn_sqrtf(x);
  const float sm = __uint_as_float(__float_as_uint(s0) - 1u), sp = __uint_as_float(__float_as_uint(s0) + 1u);
  const float rm = __builtin_fmaf(-sm, s0, x), rp = __builtin_fmaf(-sp, s0, x);
  float s = (rm <= 0.0f) ? sm : s0;
  s = (rp > 0.0f) ? sp : s;
  n = s;
  float y = __builtin_amdgcn_rcpf(s);
  const float e = __builtin_fmaf(-s, y, 1.0f);
  y = __builtin_fmaf(e, y, y);
  float q = y;                                   // 1.0f * y
  const float r0 = __builtin_fmaf(-s, q, 1.0f);
  q = __builtin_fmaf(r0, y, q);
  const float r1 = __builtin_fmaf(-s, q, 1.0f);
  inv = __builtin_fmaf(r1, y, q);
}
// a.normalized() and a.norm() together (vector.hpp:77-92), through norm_and_inverse
__device__ __forceinline__ V3 vnormalized_n(V3 a, float &n) {
  float inv;
  norm_and_inverse(a.x * a.x + a.y * a.y + a.z * a.z, n, inv);
  return vscale(a, inv);
}
// matrix::determinant, vector.hpp:218-224 (columns c0,c1,c2)
__device__ __forceinline__ float det3(V3 c0, V3 c1, V3 c2) {
  float a = c0.x, b = c1.x, c = c2.x, d = c0.y, e = c1.y, f = c2.y, g = c0.z, h = c1.z, i = c2.z;
  return a * e * i + b * f * g + c * d * h - c * e * g - a * f * h - b * d * i;
}
// std::min / std::max as the host-compiled reference binds the unqualified calls
__device__ __forceinline__ float smin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float smax(float a, float b) { return (a < b) ? b : a; }

// A suspended ray_color activation (shading.hpp:116-154) lives in LDS, never in scratch:
//   hot  fields 0..3 : rgb so far (3), material index | stage << 30 (stage 1 = waiting for the
//                      reflection child, 2 = for the pass-through child; the material record says
//                      how reflective / translucent the surface is)
//   cold fields 4..9 : incoming->start + distance*incoming->dir (3), incoming->dir (3); only
//                      materials that BOTH reflect and transmit need them (A.nf == 10)
// Layout [wave][frame][field][lane]: lane-contiguous dwords, so a push/pop is conflict-free
// whatever each lane's own stack depth is.  bounces_left of a frame is implied by its depth
// (every push decrements it): bl = bounces - depth.
enum { F_R = 0, F_G, F_B, F_MAT, F_PX, F_PY, F_PZ, F_DX, F_DY, F_DZ };  // F_MAT: material index | stage << 30

struct KArgs {
  // hot block, the first 32 bytes: what every cast needs before it reaches a mesh — fetched with ONE s_load_dwordx8 at
  // the head of the trip instead of five dependent scalar loads spread over it (each a round trip the wave waits for)
  const CADDR DPlanePair *planes;
  uint32_t n_plane_recs, n_axis_recs, n_oloop, n_mesh, tlas_root, has_mesh;
  // cold block, the next 48 bytes: what the continuation needs, fetched together at its start
  const CADDR DObj *objs;      // every object in scene order (hit records)
  const CADDR float *gnorm;
  const CADDR DLight *lights;
  const CADDR DMat *mats;
  uint32_t n_light;
  int bounces;
  uint32_t n_obj, tlas_root2;  // tlas_root2: the top-level tree over the meshes when tlas_root (hot block) names the merged tree
  const CADDR DObj *oloop;     // spheres and stand-alone triangles (sequential loop)
  const CADDR DObj *meshes;    // meshes with >= 1 triangle, in top-level-BVH leaf order
  uint32_t tlas_begin;
  float tl_mn[3], tl_mx[3];    // box of all meshes (margin of the top-level walk)
  const CADDR DTri *tris;
  const CADDR DNode *nodes;    // top-level tree over the meshes (two-wide nodes)
  const CADDR DNode4 *nodes4;  // per-mesh trees (four-wide nodes)
  const CADDR DCam *cams;      // one camera per frame of the batch (all w x h)
  uint32_t w, h;
  uint32_t first_frame, n_frames;
  uint64_t frame_stride_px;    // pixels between consecutive frames in the output buffers
  DRows rows;
  float fudge;
  uint32_t nf;        // LDS dwords per stack frame: 4, or 10 when some material reflects AND transmits
  uint32_t frames;    // LDS stack frames per lane: the recursion depth that can be reached (bounces, or 1 when no
                      // material reflects or transmits: ray_color then never recurses)
  const CADDR uint32_t *order;  // dispatch slot -> wave (tile) index, or null = identity
  uint32_t *cost;               // per wave (tile): shader-clock ticks it took, or null
  // "Host delivery" (below): null group_done = the outputs are written in place
  float *host_depth, *host_color, *host_normal;
  uint32_t *group_done;
  float *uv_out;      // KV_UV: texture coordinates of the primary hit, 2 floats per pixel
};

static_assert(offsetof(KArgs, planes) == 0 && offsetof(KArgs, has_mesh) == 28, "KArgs: the hot block is the first eight dwords");
static_assert(offsetof(KArgs, objs) == 32 && offsetof(KArgs, mats) == 56 && offsetof(KArgs, bounces) == 68, "KArgs: the cold block follows");

// The kernel's whole argument block as it lies in the kernarg segment: the output pointers are read from there where a
// pixel is written (the first cast's depth and normal, the colour when the pixel is finished) instead of living in
// eight SGPRs across every cast — where the compiler parked them in VGPR lanes and fetched them back after each cast's
// mesh walk (eight v_readlane per cast for pointers two casts in thirty use).
struct KParams {
  KArgs A;
  float *depth_out, *color_out, *normal_out;
  unsigned long long *counters;
};
static_assert(sizeof(KArgs) % 8 == 0 && offsetof(KParams, depth_out) == sizeof(KArgs), "KParams mirrors the kernel's parameter list");

// ---- Host delivery ----
// ctr_render hands the kernel page-locked HOST buffers.  Storing the pixels there tile by tile works (the memory is
// device-visible) but reaches PCIe as 32- and 96-byte runs: 43 GB/s for the stores alone, where one DMA of the frame
// makes 56 (scripts/pcie_store.hip).  So the tiles go to a tile-major staging area in device memory, and the wave
// that completes a GROUP of 64/TW horizontally adjacent tiles (64 x TH pixels; a counter per group) copies the
// group into the host buffers in runs of 256 / 768 bytes — the DMA's rate, but spread over the whole launch instead
// of after it.  The order the tiles are dispatched in keeps the tiles of a group together (order_block).
constexpr uint32_t GROUP_TILES = 64 / TW;
#ifndef CTR_CHEAP_FIRST_PCT
#define CTR_CHEAP_FIRST_PCT 25u  // order_block_groups: share of the groups, the cheapest, dispatched before the dear ones
#endif

// What a lane casts its next ray for, and how deep its recursion stack is, share ONE register (a VGPR less to carry
// through the cast loops): low 16 bits = stack depth; bit 31 = a shadow-loop cast, bit 30 = the pixel is finished,
// neither = a radiance cast.  Every test is a single compare.
constexpr uint32_t MSP_SHADOW = 0x80000000u, MSP_DONE = 0x40000000u;
#define MSP_ACTIVE(m) ((int)(m) < 0x40000000)          /* radiance or shadow */
#define MSP_IS_SHADOW(m) ((int)(m) < 0)
#define MSP_IS_RADIANCE(m) ((uint32_t)(m) < 0x40000000u)
#define MSP_DEPTH(m) ((int)((m) & 0xFFFFu))
enum { ACT_NONE = 0, ACT_LIGHT = 1, ACT_BOUNCE = 2, ACT_UNWIND = 3 };

template <uint32_t KV>
__global__ __launch_bounds__(WG_THREADS, (KV & KV_OCC6) ? 6 : CTR_MIN_WAVES_EU) void render_kernel(KArgs A, float *__restrict__ depth_out,
                                                            float *__restrict__ color_out,
                                                            float *__restrict__ normal_out,
                                                            unsigned long long *__restrict__ counters) {
  constexpr bool PREFILTER = (KV & KV_PREFILTER) != 0;
  constexpr bool ANYHIT = (KV & KV_ANYHIT) != 0;
  constexpr bool COUNT = (KV & KV_COUNT) != 0;
  constexpr bool BVH = (KV & KV_BVH) != 0;
#ifdef CTR_WAVELOG
  constexpr bool STATS = true;
#else
  constexpr bool STATS = (KV & KV_STATS) != 0;
#endif
  constexpr bool FASTPOW = (KV & KV_FASTPOW) != 0;
  constexpr bool HOSTOUT = (KV & KV_HOSTOUT) != 0;  // "Host delivery"
  constexpr bool UV = (KV & KV_UV) != 0;            // ray_cast's tex_coords of the primary cast as a fourth output
  // ray_cast's ninth argument (ray_cast.hpp:30,39-40: `if (ignore_transparent && is_transparent(mat)) continue;`).  Every caller
  // of the reference passes false; with IGNTR the cast of kernel.hpp:52 — the one depth, normal and tex_coords come from — is
  // made with true, as a trip of its own before ray_color's first cast (shading.hpp:123, false), which it otherwise shares
  constexpr bool IGNTR = (KV & KV_IGNTR) != 0;
  constexpr bool MERGE = (KV & KV_MERGE) != 0;      // "merged walk": the top-level item may be the pseudo mesh over all meshes' triangles
  static_assert(!MERGE || BVH, "the merged tree is a BVH walk");
  // wave-level work counters (STATS build only): [0] casts, [1] BVH nodes visited, [2] triangle
  // prefilters, [3] exact tests, [4] mesh entries (AABB ballot != 0), [5] sum of active lanes per cast,
  // and how many of the 64 lanes had a use for the wave-level work: [6] lanes whose ray meets one of the
  // visited node's child boxes, [7] lanes inside the leaf's box at a prefilter, [8] lanes in an exact test
  // and what a walk by every lane for itself would take, counted in steps of the whole wave: per mesh entry the LARGEST
  // number of [9] node visits and [10] triangle tests any one lane has a use for (the bound on what a per-lane walk
  // below some depth could save: DESIGN.md "Per-lane walk")
  unsigned long long st[11] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#ifdef CTR_TIMING
  // diagnostic build only: shader-clock stamps per wave -> shards[4..13] = {cast setup, planes, object
  // loop, top-level walk + mesh AABB, mesh entry setup, BVH walk without leaves, leaves, radiance
  // continuation, rest of the continuation, whole wave}
  unsigned long long tm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long tm_start = __builtin_readcyclecounter();
#define TSTAMP(v) const unsigned long long v = __builtin_readcyclecounter()
#define TACC(i, a, b) tm[i] += (b) - (a)
#else
#define TSTAMP(v)
#define TACC(i, a, b)
#endif

#ifdef CTR_WAVELOG
  const unsigned long long wl_start = __builtin_amdgcn_s_memrealtime();  // 100 MHz
#endif
  CTR_MARK(0);  // wave prologue
  const uint32_t w = A.w, h = A.h;
  const uint32_t lane = threadIdx.x & 63u;
  // wave-uniform by construction; with one wave per workgroup (the shipped build) a compile-time zero, so that no LDS base
  // offset derived from it has to be carried (parked in a VGPR lane and fetched back every cast) through the kernel
  const uint32_t wave_in_wg = WAVES_PER_WG == 1 ? 0u : __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const uint32_t slot = (blockIdx.x * WAVES_PER_WG) + wave_in_wg;
  const uint32_t tiles_x = (w + TW - 1) / TW;
  const uint32_t tiles_y = (A.rows.n_rows + TH - 1) / TH;
  const uint32_t tiles_frame = tiles_x * tiles_y;
  if (slot >= tiles_frame * A.n_frames) return;  // whole wave exits together
  // Dispatch order.  Waves differ 10x in cost and the hardware hands them out in blockIdx order, so
  // with tiles in image order the last quarter of a frame is a tail of a few slow waves on an
  // otherwise empty GPU.  Every launch records what each tile cost; the next launch of the same
  // shape starts the expensive tiles first (longest-processing-time-first list scheduling).
  const unsigned long long t_wave0 = __builtin_readcyclecounter();
  const uint32_t wave = A.order ? A.order[slot] : slot;
  if (wave >= tiles_frame * A.n_frames) return;  // cannot happen with a valid order; never write out of bounds
  // batch of frames (a camera path): frame-major waves, one camera per frame (wave-uniform)
  const uint32_t frame = wave / tiles_frame;
  const uint32_t tile = wave - frame * tiles_frame;
  const CADDR DCam &cam = A.cams[A.first_frame + frame];
  const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
  const uint32_t x_id = tx * TW + (lane % TW);
  const uint32_t k_row = ty * TH + (lane / TW);  // local (compact) row
  // local row -> global image row (interleaved row blocks, see ctr_rows).  In a batch the part a
  // rank renders may rotate from frame to frame (part_stride), which balances ranks whose row
  // blocks differ in cost.
  uint32_t y_id;
  if (A.rows.n_parts <= 1) y_id = A.rows.row_begin + k_row;
  else {
    const uint32_t part_f = (A.rows.part + frame * A.rows.part_stride) % A.rows.n_parts;
    const uint32_t b0 = A.rows.row_begin / A.rows.block_rows;
    const uint32_t first = b0 + ((part_f + A.rows.n_parts - (b0 % A.rows.n_parts)) % A.rows.n_parts);
    const uint32_t j = k_row / A.rows.block_rows;
    y_id = (first + j * A.rows.n_parts) * A.rows.block_rows + (k_row % A.rows.block_rows);
  }
  const bool in_image = x_id < w && k_row < A.rows.n_rows && y_id < A.rows.row_end;
  // kernel.hpp:54 (compact buffer).  Recomputed from the lane id at its three uses rather than kept
  // in two VGPRs for the whole wave: registers, not instructions, are what this kernel is short of.
  const uint32_t tile_px0 = ty * TH * w + tx * TW;  // wave-uniform
  // a pixel's three floats: in place, or — staged — written THROUGH to memory (agent scope), because the wave that
  // copies the group to the host usually runs on another XCD, whose L2 is not coherent with this one
  auto store3 = [&](float *__restrict__ out, size_t px_id, V3 v) {
    if (HOSTOUT) {
      __hip_atomic_store(out + 3 * px_id + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(out + 3 * px_id + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(out + 3 * px_id + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      out[3 * px_id + 0] = v.x;
      out[3 * px_id + 1] = v.y;
      out[3 * px_id + 2] = v.z;
    }
  };
  // the lane's number from the hardware (two v_mbcnt) wherever a per-lane address is formed inside the loop: neither the
  // lane id nor an address derived from it is then carried through the cast loops in a register
  auto lane_now = [&]() -> uint32_t { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); };
  auto px_index = [&]() -> size_t {
    const uint32_t l = lane_now();  // recomputed here: hoisted out of the loop the addresses would be spilled to scratch
    if (HOSTOUT) return (size_t)wave * 64u + l;  // tile-major staging ("Host delivery")
    return (size_t)frame * A.frame_stride_px + (size_t)(tile_px0 + (l / TW) * w + (l % TW));
  };

  // ---- cam::get_ray, default_schema.hpp:376-386 ----
  V3 ro, rd;
  {
    const float fw = (float)w, fh = (float)h;
    const float aspect = fw / fh;
    const V3 right = mk(cam.right[0], cam.right[1], cam.right[2]);
    const V3 up = mk(cam.up[0], cam.up[1], cam.up[2]);
    const V3 fwd = mk(cam.forward[0], cam.forward[1], cam.forward[2]);
    V3 x_v = vscale(right, (((float)x_id / fw) - 0.5f) * aspect);
    V3 y_v = vscale(up, 0.5f - ((float)y_id / fh));
    ro = mk(cam.pos[0], cam.pos[1], cam.pos[2]);
    rd = vnormalized(vadd(vadd(x_v, y_v), fwd));
  }
  const float ambient = cam.ambient;

  // ---- per-lane state machine ----
  uint32_t msp = in_image ? 0u : MSP_DONE;  // mode | stack depth (MSP_*); bounces left for the current activation = bounces - depth
  bool first_trip = true;   // wave-uniform: every in-image lane shades its primary hit in the first trip
  uint32_t wave_dbits = 0u; // wave-uniform: max finite depth bits of the tile (kernel.hpp:120-125)
  float min_t = A.fudge;
  extern __shared__ float lds_stack[];
#define STK(frame, field) (lds_stack + (size_t)wave_in_wg * A.frames * A.nf * 64 + lane_now())[((frame) * A.nf + (field)) * 64]
  V3 in_d = rd;             // direction of the radiance ray being shaded ("incoming"); its start is `ro` until the hit
  V3 nn = mk(0, 0, 0), pos = mk(0, 0, 0);  // (the hit point itself lives in `ro` from the hit on: it is
                                           //  the origin of every shadow ray of that hit)
  V3 in_dn = mk(0, 0, 0);   // incoming->dir.normalized() (shading.hpp:90,131; default_schema.hpp:245)
  // PARK (the 6-waves-per-SIMD build): five dwords of shading state that no cast reads — in_dn and two components of
  // nn, written once per shaded hit, read once per light — live in LDS behind the wave's stack instead of in VGPRs.
  // They are what the register allocator spilled to scratch at 80 VGPRs (write-back traffic all frame long: 1.9x the
  // compulsory bytes on the bunny frame, 3.5x with 64 000 triangles); five, because LDS is handed out in 1280-byte
  // granules and 5 KB of stack (bounces 5) + 1280 B is the most that still lets 24 waves share a CU
  // (profiles/r03/lds_granule.txt).
  constexpr bool PARK = (KV & KV_OCC6) != 0;
#define PRK(i) (lds_stack + (size_t)WAVES_PER_WG * A.frames * A.nf * 64 + (size_t)wave_in_wg * 5 * 64 + lane_now())[(i) * 64]
  auto set_in_dn = [&](V3 v) { if (PARK) { PRK(0) = v.x; PRK(1) = v.y; PRK(2) = v.z; } else in_dn = v; };
  auto get_in_dn = [&]() -> V3 { return PARK ? mk(PRK(0), PRK(1), PRK(2)) : in_dn; };
  auto set_nn = [&](V3 v) { if (PARK) { PRK(3) = v.x; PRK(4) = v.y; nn.z = v.z; } else nn = v; };
  auto get_nn = [&]() -> V3 { return PARK ? mk(PRK(3), PRK(4), nn.z) : nn; };
  V3 fin = mk(0, 0, 0);     // phong accumulator ("final")
  float light_dist = 0.f, intensity = 0.f;
  uint32_t mat_i = 0, li = 0;
  V3 out_rgb = mk(0, 0, 0);
  bool recv_mesh = false;   // STATS only: the surface this lane's shadow rays start on is a mesh / triangle
  // casts of the whole wave, counted on the scalar unit; the duplicated primary cast (kernel.hpp:52) counts too
  // (IGNTR: the kernel.hpp:52 cast is a trip of its own and is counted there)
  unsigned long long n_casts = IGNTR ? 0ull : (unsigned long long)__builtin_popcountll(BALLOT(in_image));
  unsigned long long n_aabb_tris = 0;

  // Cold kernel arguments — what only the continuation needs (hit records, lights, materials) — are read from
  // the kernarg segment where they are used instead of living in SGPRs across the object / BVH / triangle loops:
  // the loops need every scalar register, and an argument kept through them is spilled to a VGPR lane and comes
  // back by v_readlane (a half-rate VALU instruction per dword, every trip); an s_load from the scalar cache costs
  // no VALU slot.  The pointer is made opaque once per trip so that the loads stay inside the trip.
  const CADDR KArgs *AK = (const CADDR KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  while (BALLOT(MSP_ACTIVE(msp)) != 0ull) {
    asm volatile("" : "+s"(AK));
    typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
    const u32x8 hot = *(const CADDR u32x8 *)AK;  // KArgs' hot block
    const CADDR DPlanePair *const k_planes = (const CADDR DPlanePair *)(((uint64_t)hot[1] << 32) | hot[0]);
    const uint32_t k_plane_recs = hot[2], k_axis_recs = hot[3], k_n_oloop = hot[4], k_mesh = hot[5], k_tlas_root = hot[6];
    const bool k_has_mesh = hot[7] != 0u;
    TSTAMP(t_trip0);
    CTR_MARK(1);  // trip head: cast set-up
    typedef unsigned long long mask_t;
    const bool active = MSP_ACTIVE(msp);
    const bool shadow_cast = MSP_IS_SHADOW(msp);
    const mask_t active_m = BALLOT(active);
    const mask_t shadow_m = BALLOT(shadow_cast);  // (taken here, straight from the compare: see alive_m below)
    n_casts += (unsigned long long)__builtin_popcountll(active_m);
    if (STATS) {
      st[0]++; st[5] += __builtin_popcountll(active_m);
      uint32_t kinds = 0;
      for (uint32_t d = 0; d < 16u; d++) {
        const mask_t r_m = BALLOT(MSP_IS_RADIANCE(msp) && (uint32_t)MSP_DEPTH(msp) == d);
        const mask_t s_m = BALLOT(shadow_cast && (uint32_t)MSP_DEPTH(msp) == d);
        kinds += (r_m != 0ull) + (s_m != 0ull);
        if (lane == 0) {
          if (r_m) { atomicAdd(&g_lane_stats[d], (unsigned long long)__builtin_popcountll(r_m)); atomicAdd(&g_lane_stats[32 + d], 1ull); }
          if (s_m) { atomicAdd(&g_lane_stats[16 + d], (unsigned long long)__builtin_popcountll(s_m)); atomicAdd(&g_lane_stats[48 + d], 1ull); }
        }
      }
      const mask_t img_m = BALLOT(in_image);
      if (lane == 0) {
        const uint32_t n_live = (uint32_t)__builtin_popcountll(active_m);
        atomicAdd(&g_lane_stats[64 + ((n_live - 1u) >> 3)], 1ull);
        atomicAdd(&g_lane_stats[72], 1ull);
        atomicAdd(&g_lane_stats[73], (unsigned long long)n_live);
        if (kinds > 1u) atomicAdd(&g_lane_stats[76], 1ull);
        if (first_trip) { atomicAdd(&g_lane_stats[74], 1ull); atomicAdd(&g_lane_stats[75], (unsigned long long)__builtin_popcountll(img_m)); }
      }
    }

    // =====================================================================
    // ray_cast (ray_cast.hpp:29-55): nearest hit of (ro, rd) over all objects
    // =====================================================================
    float best = INFINITY;
    int bobj = -1, btri = -1;
    // lanes still searching (any-hit mode retires occluded lanes), as a lane MASK: a per-lane bool that crosses blocks
    // is kept as a mask by the compiler anyway, and every BALLOT of it costs two vector instructions to get it back
    mask_t alive_m = active_m;
    V3 rinv = mk(0, 0, 0);     // exact 1/dir (IEEE divisions), computed lazily: see the mesh branch
    bool have_rinv = false;    // wave-uniform
    V3 ria = mk(0, 0, 0);
    float ria_big = 0.f;
    float cmax = 0.f;
    if (k_has_mesh) {
      CTR_MARK(2);
      // 1-ulp reciprocals, clamped to +-1e30: an axis-parallel ray (d = 0 -> inf) then gives huge
      // FINITE slab distances with the right signs instead of inf - inf = NaN
      ria = mk(fminf(fmaxf(__builtin_amdgcn_rcpf(rd.x), -1e30f), 1e30f),
               fminf(fmaxf(__builtin_amdgcn_rcpf(rd.y), -1e30f), 1e30f),
               fminf(fmaxf(__builtin_amdgcn_rcpf(rd.z), -1e30f), 1e30f));
      ria_big = fmaxf(fmaxf(fabsf(ria.x), fabsf(ria.y)), fabsf(ria.z));
      cmax = fmaxf(fmaxf(fabsf(rd.x), fabsf(rd.y)), fabsf(rd.z));
    }

    TSTAMP(t_loop0);
    TACC(0, t_trip0, t_loop0);
    CTR_MARK(3);  // planes: set-up
    const bool anyhit_cast = ANYHIT && shadow_cast;
    const bool ign_now = IGNTR && first_trip;  // wave-uniform: this trip is the kernel.hpp:52 cast with ignore_transparent = true
    // ---- planes: plane::intersect, default_schema.hpp:189-201, all in lane masks ----
    {
      mask_t live_m = alive_m;
      // t0 = num/den is an IEEE division, and it is only worth doing where the quotient can matter.
      // Classification WITHOUT dividing (and without a reciprocal): with num' = num * sign(den) and
      // den' = |den| the quotient is num'/den', so
      //   t0 <= min_t for certain  if  num' < (min_t - 2^-21|min_t|) * den' - tiny
      //   t0 >  lim   for certain  if  num' > (lim   + 2^-21|lim|)   * den' + tiny
      // (2^-21 covers the roundings of the two products and of the division itself; tiny = 1e-37 covers
      // products that underflow; a NaN compares false = "needed"; den = 0 gives num' <> -+tiny, and the
      // quotient is then infinite or NaN, which plane::intersect rejects as not finite.)
      // lim = the light for a deciding shadow ray, else the nearest hit so far.
      const float mt_lo = min_t - fabsf(min_t) * 0x1p-21f;
      float lhv = anyhit_cast ? light_dist : best;
      lhv = lhv + fabsf(lhv) * 0x1p-21f;
      float tiny = 1e-37f;
      asm volatile("" : "+v"(tiny));  // keep it in a VGPR: a literal operand would halve the FMA's issue rate
      // one plane against the lanes in need_m; returns false when every lane has retired (any-hit)
      auto plane_exact = [&](uint32_t pidx, float num, float den, mask_t need_m, auto site) -> bool {
        CTR_MARK(110 + decltype(site)::value);  // a plane some lane needs the quotient of
        bool retire = false;
        if (INVB(need_m)) {
          const float t0 = num / den;
          // isfinite && min_t <= t0 (plane) && t0 > min_t (ray_cast.hpp:43)  ==  finite && t0 > min_t
          if (__builtin_isfinite(t0) && t0 > min_t) {
            // ray_cast.hpp:43: strict <, first object in scene order wins ties
            if (t0 < best || (t0 == best && (int)pidx < bobj)) {
              best = t0; bobj = (int)pidx; btri = -1;
              if (!anyhit_cast) lhv = t0 + fabsf(t0) * 0x1p-21f;
            }
            if (anyhit_cast && t0 < light_dist) retire = true;
          }
        }
        if (ANYHIT) {
          live_m &= ~BALLOT(retire);
          if (live_m == 0ull) return false;
        }
        return true;
      };
      auto plane_test = [&](uint32_t pidx, float num, float den, auto site) -> bool {  // (site: which inlined copy, for CTR_MARK)
        CTR_MARK(100 + decltype(site)::value);  // one plane's classification
        const float nump = __uint_as_float(__float_as_uint(num) ^ (__float_as_uint(den) & 0x80000000u));
        const float denp = fabsf(den);
        const mask_t need_m = live_m & ~(FCMP(nump, __builtin_fmaf(mt_lo, denp, -tiny), FC_OLT) |
                                         FCMP(nump, __builtin_fmaf(lhv, denp, tiny), FC_OGT));
        if (need_m == 0ull) return true;
        return plane_exact(pidx, num, den, need_m, site);
      };
      // Two planes per 64-byte record with interleaved coordinates: the numerators (point - origin).normal
      // and denominators dir.normal of BOTH planes come from packed f32 multiplies and adds — the
      // reference's operations in the reference's order (vector.hpp dot: x*x' + y*y' + z*z' left to
      // right), two at a time.  Three records (six planes) are fetched and evaluated per trip, so a box
      // room costs ONE scalar-load round trip per cast instead of three dependent ones.
      const float2_ rox = {ro.x, ro.x}, roy = {ro.y, ro.y}, roz = {ro.z, ro.z};
      const float2_ rdx = {rd.x, rd.x}, rdy = {rd.y, rd.y}, rdz = {rd.z, rd.z};
      auto num_den = [&](const CADDR DPlanePair &P, float2_ &num, float2_ &den) {
        const float2_ nx = ldpair(P.n[0]), ny = ldpair(P.n[1]), nz = ldpair(P.n[2]);
        const float2_ dx = ldpair(P.p[0]) - rox, dy = ldpair(P.p[1]) - roy, dz = ldpair(P.p[2]) - roz;
        num = (dx * nx + dy * ny) + dz * nz;
        den = (rdx * nx + rdy * ny) + rdz * nz;
      };
      // Axis-aligned planes (the first n_axis_recs records, whole triples: two planes normal to x, two to y, two to z):
      // the reference's  dx*nx + dy*ny + dz*nz  with two of the normal's components zero is the one product that is
      // not multiplied by zero whenever that product is not zero itself (x + (+-0) = x), and a zero of either sign
      // otherwise.  A zero numerator gives t0 = +-0 or NaN and a zero denominator an infinite or NaN t0 — a miss
      // whatever the zero's sign as long as min_t > 0; so with fudge > 0 and every live lane's origin and direction
      // finite (0 x inf would be NaN in the reference) three packed instructions per record stand for thirteen.
      const uint32_t n_recs = k_plane_recs, n_axis = k_axis_recs;
      bool axis_fast = false;  // wave-uniform
      if (n_axis != 0u && A.fudge >= 1e-30f) {
        CTR_MARK(4);
        // NaN iff some component is not finite (or the sum overflows: then the general code, which is always right)
        // (the origin's components times 4: beyond 8.5e37 the product is inf, so the differences point - origin of the
        //  planes — whose points the upload bounds by 1e37 — cannot overflow either: an infinite difference times the normal's
        //  zero is a NaN in the reference, a miss, where the one-product form would report a hit)
        const float chk = __builtin_fmaf(ro.x, 4.0f, __builtin_fmaf(ro.y, 4.0f, __builtin_fmaf(ro.z, 4.0f, (rd.x + rd.y) + rd.z))) * 0.0f;
        axis_fast = (alive_m & ~FCMP(chk, 0.0f, FC_OEQ)) == 0ull;
      }
      for (uint32_t p = 0; p < n_recs;) {
        if (n_recs - p >= 3u) {
          const CADDR DPlanePair &P0 = k_planes[p], &P1 = k_planes[p + 1], &P2 = k_planes[p + 2];
          float2_ num0, den0, num1, den1, num2, den2;
          // (indices first: their loads then travel with the coordinates', one round trip for the triple)
          const uint32_t i00 = P0.index[0], i01 = P0.index[1], i10 = P1.index[0], i11 = P1.index[1], i20 = P2.index[0],
                         i21 = P2.index[1];
          if (p < n_axis && axis_fast) {
            CTR_MARK(5);  // an axis triple
            const float2_ nx = ldpair(P0.n[0]), ny = ldpair(P1.n[1]), nz = ldpair(P2.n[2]);
            num0 = (ldpair(P0.p[0]) - rox) * nx; den0 = rdx * nx;
            num1 = (ldpair(P1.p[1]) - roy) * ny; den1 = rdy * ny;
            num2 = (ldpair(P2.p[2]) - roz) * nz; den2 = rdz * nz;
          } else {
            CTR_MARK(6);  // three plane records
            num_den(P0, num0, den0);
            num_den(P1, num1, den1);
            num_den(P2, num2, den2);
          }
          p += 3;
          // (a slot without a plane: CTR_PLANE_PAD)
          // (a slot without a plane: CTR_PLANE_PAD; IGNTR: a plane whose material is transparent does not exist for this cast)
          if (i00 != CTR_PLANE_PAD && !(IGNTR && ign_now && P0.transparent[0])) { if (!plane_test(i00, num0.x, den0.x, SITE(0))) break; }
          if (i01 != CTR_PLANE_PAD && !(IGNTR && ign_now && P0.transparent[1])) { if (!plane_test(i01, num0.y, den0.y, SITE(1))) break; }
          if (i10 != CTR_PLANE_PAD && !(IGNTR && ign_now && P1.transparent[0])) { if (!plane_test(i10, num1.x, den1.x, SITE(2))) break; }
          if (i11 != CTR_PLANE_PAD && !(IGNTR && ign_now && P1.transparent[1])) { if (!plane_test(i11, num1.y, den1.y, SITE(3))) break; }
          if (i20 != CTR_PLANE_PAD && !(IGNTR && ign_now && P2.transparent[0])) { if (!plane_test(i20, num2.x, den2.x, SITE(4))) break; }
          if (i21 != CTR_PLANE_PAD && !(IGNTR && ign_now && P2.transparent[1])) { if (!plane_test(i21, num2.y, den2.y, SITE(5))) break; }
        } else {
          const CADDR DPlanePair &P0 = k_planes[p];
          CTR_MARK(7);  // one plane record
          float2_ num0, den0;
          num_den(P0, num0, den0);
          const uint32_t i00 = P0.index[0], i01 = P0.index[1];
          p += 1;
          if (!(IGNTR && ign_now && P0.transparent[0])) { if (!plane_test(i00, num0.x, den0.x, SITE(6))) break; }
          if (i01 != CTR_PLANE_PAD && !(IGNTR && ign_now && P0.transparent[1])) {
            if (!plane_test(i01, num0.y, den0.y, SITE(7))) break;
          }
        }
      }
      if (ANYHIT) alive_m = live_m;
    }
    TSTAMP(t_planes1);
    TACC(1, t_loop0, t_planes1);
    CTR_MARK(8);  // after the planes
    // ---- spheres and stand-alone triangles, scene order ----
    // sphere::intersect normalises the direction first (default_schema.hpp:227): once per cast (at the
    // first sphere the cast meets), not once per sphere
    V3 sph_d = mk(0, 0, 0);
    float sph_dd = 0.f;
    bool sph_have = false;  // wave-uniform
    for (uint32_t oi = 0; oi < k_n_oloop; ++oi) {
      if (ANYHIT) {
        if (alive_m == 0ull) break;
      }
      const CADDR DObj &O = AK->oloop[oi];
      const uint32_t i = O.index;
      const uint32_t type = O.type;
      if (IGNTR && ign_now && __float_as_uint(O.f[7]) != 0u) continue;  // ray_cast.hpp:40
      CTR_MARK(9);  // a sphere or stand-alone triangle
      bool ok = false;
      float cand = INFINITY;
      int ctri = -1;
      if (type == CTR_OBJ_SPHERE) {
        // ---- sphere::intersect, default_schema.hpp:226-251 ----
        if (!sph_have) {
          CTR_MARK(10);
          float sx_ = rd.x, sy_ = rd.y, sz_ = rd.z;
          PIN3(sx_, sy_, sz_);  // keeps the normalisation out of scenes without spheres (no hoisting)
          sph_d = vnormalized(mk(sx_, sy_, sz_));
          sph_dd = vdot(sph_d, sph_d);
          sph_have = true;
        }
        const V3 d = sph_d, c = mk(O.f[0], O.f[1], O.f[2]);
        const float R = O.f[3];
        const V3 ec = vsub(ro, c);
        const float dec = -vdot(d, ec);
        const float dd = sph_dd;
        const float sub = dec * dec - dd * (vdot(ec, ec) - R * R);
        // sub < 0: sqrt gives NaN, both roots are NaN, the sphere is missed (that IS how the reference
        // signals a miss) — so the square root and the two divisions run only if some live lane has
        // sub >= 0 or NaN (the exact reference value of sub decides, no margin involved)
        if ((alive_m & ~FCMP(sub, 0.0f, FC_OLT)) != 0ull) {
          CTR_MARK(11);  // sphere roots
          const float sq = sqrtf(sub);
          const float t0 = (dec - sq) / dd, t1 = (dec + sq) / dd;
          const bool t0v = __builtin_isfinite(t0) && min_t <= t0, t1v = __builtin_isfinite(t1) && min_t <= t1;
          ok = t0v || t1v;
          cand = (t0v && t1v) ? smin(t0, t1) : (t0v ? t0 : t1);
        }
      } else {
        // ---- stand-alone triangle, default_schema.hpp:57-78 ----
        const CADDR DTri &T = A.tris[O.tri_begin];
        const V3 a = mk(T.ab[0][0], T.ab[1][0], T.ab[2][0]), b = mk(T.ab[0][1], T.ab[1][1], T.ab[2][1]);
        const V3 d = mk(T.px - ro.x, T.py - ro.y, T.pz - ro.z);
        const float alpha = det3(a, b, rd);
        const float beta = det3(d, b, rd) / alpha;
        const float gamma = det3(a, d, rd) / alpha;
        const float t0 = det3(a, b, d) / alpha;
        ok = beta >= 0 && gamma >= 0 && beta + gamma <= 1 && __builtin_isfinite(t0) && min_t <= t0;
        cand = t0;
        ctri = (int)O.tri_begin;
      }
      CTR_MARK(12);  // object loop: candidate merge
      // ray_cast.hpp:43 — strict <, first object in scene order wins ties
      if (INVB(alive_m) && ok && cand > min_t && (cand < best || (cand == best && (int)i < bobj))) {
        best = cand;
        bobj = (int)i;
        btri = ctri;
      }
      if (ANYHIT) alive_m &= ~BALLOT(shadow_cast && ok && cand > min_t && cand < light_dist);
    }
    // ---- meshes, reached through a top-level BVH over their boxes (bvh.h layout, one mesh per leaf)
    //      so that a cast only looks at meshes some lane's ray can touch.  Visiting order does not
    //      matter: the winner is the lexicographic minimum of (t, scene index). ----
    TSTAMP(t_oloop1);
    TACC(2, t_planes1, t_oloop1);
    CTR_MARK(13);
    const mask_t st_shadow_entry = (STATS && ANYHIT) ? (alive_m & shadow_m) : 0ull;
    const mask_t st_recv_mesh = STATS ? BALLOT(recv_mesh) : 0ull;
    if (k_mesh != 0u) {
      uint32_t t_pend = k_tlas_root;             // next top-level item: inner node or mesh leaf
      uint32_t t_stack_v = 0, t_sp = 0;        // wave-uniform stack in the lanes of one VGPR
      // ---- mesh::bound_intersects, default_schema.hpp:99-114, for the lanes in `lanes` -> the lanes that pass ----
      // The reference's slab test needs three IEEE divisions (1/dir) per ray.  With 1-ulp
      // reciprocals every t is within 3 ulp of the exact one and tmin/tmax are 1-Lipschitz in
      // the t's, so lanes whose tmin/tmax differ by more than dl = 2^-20 * max|t| are decided
      // without them; only borderline lanes (or NaN/inf: axis-parallel rays) take the exact path.
      // (site 0: a mesh of the top-level walk; site 1: the merged walk's verification)
      auto box_test = [&](const CADDR DObj &R, mask_t lanes, auto site) -> mask_t {
        constexpr int MK = decltype(site)::value ? 83 : 18;
        (void)MK;
        const float a1x = (R.f[0] - ro.x) * ria.x, a2x = (R.f[3] - ro.x) * ria.x;
        const float a1y = (R.f[1] - ro.y) * ria.y, a2y = (R.f[4] - ro.y) * ria.y;
        const float a1z = (R.f[2] - ro.z) * ria.z, a2z = (R.f[5] - ro.z) * ria.z;
        const float lo_a = fmaxf(fmaxf(fmaxf(fminf(a1x, a2x), fminf(a1y, a2y)), fminf(a1z, a2z)), 0.0f);
        const float hi_a = fminf(fminf(fmaxf(a1x, a2x), fmaxf(a1y, a2y)), fmaxf(a1z, a2z));
        const float tabs = fmaxf(fmaxf(fmaxf(fabsf(a1x), fabsf(a2x)), fmaxf(fabsf(a1y), fabsf(a2y))),
                                 fmaxf(fabsf(a1z), fabsf(a2z)));
        const float dl = tabs * 0x1p-20f;
        const mask_t def_hit = FCMP(lo_a + dl, hi_a, FC_OLT);
        const mask_t def_miss = FCMP(lo_a - dl, hi_a, FC_OGT);
        // axis-parallel rays always take the exact path: the reference's inf/NaN min/max semantics
        // (0 x inf when the origin sits exactly on a box face) are not what finite arithmetic gives
        const mask_t border = lanes & (~(def_hit | def_miss) | FCMP(ria_big, 1e29f, FC_OGE));
        mask_t pass_m = lanes & def_hit & ~border;
        if (border != 0ull) {
          CTR_MARK(MK);  // exact AABB for borderline lanes
          if (!have_rinv) {  // wave-uniform: the exact reciprocals are computed at most once per cast
            CTR_MARK(MK + 1);
            float ox = rd.x, oy = rd.y, oz = rd.z;
            PIN3(ox, oy, oz);  // keeps the three IEEE divisions in this rarely-taken branch (no hoisting)
            rinv = mk(1.0f / ox, 1.0f / oy, 1.0f / oz);  // default_schema.hpp:103
            have_rinv = true;
          }
          CTR_MARK(MK + 2);
          float tmin = 0.0f, tmax = INFINITY;
          float t1 = (R.f[0] - ro.x) * rinv.x, t2 = (R.f[3] - ro.x) * rinv.x;
          tmin = smin(smax(t1, tmin), smax(t2, tmin));
          tmax = smax(smin(t1, tmax), smin(t2, tmax));
          t1 = (R.f[1] - ro.y) * rinv.y; t2 = (R.f[4] - ro.y) * rinv.y;
          tmin = smin(smax(t1, tmin), smax(t2, tmin));
          tmax = smax(smin(t1, tmax), smin(t2, tmax));
          t1 = (R.f[2] - ro.z) * rinv.z; t2 = (R.f[5] - ro.z) * rinv.z;
          tmin = smin(smax(t1, tmin), smax(t2, tmin));
          tmax = smax(smin(t1, tmax), smin(t2, tmax));
          pass_m |= border & FCMP(tmin, tmax, FC_OLE);
        }
        return pass_m;
      };
      for (;;) {
        TSTAMP(t_tl0);
        if (ANYHIT) {
          if (alive_m == 0ull) break;
        }
        // (nothing left: TL_NONE has no leaf flag, and without this test a scene's LAST mesh was followed by one more
        //  descent set-up — ~30 vector instructions per cast for nothing, found by the execution profile, scripts/dynamic_mix.py)
        if (t_pend == TL_NONE) break;
        // descend the top-level tree to the next mesh leaf
        if (!(t_pend & BVH_LEAF_FLAG)) {
          CTR_MARK(14);  // top-level descent set-up
          const mask_t lv_m = alive_m;
          const float t_lim = anyhit_cast ? light_dist : best;
          // conservative box test constants, as in the per-mesh walk below (world-space margin
          // 2^-14 x G); recomputed per descent so that nothing stays live across the mesh code
          const float gx = fmaxf(fabsf(AK->tl_mn[0] - ro.x), fabsf(AK->tl_mx[0] - ro.x));
          const float gy = fmaxf(fabsf(AK->tl_mn[1] - ro.y), fabsf(AK->tl_mx[1] - ro.y));
          const float gz = fmaxf(fabsf(AK->tl_mn[2] - ro.z), fabsf(AK->tl_mx[2] - ro.z));
          const float mw = fmaxf(fmaxf(gx, gy), gz) * 0x1p-14f;
          const V3 t_ka = mk((ro.x + mw) * ria.x, (ro.y + mw) * ria.y, (ro.z + mw) * ria.z);
          const V3 t_kb = mk((ro.x - mw) * ria.x, (ro.y - mw) * ria.y, (ro.z - mw) * ria.z);
          const uint32_t t_lead = (uint32_t)__builtin_ctzll(lv_m);
          const uint32_t t_neg =
              (((uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.x), t_lead) >> 31) |
               (((uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.y), t_lead) >> 31) << 1) |
               (((uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.z), t_lead) >> 31) << 2)) ^
              ((ANYHIT && ((shadow_m >> t_lead) & 1ull) != 0ull) ? 7u : 0u);
          while (t_pend != TL_NONE && !(t_pend & BVH_LEAF_FLAG)) {
            const CADDR DNode &N = A.nodes[AK->tlas_begin + t_pend];
            CTR_MARK(15);  // top-level node
            auto t_hits = [&](int c) -> mask_t {
              const float t1x = __builtin_fmaf(N.mn[0][c], ria.x, -t_ka.x), t2x = __builtin_fmaf(N.mx[0][c], ria.x, -t_kb.x);
              const float t1y = __builtin_fmaf(N.mn[1][c], ria.y, -t_ka.y), t2y = __builtin_fmaf(N.mx[1][c], ria.y, -t_kb.y);
              const float t1z = __builtin_fmaf(N.mn[2][c], ria.z, -t_ka.z), t2z = __builtin_fmaf(N.mx[2][c], ria.z, -t_kb.z);
              // as in the per-mesh walk: a lane takes the child unless max(entry, min_t) > min(exit, limit) for certain
              // (v_min / v_max drop a NaN operand: that axis' constraint drops out; one compare instead of three)
              const float lo = slab_lo4(t1x, t2x, t1y, t2y, t1z, t2z, min_t);
              const float hi = slab_hi4(t1x, t2x, t1y, t2y, t1z, t2z, t_lim);
              return lv_m & ~FCMP(lo, hi, FC_OGT);
            };
            // visiting order as in the per-mesh walk (speed only): along the lead ray for nearest-hit casts, against
            // it for any-hit shadow casts
            const bool t_rev = ((t_neg >> N.axis) & 1u) != 0u;
            // (both children are VALUES here, fetched with the boxes: the compiler otherwise selects the ADDRESS and loads
            //  the one it wants after the test — a second dependent round trip per node)
            uint32_t n_left = N.left, n_right = N.right;
            asm volatile("" : "+s"(n_left), "+s"(n_right));
            // The two hit masks stay lane masks and the four cases are told apart by comparing them with zero (round 4: the
            // earlier form — hit counts folded into two bits of an integer, swapped by `t_rev`, then a chain of four equality
            // tests — cost 47 scalar instructions and 15 branches per node; this one about half of that).
            const mask_t tm0 = t_hits(0), tm1 = t_hits(1);
            const mask_t mN = t_rev ? tm1 : tm0, mF = t_rev ? tm0 : tm1;   // the nearer / the farther child along the lead ray
            const uint32_t dN = t_rev ? n_right : n_left, dF = t_rev ? n_left : n_right;
            if (mN != 0ull) {
              if (mF != 0ull) {
                t_stack_v = ctr_writelane(__builtin_amdgcn_readfirstlane(dF), __builtin_amdgcn_readfirstlane(t_sp), t_stack_v);
                t_sp++;
              }
              t_pend = dN;
            } else if (mF != 0ull) {
              t_pend = dF;
            } else if (t_sp != 0u) {
              t_sp--;
              t_pend = (uint32_t)__builtin_amdgcn_readlane((int)t_stack_v, (int)t_sp);
            } else {
              t_pend = TL_NONE;
            }
          }
        }
        CTR_MARK(16);
        if (t_pend == TL_NONE) break;
        const CADDR DObj &O = AK->meshes[t_pend & 0xFFFFFFu];
        // advance first, so that `continue` below moves on to the next mesh
        if (t_sp != 0u) {
          t_sp--;
          t_pend = (uint32_t)__builtin_amdgcn_readlane((int)t_stack_v, (int)t_sp);
        } else {
          t_pend = TL_NONE;
        }
        if (IGNTR && ign_now && __float_as_uint(O.f[7]) != 0u) continue;  // ray_cast.hpp:40: a mesh with a transparent material
        CTR_MARK(17);  // a mesh: AABB test
        const uint32_t i = O.index;
        // (the whole record in one round trip: what the walk needs is requested with the box, not after its test)
        const uint32_t o_tri_begin = O.tri_begin, o_tri_count = O.tri_count, o_node_begin = O.node_begin, o_bvh_root = O.bvh_root;
        const bool merged = MERGE && i == 0xFFFFFFFFu;  // wave-uniform: the pseudo mesh (CTR_OBJ_MERGED) is no object of the scene
        bool ok = false;
        float cand = INFINITY;
        int ctri = -1;
        int obj_lane = (int)i;
        {
          // Everything in this branch works on 64-bit lane masks (v_cmp results kept in SGPRs and
          // combined on the scalar unit) instead of per-lane booleans.
          const mask_t live_m = alive_m;
          // One tree over the triangles of ALL meshes stands behind the pseudo mesh (ctr_api.cpp ctr_scene::Merged): every
          // live lane enters it; the AABB test of the mesh a lane's triangle belongs to is made after the walk, for the
          // lanes the walk decided something for ("merged walk" below).  A real mesh: the reference's test first.
          mask_t bb_m;
          if (merged) {
            // (a ray with a NaN or infinite component never meets a triangle — alpha or the numerators of
            //  default_schema.hpp:59-62 are then not finite and `isfinite(t0)` or a barycentric test fails — but the walk's
            //  box tests let a NaN through by design, so such a lane would drag its wave through the WHOLE tree on every
            //  cast; a mesh's own AABB test stops it in the two-level walk.  Reflections off zero-area triangles make
            //  them: C3-deep ran 15x longer for a handful of such waves)
            const float fin_chk = ((ro.x * 0.0f + ro.y * 0.0f) + (ro.z * 0.0f + rd.x * 0.0f)) + (rd.y * 0.0f + rd.z * 0.0f);
            // ... and a ray that misses the box of ALL meshes for certain misses every mesh's box for certain (each lies inside:
            // its slab distances lie between the union's, so its entry / exit gap is at least the union's and the rounding
            // of the reference's own test, 3 ulp of distances no larger than the union's, is far below dl).  That gate is
            // what keeps a ray whose ORIGIN is far away out of the walk — one that left the room through its open side and
            // met a wall plane a million units off: the box tests' world-space margin (2^-14 x the distance to the meshes)
            // then exceeds the room, every box test passes and the wave walks the whole tree on each such cast.
            const float a1x = (O.f[0] - ro.x) * ria.x, a2x = (O.f[3] - ro.x) * ria.x;
            const float a1y = (O.f[1] - ro.y) * ria.y, a2y = (O.f[4] - ro.y) * ria.y;
            const float a1z = (O.f[2] - ro.z) * ria.z, a2z = (O.f[5] - ro.z) * ria.z;
            const float lo_a = fmaxf(fmaxf(fmaxf(fminf(a1x, a2x), fminf(a1y, a2y)), fminf(a1z, a2z)), 0.0f);
            const float hi_a = fminf(fminf(fmaxf(a1x, a2x), fmaxf(a1y, a2y)), fmaxf(a1z, a2z));
            const float tabs = fmaxf(fmaxf(fmaxf(fabsf(a1x), fabsf(a2x)), fmaxf(fabsf(a1y), fabsf(a2y))),
                                     fmaxf(fabsf(a1z), fabsf(a2z)));
            const mask_t def_miss = FCMP(lo_a - tabs * 0x1p-20f, hi_a, FC_OGT) & ~FCMP(ria_big, 1e29f, FC_OGE);
            bb_m = live_m & FCMP(fin_chk, 0.0f, FC_OEQ) & ~def_miss;
            // A lane of those that starts more than 16 box sizes away (and still aims at the box: a shadow ray from such a hit
            // point to a light in the room) gets box margins of 0.1 % of the box and more, growing with its distance: the
            // cast goes through the two-level walk instead, where the meshes' own AABB tests — exact at any distance —
            // stand before the walk.
            const float g_far = fmaxf(fmaxf(fmaxf(fabsf(O.f[0] - ro.x), fabsf(O.f[3] - ro.x)), fmaxf(fabsf(O.f[1] - ro.y), fabsf(O.f[4] - ro.y))),
                                      fmaxf(fabsf(O.f[2] - ro.z), fabsf(O.f[5] - ro.z)));
            const float ext16 = 16.0f * fmaxf(fmaxf(O.f[3] - O.f[0], O.f[4] - O.f[1]), O.f[5] - O.f[2]);
            if ((bb_m & FCMP(g_far, ext16, FC_OGT)) != 0ull) {
              CTR_MARK(88);
              if (STATS && lane == 0) atomicAdd(&g_lane_stats[77], 1ull);
              t_pend = AK->tlas_root2;
              t_sp = 0u;
              continue;
            }
          } else bb_m = box_test(O, live_m, SITE(0));
          TSTAMP(t_bb);
          TACC(3, t_tl0, t_bb);
          CTR_MARK(21);
          if (bb_m == 0ull) continue;  // no lane of this wave needs the mesh
          CTR_MARK(22);  // mesh entered: walk set-up
          const uint32_t beg = o_tri_begin, cnt = o_tri_count;
          if (COUNT) n_aabb_tris += INVB(bb_m) ? (unsigned long long)cnt : 0ull;
          if (STATS) st[4]++;
          uint32_t pl_nodes = 0, pl_tris = 0;  // STATS: this lane's own share of the mesh entry's node visits / triangle tests
          // ---- mesh::intersect, default_schema.hpp:125-144: smallest valid t, FIRST triangle in
          //      file order on ties (strict < over file order)  ==  lexicographic min of (t, orig) ----
          float mt = INFINITY;
          uint32_t morig = 0xFFFFFFFFu;
          const bool anyhit_now = ANYHIT && shadow_cast;
          const mask_t anyhit_m = ANYHIT ? shadow_m : 0ull;
          const bool t_filter = A.fudge >= 1e-30f;  // then min_t > 0 in every cast (wave-uniform)
          // no triangle/node beyond `lim` can matter: the light for a deciding shadow ray, else the
          // nearest hit so far (other objects, then this mesh)
          float lim = anyhit_now ? light_dist : best;

          const float2_ ro_xy = {ro.x, ro.y};
          // one triangle against the lanes in `lanes_m` (wave-uniform T: SGPR operands)
          auto tri_test = [&](const TriR &T, uint32_t tri_index, mask_t lanes_m) {
            CTR_MARK(23);  // triangle: prefilter stage 1
            mask_t c_m = lanes_m;
            if (STATS) { st[2]++; st[7] += __builtin_popcountll(lanes_m); pl_tris += INVB(lanes_m) ? 1u : 0u; }
            // d = p2 - start (default_schema.hpp:58): x and y in one packed subtraction (same IEEE result)
            const float2_ dxy = T.pxy - ro_xy;
            const float dx = dxy.x, dy = dxy.y, dz = T.pz - ro.z;
            uint32_t sgn = 0;
            float sA1 = 0.f, sA2 = 0.f, absa = 0.f, dmax = 0.f, E = 0.f;
            mask_t flat_m = 0ull;
            if (PREFILTER) {
              // Conservative reject test.  Same quantities as the exact test
              // (alpha = det[a b c], A1 = det[d b c], A2 = det[a d c]) evaluated with FMAs
              // as triple products; every comparison carries a slack E that bounds both this
              // evaluation's and the reference's rounding, and a
              // NaN anywhere makes the lane a candidate.
              const float alpha = __builtin_fmaf(rd.x, T.nx, __builtin_fmaf(rd.y, T.ny, rd.z * T.nz));
              float2_ q_xy, q_z;
              q_xy.x = __builtin_fmaf(dy, rd.z, -(dz * rd.y));
              q_xy.y = __builtin_fmaf(dz, rd.x, -(dx * rd.z));
              q_z.x = __builtin_fmaf(dx, rd.y, -(dy * rd.x));
              q_z.y = 0.0f;
              // (a.q, b.q) = (A2, -A1) together: the (a, b) pairs of the record times q, three packed instructions
              float2_ m;
              PKMULB(m, T.ab0, q_xy, 0);
              PKFMAB(m, T.ab1, q_xy, 1, m);
              PKFMAB(m, T.ab2, q_z, 0, m);
              // (s A2, s A1) with s = sign(alpha): multiply by (s, -s), exact
              sgn = __float_as_uint(alpha) & 0x80000000u;
              float2_ s1;
              s1.x = __uint_as_float(sgn | 0x3f800000u);
              s1.y = 0.0f;
              float2_ sm;
              asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(sm) : "v"(m), "v"(s1));
              sA2 = sm.x;
              sA1 = sm.y;
              absa = fabsf(alpha);
              dmax = fmaxf(fmaxf(fabsf(dx), fabsf(dy)), fabsf(dz));
              // kappa * max(dmax,emax) * emax * cmax; both operands are >= 0, so the maximum is taken on the
              // bit patterns (fmaxf would first canonicalise the loaded ke2: one more VALU op per triangle)
              const uint32_t e_a = __float_as_uint(dmax * T.ke), e_b = __float_as_uint(T.ke2);
              E = __uint_as_float(e_a > e_b ? e_a : e_b) * cmax;
              // reject if  s A1 < -E  or  s A2 < -E  or  s A1 + s A2 > |alpha| + E:  ONE comparison of the smallest
              // of the three slacks (v_min3 drops a NaN operand: a lane is rejected only on a definite violation)
              float slack;
              {
                const float third = absa - (sA1 + sA2);
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(slack) : "v"(sA1), "v"(sA2), "v"(third));
              }
              const mask_t rej = FCMP(slack, -E, FC_OLT);
              flat_m = FCMP(absa, E, FC_OLE);  // alpha within rounding of 0: always a candidate
              c_m = lanes_m & (~rej | flat_m);
              if (c_m == 0ull) return;
              if (t_filter) {
                CTR_MARK(24);  // prefilter stage 2
                // Second stage, only when some lane survived the first: the ray parameter.
                // t0 = A0/alpha with A0 = det[a b d] = d.(a x b); E0 bounds the rounding of A0 the
                // way E does for the other three.  A lane whose t0 is certainly below min_t or
                // certainly beyond `lim` (the nearest hit so far / the light) cannot be affected by
                // this triangle — typically the triangle the ray starts on, and hits behind the
                // current best.
                const float A0 = __builtin_fmaf(dx, T.nx, __builtin_fmaf(dy, T.ny, dz * T.nz));
                const float sA0 = __uint_as_float(__float_as_uint(A0) ^ sgn);
                const float E0 = dmax * T.ke2;
                const float a_lo = (absa - E) * 0x1.fffep-1f, a_hi = (absa + E) * 0x1.0002p+0f;
                const mask_t near_m = FCMP(sA0 + E0, min_t * a_lo, FC_OLT);   // t0 < min_t for sure
                const mask_t far_m = FCMP(sA0 - E0, lim * a_hi, FC_OGT);      // t0 > lim for sure (inf/NaN: no)
#ifndef CTR_NO_TREJ
                c_m &= ~(near_m | far_m) | flat_m;
#endif
#ifndef CTR_NO_OCC
                if (anyhit_m != 0ull) {  // wave-uniform: lane masks must never be updated under a per-lane branch
                  CTR_MARK(25);
                  // a deciding shadow ray needs no value at all: clearly inside the triangle and
                  // clearly between min_t and the light = occluded, no exact test
                  const mask_t in_m = FCMP(sA1, E, FC_OGT) & FCMP(sA2, E, FC_OGT) & FCMP(sA1 + sA2, absa - E, FC_OLT);
                  const mask_t tin_m = FCMP(sA0 - E0, min_t * a_hi, FC_OGT) & FCMP(sA0 + E0, light_dist * a_lo, FC_OLT);
                  const mask_t occ_m = c_m & in_m & tin_m & ~flat_m & anyhit_m;
                  if (occ_m != 0ull) {
                    CTR_MARK(26);
                    // (which triangle decided the lane: the merged walk verifies that triangle's mesh afterwards)
                    if (MERGE) morig = INVB(occ_m) ? T.orig : morig;
                    bb_m &= ~occ_m;  // (retired: best / bobj are set once, where the mesh is left)
                    alive_m &= ~occ_m;
                    c_m &= ~occ_m;
                  }
                }
#endif
                CTR_MARK(27);
                if (c_m == 0ull) return;
              }
            }
            if (c_m == 0ull) return;
            if (STATS) { st[3]++; st[8] += __builtin_popcountll(c_m); }
            CTR_MARK(28);  // exact test
            bool retire = false;  // any-hit: this lane found its occluder
            if (INVB(c_m)) {
              // ---- triangle::intersect, default_schema.hpp:57-78: the four determinants in the
              //      reference's operation order ----
              const V3 a = mk(T.ab0.x, T.ab1.x, T.ab2.x), b = mk(T.ab0.y, T.ab1.y, T.ab2.y);
              const V3 d = mk(dx, dy, dz);
              const float alpha = det3(a, b, rd);
              const float A1 = det3(d, b, rd), A2 = det3(a, d, rd), A0 = det3(a, b, d);
              // The three IEEE divisions (beta, gamma, t0) decide five comparisons and deliver t0.
              // 1-ulp-reciprocal quotients are within 2^-21 relative of the exact ones, so with a
              // 2^-18 margin (and an absolute floor for zero/denormal quotients) most lanes are
              // decided without dividing; t0 itself is divided only where its value can matter.
              const float r = __builtin_amdgcn_rcpf(alpha);
              const float bq = A1 * r, gq = A2 * r, tq = A0 * r, sq = bq + gq;
              const float eb = fabsf(bq) * 0x1p-18f + 1e-30f, eg = fabsf(gq) * 0x1p-18f + 1e-30f;
              const float es = (fabsf(bq) + fabsf(gq) + 1.0f) * 0x1p-16f;
              const float et = fabsf(tq) * 0x1p-18f + 1e-30f;
              const bool def_rej = (bq < -eb) | (gq < -eg) | (sq > 1.0f + es) | (tq < min_t - et);
              const bool def_acc = (bq > eb) & (gq > eg) & (sq < 1.0f - es) & (tq > min_t + et) & (fabsf(tq) < 1e37f);
              bool acc = def_acc;
              float t0 = tq;
              bool exact_t = false;
              if (!(def_rej | def_acc)) {
                CTR_MARK(29);  // the reference's three divisions
                // borderline (or NaN/inf): the reference's own arithmetic
                const float beta = A1 / alpha, gamma = A2 / alpha;
                t0 = A0 / alpha;
                exact_t = true;
                acc = beta >= 0 && gamma >= 0 && beta + gamma <= 1 && __builtin_isfinite(t0) && min_t <= t0;
              }
              if (acc) {
                CTR_MARK(30);
                if (anyhit_now) {
                  CTR_MARK(121);
                  // A deciding shadow ray only asks whether some valid t lies in (min_t, light_dist).
                  // tq is within et of the exact t0 and already > min_t + et, so divide only when tq is
                  // within et of the light distance.
                  if (!exact_t && !(tq + et < light_dist) && !(tq - et >= light_dist)) { CTR_MARK(122); t0 = A0 / alpha; }
                  CTR_MARK(123);
                  if (t0 > min_t && t0 < light_dist) { retire = true; if (MERGE) morig = T.orig; }  // (best / bobj: where the mesh is left)
                } else {
                  CTR_MARK(124);
                  // the exact value of t0 matters only if it can beat or tie the nearest hit so far
                  if (!exact_t && !(tq - et > lim)) { CTR_MARK(125); t0 = A0 / alpha; exact_t = true; }
                  CTR_MARK(126);
                  const uint32_t orig = T.orig;
                  if (exact_t && (t0 < mt || (t0 == mt && orig < morig))) {
                    mt = t0; morig = orig;
                    lim = fminf(lim, mt);
                  }
                }
              }
            }
            CTR_MARK(31);
            if (ANYHIT) {
              const mask_t rm = BALLOT(retire);
              bb_m &= ~rm;
              alive_m &= ~rm;
            }
          };

#ifdef CTR_TIMING
          unsigned long long t_leaves = 0, t_w0 = 0;
#endif
          if (BVH) {
            // Walk of the mesh's four-wide BVH by the whole wave together (bvh.h, DNode4): one visit =
            // one 128-byte node = the boxes of FOUR children, tested with twelve v_pk_fma_f32 (two
            // children per instruction).  Leaves among the hit children are tested at once, inner
            // children wait on a wave-uniform stack kept in the lanes of ONE VGPR (v_writelane /
            // v_readlane), nearest first: the children are stored sorted along the node's order axis
            // and a wave whose lead ray points the other way takes them in reverse.
            // The box test is conservative: every box is widened in WORLD space by
            //   m = 2^-14 x (largest |coordinate difference| between the ray origin and the mesh box)
            // per axis ((mn - m - o)/d and (mx + m - o)/d, folded into the two FMA constants below), far
            // above the rounding of either this test or the reference's triangle test (DESIGN.md
            // §bvh); a NaN (0 x inf for an axis-parallel ray) drops that axis' constraint.
            const float gx = fmaxf(fabsf(O.f[0] - ro.x), fabsf(O.f[3] - ro.x));
            const float gy = fmaxf(fabsf(O.f[1] - ro.y), fabsf(O.f[4] - ro.y));
            const float gz = fmaxf(fabsf(O.f[2] - ro.z), fabsf(O.f[5] - ro.z));
            const float mw = fmaxf(fmaxf(gx, gy), gz) * 0x1p-14f;
            const V3 ka = mk((ro.x + mw) * ria.x, (ro.y + mw) * ria.y, (ro.z + mw) * ria.z);  // for box minima
            const V3 kb = mk((ro.x - mw) * ria.x, (ro.y - mw) * ria.y, (ro.z - mw) * ria.z);  // for box maxima
            const CADDR DNode4 *nodes4 = A.nodes4 + o_node_begin;
            // direction signs of the first lane that needs the mesh decide the visiting order (speed only)
            const uint32_t lead = (uint32_t)__builtin_ctzll(bb_m);
            // (readlane returns int: shift its bits as unsigned — only bits 0..2 are used below, but an arithmetic shift
            //  would smear the sign over the whole word)
            const uint32_t nb_x = (uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.x), lead) >> 31,
                           nb_y = (uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.y), lead) >> 31,
                           nb_z = (uint32_t)__builtin_amdgcn_readlane(__float_as_uint(rd.z), lead) >> 31;
            // Visiting order (speed only; any order gives the same hit).  Nearest-hit casts go near to far along the lead
            // ray — what they find prunes the rest.  Shadow casts go FAR to near: they start on a surface, whose own
            // neighbourhood is what a near-to-far walk would test first and what occludes least (bunny -2 %, the
            // 64 000-triangle mesh -8 % against near-to-far for both; profiles/r02/traversal_order.txt).
            // (only where a shadow cast stops at its first occluder; the ordered shadow loop of scenes with transparent
            //  materials is a nearest-hit cast)
            const uint32_t lead_shadow = (ANYHIT && ((shadow_m >> lead) & 1ull) != 0ull) ? 7u : 0u;
            const uint32_t neg_bits = lead_shadow ^ (nb_x | (nb_y << 1) | (nb_z << 2));
            // the nine per-ray constants of the box test, two to a register pair; PKFMA picks the half it
            // needs with op_sel, so packing costs no extra registers
            const float2_ c_rxy = {ria.x, ria.y}, c_rzk = {ria.z, ka.x}, c_kyz = {ka.y, ka.z};
            const float2_ c_bxy = {kb.x, kb.y}, c_bz = {kb.z, 0.0f};
            // children (c, c+1) of one node: six v_pk_fma_f32 give the six slab distances of both boxes;
            // a lane takes a child unless it misses for certain: max(entry, min_t) > min(exit, lim)
            // (v_min/v_max drop a NaN operand, a NaN that survives compares false -> entered)
            auto box_hits2 = [&](const auto &N, int c, mask_t &ha, mask_t &hb) {
              float2_ t1x, t1y, t1z, t2x, t2y, t2z;
              PKFMA(t1x, (float2_{N.lo[0][c], N.lo[0][c + 1]}), c_rxy, 0, c_rzk, 1);
              PKFMA(t1y, (float2_{N.lo[1][c], N.lo[1][c + 1]}), c_rxy, 1, c_kyz, 0);
              PKFMA(t1z, (float2_{N.lo[2][c], N.lo[2][c + 1]}), c_rzk, 0, c_kyz, 1);
              PKFMA(t2x, (float2_{N.hi[0][c], N.hi[0][c + 1]}), c_rxy, 0, c_bxy, 0);
              PKFMA(t2y, (float2_{N.hi[1][c], N.hi[1][c + 1]}), c_rxy, 1, c_bxy, 1);
              PKFMA(t2z, (float2_{N.hi[2][c], N.hi[2][c + 1]}), c_rzk, 0, c_bz, 0);
              // both boxes' slab arithmetic as ONE statement (the same twenty-two instructions, interleaved): between separate
              // statements the compiler pads with s_nop for hazards it cannot rule out — eleven per node visit
              float u0, u1, u2, u3, w0, w1, w2, w3;
              mask_t ma, mb;
              asm("v_min_f32 %0, %10, %13\n\tv_min_f32 %4, %16, %19\n\t"
                  "v_min_f32 %1, %11, %14\n\tv_min_f32 %5, %17, %20\n\t"
                  "v_min_f32 %2, %12, %15\n\tv_min_f32 %6, %18, %21\n\t"
                  "v_max_f32 %3, %10, %13\n\tv_max_f32 %7, %16, %19\n\t"
                  "v_max3_f32 %0, %0, %1, %2\n\tv_max3_f32 %4, %4, %5, %6\n\t"
                  "v_max_f32 %1, %11, %14\n\tv_max_f32 %5, %17, %20\n\t"
                  "v_max_f32 %2, %12, %15\n\tv_max_f32 %6, %18, %21\n\t"
                  "v_max_f32 %0, %0, %22\n\tv_max_f32 %4, %4, %22\n\t"
                  "v_min3_f32 %1, %3, %1, %2\n\tv_min3_f32 %5, %7, %5, %6\n\t"
                  "v_min_f32 %1, %1, %23\n\tv_min_f32 %5, %5, %23\n\t"
                  "v_cmp_gt_f32_e64 %8, %0, %1\n\tv_cmp_gt_f32_e64 %9, %4, %5"
                  : "=&v"(u0), "=&v"(u1), "=&v"(u2), "=&v"(u3), "=&v"(w0), "=&v"(w1), "=&v"(w2), "=&v"(w3), "=&s"(ma), "=&s"(mb)
                  : "v"(t1x.x), "v"(t1y.x), "v"(t1z.x), "v"(t2x.x), "v"(t2y.x), "v"(t2z.x),
                    "v"(t1x.y), "v"(t1y.y), "v"(t1z.y), "v"(t2x.y), "v"(t2y.y), "v"(t2z.y), "v"(min_t), "v"(lim));
              ha = bb_m & ~ma;
              hb = bb_m & ~mb;
            };
            const CADDR DTri *tb_ = A.tris + beg;   // the mesh's first record: leaves address theirs by a 32-bit byte offset from it
            auto leaf = [&](uint32_t desc, mask_t lanes) {
              TSTAMP(t_leaf0);
              CTR_MARK(80);  // a leaf's triangles
              uint32_t t_off = (desc & 0xFFFFFFu) << 6;   // byte offset of the leaf's first record from the mesh's first
              const uint32_t n_l = (desc >> 24) & 0x7Fu;
              // The leaf's first triangle is fetched as one 64-byte record, and one dword of each of the next three cache
              // lines is requested with it, into a register nobody reads ("touch"): the loads of the leaf's other triangles —
              // one dependent round trip each — then find their lines in the scalar cache.  64 000 triangles -3 %, first
              // launch -5 %; a 1 000-triangle mesh, which stays in the scalar cache anyway, +-0.5 %
              // (profiles/r03/exp_leaf_touch.txt).  The wait is part of the statement, so the unread register is dead when it
              // ends.  (ctr_api.cpp allocates 256 bytes beyond every array for requests past the last triangle.)
              // (no early exit for an empty leaf — the unused slots of a node, whose far-away point box no ray enters: it
              //  would only re-test the mesh's first triangle, which changes nothing, and the extra edge costs six register
              //  copies per leaf visit in the compiler's output)
              typedef uint32_t u32x16_ __attribute__((ext_vector_type(16)));
              u32x16_ t0;
              uint32_t touch_;
              asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dword %1, %2, %3 offset:0x40\n\ts_load_dword %1, %2, %3 offset:0x80\n\ts_load_dword %1, %2, %3 offset:0xc0\n\ts_waitcnt lgkmcnt(0)"
                           : "=&s"(t0), "=&s"(touch_) : "s"(tb_), "s"(t_off));
              TriR cur;
#define CTR_UNPACK_TRI()                                                                                                                              \
              cur.ab0.x = __uint_as_float(t0[0]); cur.ab0.y = __uint_as_float(t0[1]); cur.ab1.x = __uint_as_float(t0[2]); cur.ab1.y = __uint_as_float(t0[3]); \
              cur.ab2.x = __uint_as_float(t0[4]); cur.ab2.y = __uint_as_float(t0[5]); cur.pxy.x = __uint_as_float(t0[6]); cur.pxy.y = __uint_as_float(t0[7]); \
              cur.pz = __uint_as_float(t0[8]); cur.orig = t0[9]; cur.nx = __uint_as_float(t0[10]); cur.ny = __uint_as_float(t0[11]);                       \
              cur.nz = __uint_as_float(t0[12]); cur.ke = __uint_as_float(t0[13]); cur.ke2 = __uint_as_float(t0[14]);
              CTR_UNPACK_TRI()
              for (uint32_t k = 0;;) {
                tri_test(cur, 0u, lanes & bb_m);
                if (ANYHIT) {
                  if (bb_m == 0ull) break;
                }
                if (++k >= n_l) break;
                CTR_MARK(82);  // the leaf's next triangle
                // the leaf's next record: ONE 64-byte request at the running offset (the compiler's own form is a 64-bit address
                // and four requests for the record's fifteen words)
                t_off += 64u;
                asm volatile("s_load_dwordx16 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=&s"(t0) : "s"(tb_), "s"(t_off));
                CTR_UNPACK_TRI()
              }
#ifdef CTR_TIMING
              t_leaves += __builtin_readcyclecounter() - t_leaf0;
#endif
            };
            // push one inner child (wave-uniform value and slot: both travel as SGPRs)
            uint32_t stack_v = 0;  // lane k of this VGPR = stack slot k (at most 3 per level, BVH4_MAX_DEPTH levels)
            uint32_t sp = 0;
            auto push = [&](uint32_t d) {
              stack_v = ctr_writelane(__builtin_amdgcn_readfirstlane(d), __builtin_amdgcn_readfirstlane(sp), stack_v);
              sp++;
            };
            // node 0 is the root (a mesh that fits one leaf has a root with one child).  A mesh with guard
            // records (triangles every lane must meet whatever their box, ctr_api.cpp refresh_linear_meshes)
            // starts one node earlier, at an extra node whose children are the guard leaf and the root.
            uint32_t cur = o_bvh_root;
#ifdef CTR_TIMING
            t_w0 = __builtin_readcyclecounter();
            tm[4] += t_w0 - t_bb;
#endif
            for (;;) {
              if (STATS) st[1]++;
              CTR_MARK(32);  // BVH node
              // the node's 128 bytes as two requests at a 32-bit byte offset from the mesh's first node (one shift; the compiler's
              // own form was a 64-bit address — four scalar instructions — and five requests in three waits)
              typedef uint32_t u32x16n_ __attribute__((ext_vector_type(16)));
              u32x16n_ nq0, nq1;
              asm volatile("s_load_dwordx16 %0, %2, %3\n\ts_load_dwordx16 %1, %2, %3 offset:0x40\n\ts_waitcnt lgkmcnt(0)"
                           : "=&s"(nq0), "=&s"(nq1) : "s"(nodes4), "s"(cur << 7));
              struct NodeR { float lo[3][4], hi[3][4]; uint32_t child[4], axis; } N;
              for (int a_ = 0; a_ < 3; ++a_) for (int c_ = 0; c_ < 4; ++c_) N.lo[a_][c_] = __uint_as_float(nq0[a_ * 4 + c_]);
              for (int c_ = 0; c_ < 4; ++c_) N.hi[0][c_] = __uint_as_float(nq0[12 + c_]);
              for (int a_ = 1; a_ < 3; ++a_) for (int c_ = 0; c_ < 4; ++c_) N.hi[a_][c_] = __uint_as_float(nq1[(a_ - 1) * 4 + c_]);
              for (int c_ = 0; c_ < 4; ++c_) N.child[c_] = nq1[8 + c_];
              N.axis = nq1[12];
              mask_t h0, h1, h2, h3;
              box_hits2(N, 0, h0, h1);
              box_hits2(N, 2, h2, h3);
              // (round 4 tried NOT testing the second pair of a node that has only two children — a third of the bunny tree's
              //  nodes, unused slots come last: 28 of a visit's 56 box instructions saved there, and bunny +3.4 %, 64 000 triangles
              //  +2.7 %, C4 +3.3 % — the branch waits for the child descriptors before the second pair's arithmetic can issue)
              const uint32_t d0 = N.child[0], d1 = N.child[1], d2 = N.child[2], d3 = N.child[3], n_axis_ = N.axis;
              if (STATS) { st[6] += __builtin_popcountll(h0 | h1 | h2 | h3); pl_nodes += INVB(h0 | h1 | h2 | h3) ? 1u : 0u; }
              // children are stored sorted along the node's order axis; a wave whose lead ray points the other
              // way takes them in reverse: TWO copies of the dispatch below behind one scalar branch — selecting the four
              // descriptors and four masks into "nearest ... farthest" registers first was ten scalar instructions per visit
              // (round 4: bunny -3.4 %, 64 000 triangles -2.7 %; profiles/r04/exp_scalar_diet_ab.txt)
              const bool rev = ((neg_bits >> n_axis_) & 1u) != 0u;
              // ONE pass, farthest child first: a hit leaf is tested at once; a hit inner child becomes the node to
              // visit next and whatever was to be visited next is pushed — so the nearest inner child is visited
              // next and the stack pops nearest-first.  No per-child bookkeeping beyond two scalar tests.
              CTR_MARK(33);  // child dispatch
              uint32_t next = 0xFFFFFFFFu;
              bool done = false;
#define CTR_CHILD(e, g, id)                                         \
  CTR_MARK(id);                                                     \
  if (!done && (g) != 0ull) {                                       \
    CTR_MARK(id + 5);                                               \
    if ((e) & BVH_LEAF_FLAG) {                                      \
      leaf((e), (g));                                               \
      CTR_MARK(id + 10);                                            \
      if (ANYHIT) {                                                 \
        if (bb_m == 0ull) done = true;                              \
      }                                                             \
    } else {                                                        \
      CTR_MARK(id + 15);                                            \
      if (next != 0xFFFFFFFFu) push(next);                          \
      next = (e);                                                   \
    }                                                               \
  }
              if (rev) {
                CTR_CHILD(d0, h0, 60)
                CTR_CHILD(d1, h1, 61)
                CTR_CHILD(d2, h2, 62)
                CTR_CHILD(d3, h3, 63)
              } else {
                CTR_CHILD(d3, h3, 60)
                CTR_CHILD(d2, h2, 61)
                CTR_CHILD(d1, h1, 62)
                CTR_CHILD(d0, h0, 63)
              }
#undef CTR_CHILD
              CTR_MARK(34);
              if (done) break;
              if (next == 0xFFFFFFFFu) {
                CTR_MARK(81);
                if (sp == 0) break;
                sp--;
                next = __builtin_amdgcn_readlane(stack_v, sp);
              }
              cur = next;
            }
          } else {
            for (uint32_t k = 0; k < cnt; ++k) {
              tri_test(load_tri(A.tris[beg + k]), beg + k, bb_m);  // wave-uniform: one s_load_dwordx16
              if (ANYHIT) {
                if (bb_m == 0ull) break;
              }
            }
          }
          if (STATS) {
            for (int off = 32; off > 0; off >>= 1) {
              const uint32_t on = (uint32_t)__shfl_xor((int)pl_nodes, off), ot = (uint32_t)__shfl_xor((int)pl_tris, off);
              pl_nodes = on > pl_nodes ? on : pl_nodes;
              pl_tris = ot > pl_tris ? ot : pl_tris;
            }
            st[9] += (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_nodes);
            st[10] += (uint32_t)__builtin_amdgcn_readfirstlane((int)pl_tris);
          }
          // ---- merged walk: the AABB test of the meshes the walk found something in ----
          // The walk above went through ONE tree over the triangles of all meshes; a lane's `morig` now names the triangle
          // that decided it (nearest valid hit, or the occluder of a deciding shadow ray) as (mesh rank << 24 | file index).
          // The reference tests a mesh's box BEFORE its triangles (default_schema.hpp:126): a ray that fails it misses the
          // mesh whatever the triangle test says — possible only within rounding of the box's faces, but then the lane's
          // result, and everything the walk pruned because of it, is wrong.  So: per mesh that decided a lane (usually
          // one or two per cast) the reference's test for those lanes.  Any lane failing it, or a mesh whose nearest valid
          // t equals min_t exactly (ray_cast.hpp:43 then rejects the whole mesh, strict >), and the cast walks the
          // scene's two-level structure instead, which has the reference's form — measure-zero events, not a fast path.
          int i_lane = merged ? 0 : (int)i;   // scene index of the mesh that decided the lane
          int ctri_lane = (int)(beg + morig); // its triangle's position in the normals array (file order within its mesh)
          if (MERGE && merged) {
            CTR_MARK(86);
            if (STATS && lane == 0) atomicAdd(&g_lane_stats[78], 1ull);
            mask_t todo = BALLOT(morig != 0xFFFFFFFFu);
            mask_t bad_m = BALLOT(mt != INFINITY && !(mt > min_t));
            while (todo != 0ull) {
              const uint32_t rk = (uint32_t)__builtin_amdgcn_readlane((int)morig, (int)__builtin_ctzll(todo)) >> 24;
              const mask_t same_m = todo & BALLOT((morig >> 24) == rk);
              const CADDR DObj &R = AK->meshes[k_mesh + 1u + rk];
              const uint32_t r_index = R.index, r_beg = R.tri_begin;
              bad_m |= same_m & ~box_test(R, same_m, SITE(1));
              if (INVB(same_m)) { i_lane = (int)r_index; ctri_lane = (int)(r_beg + (morig & 0xFFFFFFu)); }
              todo &= ~same_m;
            }
            if (bad_m != 0ull) {
              CTR_MARK(87);  // this cast again, through the top-level tree and the meshes' own trees
              if (STATS && lane == 0) atomicAdd(&g_lane_stats[77], 1ull);
              if (ANYHIT) {
                alive_m |= bad_m & anyhit_m;  // (a deciding shadow ray whose occluder does not count is alive again)
                if (INVB(anyhit_m & ~alive_m)) { best = fminf(0.5f * light_dist, 1e30f); bobj = i_lane; }
              }
              t_pend = AK->tlas_root2;
              t_sp = 0u;
              continue;
            }
          }
          CTR_MARK(35);  // mesh left
          if (ANYHIT) {
            // lanes retired inside the mesh (deciding shadow rays that met an occluder): the handler only asks
            // best < light_dist.  Written here, once, rather than in the triangle test: values a loop changes are copied
            // in and out of it by the compiler (six v_mov each way per leaf visit for six such values; now three).
            // (every deciding shadow ray that has retired by now, in this mesh or before it: writing the same again for the
            //  earlier ones is harmless — the handler never asks which object it was — and saves carrying the mesh's entry
            //  mask through the walk: scalar registers are what this kernel is shortest of)
            if (INVB(anyhit_m & ~alive_m)) { best = fminf(0.5f * light_dist, 1e30f); bobj = i_lane; }
          }
          ok = mt != INFINITY;  // default_schema.hpp:143 (lanes outside the AABB never set mt)
          cand = mt;
          ctri = ctri_lane;  // gnorm is indexed by the triangle's FILE-order position within its mesh
          obj_lane = i_lane;
#ifdef CTR_TIMING
          {
            const unsigned long long t_mesh1 = __builtin_readcyclecounter();
            if (t_w0 == 0) t_w0 = t_bb;  // linear walk
            tm[5] += (t_mesh1 - t_w0) - t_leaves;
            tm[6] += t_leaves;
          }
#endif

        }
        // ray_cast.hpp:43 — strict <, first object in scene order wins ties
        if (INVB(alive_m) && ok && cand > min_t && (cand < best || (cand == best && obj_lane < bobj))) {
          best = cand;
          bobj = obj_lane;
          btri = ctri;
        }
        // (no "a deciding shadow ray with a valid mesh hit stops searching" here: in the any-hit build such a lane never
        //  records a mesh hit — it retires inside the walk, above)
      }
    }
    if (STATS && st_shadow_entry != 0ull && lane == 0) {
      const bool planes_only = (st_recv_mesh & st_shadow_entry) == 0ull, none_occluded = (st_shadow_entry & ~alive_m) == 0ull;
      atomicAdd(&g_lane_stats[80], 1ull);
      if (planes_only) atomicAdd(&g_lane_stats[81], 1ull);
      if (planes_only && none_occluded) atomicAdd(&g_lane_stats[82], 1ull);
      if (none_occluded) atomicAdd(&g_lane_stats[83], 1ull);
    }
    const bool was_hit = bobj >= 0;
    TSTAMP(t_loop1);
    CTR_MARK(36);  // continuation

    // =====================================================================
    // continuation: what did this lane cast the ray for?
    // =====================================================================
    // the continuation's kernel arguments in ONE round trip (pointer by pointer, each was a scalar load the wave waited
    // for before it could even request the record behind it)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x8 cold = *(const CADDR u32x8 *)&AK->objs;
    const u32x4 cold2 = *(const CADDR u32x4 *)&AK->n_light;
    const CADDR DObj *const k_objs = (const CADDR DObj *)(((uint64_t)cold[1] << 32) | cold[0]);
    const CADDR float *const k_gnorm = (const CADDR float *)(((uint64_t)cold[3] << 32) | cold[2]);
    const CADDR DLight *const k_lights = (const CADDR DLight *)(((uint64_t)cold[5] << 32) | cold[4]);
    const CADDR DMat *const k_mats = (const CADDR DMat *)(((uint64_t)cold[7] << 32) | cold[6]);
    const uint32_t k_n_light = cold2[0];
    const int k_bounces = (int)cold2[1];
    int act = ACT_NONE;
    float first_depth = 0.f;  // primary-hit depth of this lane, alive in the first trip only
    // the light the NEXT shadow ray goes to (type and position / direction), requested together with whatever else this
    // continuation reads first — the hit record, or the material and light being shaded — instead of in a round trip of its
    // own afterwards.  (lights[n_light] is inside the allocation: ctr_api.cpp pads every array.)
    uint32_t nl_type = 0u;
    V3 nl_v = mk(0.f, 0.f, 0.f);
    if (MSP_IS_RADIANCE(msp)) {
      const V3 ro_keep = ro;  // (IGNTR: the primary ray is cast once more, for ray_color)
      // ---- hit record: hit point, normal (per primitive), material ----
      V3 normal = mk(0, 0, 0);
      float tc_u = 0.0f, tc_v = 0.0f;  // (UV) uv{} of kernel.hpp:51 on a miss
      if (was_hit) {
        const CADDR DObj &H = k_objs[bobj];
        CTR_MARK(37);  // hit record
        // (type, material, the six floats and the triangle's normal are requested together, before the type decides
        //  which of them is used: one vector-memory round trip instead of two)
        const uint32_t ht = H.type;
        const float hf0 = H.f[0], hf1 = H.f[1], hf2 = H.f[2], hf3 = H.f[3], hf4 = H.f[4], hf5 = H.f[5];
        float gn0 = 0.f, gn1 = 0.f, gn2 = 0.f;
        if (btri >= 0) { gn0 = k_gnorm[4 * btri + 0]; gn1 = k_gnorm[4 * btri + 1]; gn2 = k_gnorm[4 * btri + 2]; }  // (a scene without triangles has no such array)
        mat_i = H.mat;
        if (STATS) recv_mesh = (ht == CTR_OBJ_MESH || ht == CTR_OBJ_TRIANGLE);
        { const CADDR DLight &L0 = k_lights[0]; nl_type = L0.type; nl_v = mk(L0.vx, L0.vy, L0.vz); }
        pos = vadd(ro, vscale(in_d, best));  // start + dist*dir (triangle/plane hit; shading.hpp:133,143)
        float unused_n0;
        const V3 hit_dn = vnormalized_n(in_d, unused_n0);
        CTR_MARK(92);
        set_in_dn(hit_dn);
        if (ht == CTR_OBJ_SPHERE) {
          CTR_MARK(38);
          // default_schema.hpp:245-246: hit uses the NORMALIZED direction
          const V3 hit = vadd(ro, vscale(hit_dn, best));
          normal = vnormalized(vsub(hit, mk(hf0, hf1, hf2)));
          ro = hit;
          if (UV && first_trip) {  // default_schema.hpp:246-249: delta = (hit - center).normalized() is the normal again
            tc_u = 0.5f + (atan2f(normal.z, normal.x) / (2.0f * (float)M_PI));
            tc_v = 0.5f + (asinf(normal.y) / (float)M_PI);
          }
        } else if (ht == CTR_OBJ_PLANE) {
          ro = pos;
          normal = mk(hf3, hf4, hf5);
          if (UV && first_trip) {  // plane::uv_for, default_schema.hpp:169-178
            const V3 ax1 = vnormalized(mk(normal.y, -normal.x, 0.0f));
            const V3 ax2 = vcross(normal, ax1);
            const V3 mod_pt = vsub(mk(hf0, hf1, hf2), pos);
            tc_u = vdot(ax1, mod_pt);
            tc_v = vdot(ax2, mod_pt);
          }
        } else {
          ro = pos;
          normal = mk(gn0, gn1, gn2);
          if (UV && first_trip) {
            if (ht == CTR_OBJ_MESH) {  // mesh::intersect, default_schema.hpp:138-139
              tc_u = pos.x;
              tc_v = pos.y;
            } else {                   // triangle::uv_for, default_schema.hpp:37-46
              const CADDR DTri &T = A.tris[btri];
              const V3 p1 = mk(hf0, hf1, hf2), p2 = mk(T.px, T.py, T.pz), p3 = mk(hf3, hf4, hf5);
              const V3 p2p1 = vsub(p2, p1), p3p1 = vsub(p3, p1), xp1 = vsub(pos, p1);
              const V3 proj_u = vscale(p2p1, vdot(xp1, p2p1) / vdot(p2p1, p2p1));
              const V3 proj_v = vscale(p3p1, vdot(xp1, p3p1) / vdot(p3p1, p3p1));
              tc_u = vnorm(proj_u) / vnorm(p2p1);
              tc_v = vnorm(proj_v) / vnorm(p3p1);
            }
          }
        }
      }
      CTR_MARK(39);
      if (first_trip) {
        // kernel.hpp:55-56 (depth = +inf, normal = 0 on a miss)
        CTR_MARK(40);
        const size_t px_id = px_index();
        float *const k_depth_out = ((const CADDR KParams *)AK)->depth_out, *const k_normal_out = ((const CADDR KParams *)AK)->normal_out;
        if (HOSTOUT) __hip_atomic_store(k_depth_out + px_id, best, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else k_depth_out[px_id] = best;
        store3(k_normal_out, px_id, normal);
        if (UV) {
          float *const uvp = AK->uv_out;
          uvp[2 * px_id + 0] = tc_u;
          uvp[2 * px_id + 1] = tc_v;
        }
        first_depth = best;
      }
      CTR_MARK(41);
      if (IGNTR && ign_now) {
        // that was the cast of kernel.hpp:52; now ray_color's own first cast (shading.hpp:123, ignore_transparent = false) of the same ray
        ro = ro_keep;
        min_t = A.fudge;
      } else if (!was_hit) {
        out_rgb = mk(0.f, 0.f, 0.f);  // shading.hpp:119
        act = ACT_UNWIND;
      } else {
        // phong prologue, shading.hpp:66-76
        CTR_MARK(42);
        const CADDR DMat &M = k_mats[mat_i];
        fin = vscale(mk(M.cx, M.cy, M.cz), ambient);
        { float unused_n; const V3 nrm_ = vnormalized_n(normal, unused_n); CTR_MARK(93); set_nn(nrm_); }
        li = 0;
        act = ACT_LIGHT;
      }
    } else if (MSP_IS_SHADOW(msp)) {
      // ---- one iteration of shadow_intensity's loop, shading.hpp:32-42 ----
      CTR_MARK(43);
      { const CADDR DLight &L1 = k_lights[li + 1u]; nl_type = L1.type; nl_v = mk(L1.vx, L1.vy, L1.vz); }
      bool done_shadow;
      float shadow_fac = 0.f;
      if (was_hit && best < light_dist) {
        // scenes without any transparency (the any-hit builds) need no material lookup here
        const float trans = ANYHIT ? 0.0f : k_mats[k_objs[bobj].mat].transparency;
        if (!ANYHIT) intensity += (1.0f - trans);
        if (ANYHIT || intensity >= 1.0f) { shadow_fac = 1.0f; done_shadow = true; }
        else {
          min_t = (float)((double)best + 1e-3);  // last_hit + 1e-3 is a double add, shading.hpp:32
          done_shadow = false;                   // cast again (same ray)
        }
      } else {
        shadow_fac = ANYHIT ? 0.0f : intensity;  // any-hit builds: the first occluder already ended the loop
        done_shadow = true;
      }
      if (done_shadow) {
        if (shadow_fac < 1.0f) {
          // shading.hpp:86-95
          CTR_MARK(44);  // shade one light
          const CADDR DMat &M = k_mats[mat_i];
          const CADDR DLight &Lg = k_lights[li];
          const V3 diffuse = mk(M.cx, M.cy, M.cz);
          const V3 specular = vscale(diffuse, M.specular);  // default_schema.hpp:328
          const V3 color = mk(Lg.cx, Lg.cy, Lg.cz);
          const V3 nd = rd;  // the shadow ray's direction IS the normalized direction to the light
          const V3 nn_ = get_nn();
          const float fd = smax(0.0f, vdot(nn_, nd));
          const V3 ld = vmul(diffuse, color);
          const V3 hsum = vadd(vscale(get_in_dn(), -1.0f), nd);
          // the half vector feeds only the specular colour term: with the fast specular path its
          // normalisation uses the 1-ulp v_rsq_f32 instead of IEEE sqrt + division
          const V3 hv = FASTPOW ? vscale(hsum, __builtin_amdgcn_rsqf(vdot(hsum, hsum))) : vnormalized(hsum);
          const float sx = smax(0.0f, vdot(nn_, hv));
          float fs;
          if (FASTPOW) {
            // exp2(e*log2(x)) on the f32 transcendental pipe: relative error ~1e-6 wherever the
            // result is not negligible; feeds only the specular colour term (<= 1e-6 absolute)
            fs = (M.phong_exp == 0.0f) ? 1.0f : __builtin_amdgcn_exp2f(M.phong_exp * __builtin_amdgcn_logf(sx));
          } else {
            fs = (float)pow((double)sx, (double)M.phong_exp);  // f64, rounded once: <= 1 ulp from glibc powf
          }
          const V3 ls = vmul(specular, color);
          fin = vadd(fin, vscale(vadd(vscale(ld, fd), vscale(ls, fs)), 1 - shadow_fac));
        }
        CTR_MARK(45);
        li++;
        act = ACT_LIGHT;
      }
    }

    TSTAMP(t_cont_mid);
    TACC(7, t_loop1, t_cont_mid);
    CTR_MARK(46);
    if (act == ACT_LIGHT) {
      if (li < k_n_light) {
        // shading.hpp:79-85: direction/distance to light li, shadow ray from *hit
        CTR_MARK(47);  // next light
        const uint32_t lg_type = nl_type;
        const V3 lg_v = nl_v;
        V3 direction;
        float distance;
        if (lg_type == CTR_LIGHT_SUN) {  // default_schema.hpp:280-283
          direction = vscale(lg_v, -1.0f);
          distance = INFINITY;
        } else {                         // default_schema.hpp:305-308
          CTR_MARK(48);  // point light
          const V3 diff = vsub(lg_v, ro);  // ro == *hit
          direction = vnormalized_n(diff, distance);
          CTR_MARK(94);
        }
        CTR_MARK(49);
        float dir_norm;
        rd = vnormalized_n(direction, dir_norm);  // shadow ray {*hit, direction.normalized()}, shading.hpp:80
        CTR_MARK(95);
        light_dist = distance * dir_norm;
        intensity = 0.0f;
        min_t = (float)(0.0 + 1e-3);  // last_hit = 0
        msp |= MSP_SHADOW;
      } else {
        act = ACT_BOUNCE;  // phong returned `fin`
      }
    }

    if (act == ACT_BOUNCE) {
      // shading.hpp:126-150 with rgb = fin
      CTR_MARK(50);
      const CADDR DMat &M = k_mats[mat_i];
      const float reflective = M.reflexivity, translucent = M.transparency;
      const int sp = MSP_DEPTH(msp);
      const bool more = sp < k_bounces;  // `if constexpr (bounces != 0)`
      const bool do_refl = more && (double)reflective >= 1e-6;
      const bool do_trans = more && (double)translucent >= 1e-6;
      if (do_refl || do_trans) {
        CTR_MARK(51);  // push a frame, cast the child
        STK(sp, F_R) = fin.x; STK(sp, F_G) = fin.y; STK(sp, F_B) = fin.z;
        STK(sp, F_MAT) = __uint_as_float(mat_i | (do_refl ? 1u << 30 : 2u << 30));  // the material says reflective / translucent
        if (do_refl && do_trans) {  // the pass-through child is cast after the reflection returns
          CTR_MARK(96);
          STK(sp, F_PX) = pos.x; STK(sp, F_PY) = pos.y; STK(sp, F_PZ) = pos.z;
          STK(sp, F_DX) = in_d.x; STK(sp, F_DY) = in_d.y; STK(sp, F_DZ) = in_d.z;
        }
        CTR_MARK(120);
        msp = (uint32_t)(sp + 1);  // one frame deeper, a radiance cast
        if (do_refl) {
          // reflect(nd, nn) = nd - (2*(nn.nd))*nn, vector.hpp:204-206
          const V3 nn_ = get_nn(), dn_ = get_in_dn();
          in_d = vsub(dn_, vscale(nn_, 2.0f * vdot(nn_, dn_)));
        }
        ro = pos; rd = in_d;
        min_t = A.fudge;
      } else {
        out_rgb = fin;
        act = ACT_UNWIND;
      }
    }

    if (act == ACT_UNWIND) {
      // return `out_rgb` to the suspended callers
      int sp = MSP_DEPTH(msp);
      for (;;) {
        CTR_MARK(52);  // unwind one level
        if (sp == 0) {
          const size_t px_id = px_index();
          store3(((const CADDR KParams *)AK)->color_out, px_id, out_rgb);
          msp = MSP_DONE;
          break;
        }
        CTR_MARK(53);
        --sp;
        V3 rgb = mk(STK(sp, F_R), STK(sp, F_G), STK(sp, F_B));
        const uint32_t f_mat = __float_as_uint(STK(sp, F_MAT));
        const CADDR DMat &FM = k_mats[f_mat & 0x3FFFFFFFu];
        const float f_transl = FM.transparency;
        if ((f_mat >> 30) == 1u) {
          CTR_MARK(97);
          rgb = vadd(rgb, vscale(out_rgb, FM.reflexivity));  // shading.hpp:138
          if ((double)f_transl >= 1e-6) {
            CTR_MARK(98);
            STK(sp, F_R) = rgb.x; STK(sp, F_G) = rgb.y; STK(sp, F_B) = rgb.z;
            STK(sp, F_MAT) = __uint_as_float((f_mat & 0x3FFFFFFFu) | (2u << 30));
            in_d = mk(STK(sp, F_DX), STK(sp, F_DY), STK(sp, F_DZ));
            ro = mk(STK(sp, F_PX), STK(sp, F_PY), STK(sp, F_PZ));
            rd = in_d;
            msp = (uint32_t)(sp + 1);  // the frame stays; a radiance cast
            min_t = A.fudge;
            break;
          }
          out_rgb = rgb;
        } else {
          CTR_MARK(99);
          // shading.hpp:148
          out_rgb = vadd(vscale(rgb, 1.0f - f_transl), vscale(out_rgb, f_transl));
        }
      }
    }
    if (first_trip) {
      // max finite depth of the tile, reduced once here instead of carrying the depth to the end
      CTR_MARK(54);
      uint32_t dbits = (__builtin_isfinite(first_depth) && first_depth > 0.f) ? __float_as_uint(first_depth) : 0u;
      for (int off = 32; off > 0; off >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)dbits, off);
        dbits = o > dbits ? o : dbits;
      }
      wave_dbits = __builtin_amdgcn_readfirstlane(dbits);
      first_trip = false;
    }
    TSTAMP(t_cont1);
    TACC(8, t_cont_mid, t_cont1);
    CTR_MARK(55);
  }
#undef STK
#undef PRK

  CTR_MARK(56);  // wave epilogue
  // (the tile's cost for the next launch's order ends here: what the group copy below costs the one wave that makes
  //  it depends on the link, not on the tile)
  const unsigned long long t_wave1 = HOSTOUT ? __builtin_readcyclecounter() : 0ull;
  if (HOSTOUT) {
    const CADDR KArgs *AE = (const CADDR KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(AE));  // the kernel arguments once more (see AK): none of this is worth an SGPR across the loop
    // "Host delivery": this tile is complete; the wave that completes its group copies the group to the host
    const uint32_t e_w = AE->w, e_tiles_x = (e_w + TW - 1) / TW, e_ty = wave / e_tiles_x, e_tx = wave - e_ty * e_tiles_x;
    const uint32_t groups_x = (e_tiles_x + GROUP_TILES - 1) / GROUP_TILES;
    const uint32_t gx = e_tx / GROUP_TILES, t0 = gx * GROUP_TILES;
    const uint32_t nt = e_tiles_x - t0 < GROUP_TILES ? e_tiles_x - t0 : GROUP_TILES;
    uint32_t *const done = AE->group_done + (e_ty * groups_x + gx);
    // The tile's pixels were written through (store3) and have arrived when the counter is incremented: an agent-
    // scope release fence here would write this XCD's whole L2 back, and the acquire below invalidate it, once per
    // wave — 3 ms per frame, measured.  The copy reads with agent-scope loads, which do not hit a stale L2 line.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    uint32_t old = 0;
    if (lane == 0) old = atomicAdd(done, 1u);
    old = __builtin_amdgcn_readfirstlane(old);
    if (old == nt - 1) {
      CTR_MARK(57);  // host delivery: copy a group
      // (the counter keeps its final value nt: ctr_api.cpp clears the counters before every launch and, after it, checks
      //  that every group counted all its tiles — a launch cut short cannot leave a later one a poisoned counter, and a
      //  group that was never delivered is an error, not stale pixels)
      const uint32_t row0 = e_ty * TH, n_rows = AE->rows.n_rows;
      const uint32_t rows_valid = n_rows - row0 < (uint32_t)TH ? n_rows - row0 : (uint32_t)TH;
      const uint32_t x0 = t0 * TW;
      const uint32_t cols = e_w - x0 < nt * TW ? e_w - x0 : nt * TW;  // pixels of the group inside the image
      const size_t stile0 = (size_t)(e_ty * e_tiles_x + t0) * 64u;   // the group's first staging pixel
      float *const h_depth = AE->host_depth, *const h_color = AE->host_color, *const h_normal = AE->host_normal;
      // four rows at a time: 28 loads in flight, then their 28 stores (an atomic load is not moved across a
      // store by the compiler, and a copy that waits for every load in turn would take ~100 us per group)
      constexpr uint32_t RB = TH < 4 ? TH : 4;
      for (uint32_t rb = 0; rb < rows_valid; rb += RB) {
        float vd[RB], vc[RB][3], vn[RB][3];
#pragma unroll
        for (uint32_t q = 0; q < RB; q++) {
          const uint32_t r = rb + q;
          const bool row_ok = r < rows_valid;
          vd[q] = 0.f;
          if (row_ok && lane < cols)
            vd[q] = __hip_atomic_load(depth_out + stile0 + (size_t)(lane / TW) * 64u + r * TW + lane % TW, __ATOMIC_RELAXED,
                                      __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
          for (uint32_t j = 0; j < 3; j++) {
            const uint32_t f = j * 64u + lane, p = f / 3u;  // float f of the row's 3*cols, pixel p of the row
            vc[q][j] = vn[q][j] = 0.f;
            if (row_ok && p < cols) {
              const size_t src = (stile0 + (size_t)(p / TW) * 64u + r * TW + p % TW) * 3u + f % 3u;
              vc[q][j] = __hip_atomic_load(color_out + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              vn[q][j] = __hip_atomic_load(normal_out + src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
          }
        }
#pragma unroll
        for (uint32_t q = 0; q < RB; q++) {
          const uint32_t r = rb + q;
          if (r >= rows_valid) break;
          const size_t dst_px = (size_t)(row0 + r) * e_w + x0;
          // written through to the host (system scope): 2-7 % faster per call than plain stores, which leave it to the
          // L2 when the lines go out (the host buffers may be cached there like any other memory)
          if (lane < cols) __hip_atomic_store(h_depth + dst_px + lane, vd[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
          for (uint32_t j = 0; j < 3; j++) {
            const uint32_t f = j * 64u + lane;
            if (f / 3u < cols) {
              __hip_atomic_store(h_color + dst_px * 3u + f, vc[q][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
              __hip_atomic_store(h_normal + dst_px * 3u + f, vn[q][j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
          }
        }
      }
    }
  }

  CTR_MARK(58);
  if (A.cost && lane == 0) {
    const unsigned long long dt = ((HOSTOUT ? t_wave1 : __builtin_readcyclecounter()) - t_wave0) >> 6;
    const uint32_t c = dt > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)dt;
    A.cost[wave] = c;
  }
#ifdef CTR_WAVELOG
  // diagnostic build only: lane 0 overwrites its pixel's normal with {start, end} of the wave (10 ns ticks)
  if (lane == 0 && in_image) {
    const size_t px_id = px_index();
    const unsigned long long wl_end = __builtin_amdgcn_s_memrealtime();
    normal_out[px_id * 3 + 0] = __uint_as_float((uint32_t)wl_start);
    normal_out[px_id * 3 + 1] = __uint_as_float((uint32_t)wl_end);
    normal_out[px_id * 3 + 2] = __uint_as_float((uint32_t)(st[1] | (st[2] << 12) | (st[3] << 24)));  // nodes, prefilters, exact
  }
#endif
  // ---- per-wave reductions -> 2-3 atomics per wave, spread over CTR_SHARDS cache lines ----
  // (all waves adding into ONE address serialise at the memory side: 0.76 ms per 1080p frame,
  //  measured; one 128-byte shard per wave%CTR_SHARDS costs nothing measurable.  after_render
  //  then adds the shards into the caller's counters and clears them.)
  if (counters) {
    const unsigned long long c = n_casts;
    const uint32_t dbits = wave_dbits;
    unsigned long long t = n_aabb_tris;
    if (COUNT) {
      for (int off = 32; off > 0; off >>= 1) t += __shfl_xor(t, off);
    }
    unsigned long long *sh = counters + (size_t)(wave % CTR_SHARDS) * CTR_SHARD_WORDS;
    if (STATS && lane == 0) {
      for (int q = 0; q < 11; q++) atomicAdd(&sh[4 + q], st[q]);
    }
#ifdef CTR_TIMING
    if (lane == 0) {
      tm[9] = __builtin_readcyclecounter() - tm_start;
      for (int q = 0; q < 10; q++) atomicAdd(&sh[4 + q], tm[q]);
    }
#endif
    if (lane == 0) {
      atomicAdd(&sh[0], c);
      atomicMax(&sh[1], (unsigned long long)dbits);
      if (COUNT) atomicAdd(&sh[2], t);
    }
  }
}
#undef CTR_MARK
// ---- after_render: block 0 folds the counter shards, block 1 builds the next dispatch order ----
// block of CTR_SHARDS threads: thread t owns shard t; wave-level reduction, then one LDS atomic per
// wave and word; adds into out[0..14] (max for word 1) and zeroes the shards for the next launch
__device__ void fold_block(unsigned long long *__restrict__ shards, unsigned long long *__restrict__ out) {
  constexpr int NW = 15;
  __shared__ unsigned long long acc[NW];
  if (threadIdx.x < NW) acc[threadIdx.x] = 0ull;
  __syncthreads();
  unsigned long long *sh = shards + (size_t)threadIdx.x * CTR_SHARD_WORDS;
  unsigned long long v[NW];
#pragma unroll
  for (int q = 0; q < NW; q++) {
    v[q] = sh[q];
    if (v[q]) sh[q] = 0ull;
  }
#pragma unroll
  for (int q = 0; q < NW; q++) {
    unsigned long long x = v[q];
    if (__builtin_amdgcn_ballot_w64(x != 0ull) == 0ull) continue;  // word unused by this build (wave-uniform)
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(x, off);
      x = (q == 1) ? (o > x ? o : x) : x + o;
    }
    if ((threadIdx.x & 63) == 0) {
      if (q == 1) atomicMax(&acc[q], x); else atomicAdd(&acc[q], x);
    }
  }
  __syncthreads();
  if (threadIdx.x < NW) {
    const unsigned long long r = acc[threadIdx.x];
    if (r) {
      if (threadIdx.x == 1) atomicMax(&out[1], r); else atomicAdd(&out[threadIdx.x], r);
    }
  }
}

// The same counting sort (see order_block) over groups of GROUP_TILES horizontally adjacent tiles: a group's key is
// the cost of its most expensive tile, its tiles are emitted consecutively.
__device__ void order_block_groups(const uint32_t *__restrict__ cost, uint32_t *__restrict__ order, uint32_t n, uint32_t tiles_x) {
  __shared__ uint32_t scan[CTR_COST_BINS];
  __shared__ uint32_t wsum[CTR_COST_BINS / 64];
  __shared__ uint32_t smax;
  const uint32_t t = threadIdx.x, ln = t & 63u, wv = t >> 6;
  const uint32_t groups_x = (tiles_x + GROUP_TILES - 1) / GROUP_TILES;
  const uint32_t n_groups = (n / tiles_x) * groups_x;  // (single frame: n = tiles_x * tiles_y)
  auto group = [&](uint32_t g, uint32_t &tile0, uint32_t &nt) -> uint32_t {  // -> the group's key
    const uint32_t gy = g / groups_x, gx = g - gy * groups_x;
    tile0 = gy * tiles_x + gx * GROUP_TILES;
    nt = tiles_x - gx * GROUP_TILES < GROUP_TILES ? tiles_x - gx * GROUP_TILES : GROUP_TILES;
    uint32_t c = 0;
    for (uint32_t k = 0; k < nt; k++) c = cost[tile0 + k] > c ? cost[tile0 + k] : c;
    return c;
  };
  scan[t] = 0u;
  if (t == 0) smax = 1u;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t g = t; g < n_groups; g += CTR_COST_BINS) {
    uint32_t tile0, nt;
    const uint32_t c = group(g, tile0, nt);
    m = c > m ? c : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  if (ln == 0) atomicMax(&smax, m);
  __syncthreads();
  const uint32_t top = __float_as_uint((float)smax);
  auto bin = [&](uint32_t c, uint32_t i) -> uint32_t {
    const uint32_t fb = __float_as_uint((float)c);
    uint32_t cls = fb < top ? (top - fb) >> 20 : 0u;
    cls = cls > 63u ? 63u : cls;
    return cls * 16u + (i & 15u);
  };
  for (uint32_t g = t; g < n_groups; g += CTR_COST_BINS) {
    uint32_t tile0, nt;
    const uint32_t c = group(g, tile0, nt);
    atomicAdd(&scan[bin(c, g)], nt);
  }
  __syncthreads();
  const uint32_t v = scan[t];
  uint32_t x = v;
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)x, off);
    if (ln >= (uint32_t)off) x += o;
  }
  if (ln == 63u) wsum[wv] = x;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t q = 0; q < wv; q++) base += wsum[q];
  scan[t] = base + x - v;
  __syncthreads();
  for (uint32_t g = t; g < n_groups; g += CTR_COST_BINS) {
    uint32_t tile0, nt;
    const uint32_t c = group(g, tile0, nt);
    uint32_t at = atomicAdd(&scan[bin(c, g)], nt);
    if (CTR_CHEAP_FIRST_PCT && tiles_x % GROUP_TILES == 0) {  // (every group is whole)
      // the cheapest groups first, then the rest from the dearest down: the link to the host has something to carry
      // from the start, while the dear tiles — which complete late whatever the order — are under way
      const uint32_t tail = (uint32_t)((uint64_t)n_groups * CTR_CHEAP_FIRST_PCT / 100u) * GROUP_TILES;  // tiles moved to the front
      at = at >= n - tail ? (n - nt - at) : at + tail;   // (the front in ascending cost)
    }
    for (uint32_t k = 0; k < nt; k++) order[at + k] = tile0 + k;
  }
}

// block of CTR_COST_BINS threads: counting sort of the launch's waves by cost class, expensive first:
// order[slot] = wave.  64 classes (8 per octave below the maximum) x 16 sub-bins by wave index —
// the sub-bins only spread the LDS atomics of neighbouring waves, which usually share a class.
// Any permutation is a correct order; cost only shapes the tail of the next launches.
// tiles_x != 0: keep the GROUP_TILES tiles of a host-delivery group together (sorted by the group's cost), so that
// a group completes — and its pixels leave for the host — soon after its first tile starts.
__device__ void order_block(const uint32_t *__restrict__ cost, uint32_t *__restrict__ order, uint32_t n, uint32_t tiles_x) {
  if (tiles_x) {
    order_block_groups(cost, order, n, tiles_x);
    return;
  }
  __shared__ uint32_t scan[CTR_COST_BINS];
  __shared__ uint32_t wsum[CTR_COST_BINS / 64];
  __shared__ uint32_t smax;
  constexpr uint32_t U = 8;
  const uint32_t t = threadIdx.x, ln = t & 63u, wv = t >> 6;
  scan[t] = 0u;
  if (t == 0) smax = 1u;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t b0 = 0; b0 < n; b0 += U * CTR_COST_BINS) {
    uint32_t c[U];
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = b0 + k * CTR_COST_BINS + t;
      c[k] = i < n ? cost[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < U; k++) m = c[k] > m ? c[k] : m;
  }
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)m, off);
    m = o > m ? o : m;
  }
  if (ln == 0) atomicMax(&smax, m);
  __syncthreads();
  const uint32_t top = __float_as_uint((float)smax);
  auto bin = [&](uint32_t c, uint32_t i) -> uint32_t {
    const uint32_t fb = __float_as_uint((float)c);       // exponent | mantissa: log-linear in c
    uint32_t cls = fb < top ? (top - fb) >> 20 : 0u;     // 1/8 octave steps below the maximum
    cls = cls > 63u ? 63u : cls;
    return cls * 16u + (i & 15u);
  };
  for (uint32_t b0 = 0; b0 < n; b0 += U * CTR_COST_BINS) {
    uint32_t c[U];
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = b0 + k * CTR_COST_BINS + t;
      c[k] = i < n ? cost[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = b0 + k * CTR_COST_BINS + t;
      if (i < n) atomicAdd(&scan[bin(c[k], i)], 1u);
    }
  }
  __syncthreads();
  const uint32_t v = scan[t];
  uint32_t x = v;  // inclusive scan inside the wave, then across the 16 waves
  for (int off = 1; off < 64; off <<= 1) {
    const uint32_t o = (uint32_t)__shfl_up((int)x, off);
    if (ln >= (uint32_t)off) x += o;
  }
  if (ln == 63u) wsum[wv] = x;
  __syncthreads();
  uint32_t base = 0;
  for (uint32_t q = 0; q < wv; q++) base += wsum[q];
  scan[t] = base + x - v;  // exclusive offset of bin t
  __syncthreads();
  for (uint32_t b0 = 0; b0 < n; b0 += U * CTR_COST_BINS) {
    uint32_t c[U];
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = b0 + k * CTR_COST_BINS + t;
      c[k] = i < n ? cost[i] : 0u;
    }
#pragma unroll
    for (uint32_t k = 0; k < U; k++) {
      const uint32_t i = b0 + k * CTR_COST_BINS + t;
      if (i < n) order[atomicAdd(&scan[bin(c[k], i)], 1u)] = i;
    }
  }
}

// ---- first_order: the dispatch order of a shape nothing is known about yet ----
// Without measured costs the expensive tiles cannot be started first, but a prior helps: what a frame is about sits
// near its centre, walls and sky at its borders.  Tiles are dispatched in blocks of 16x16 tiles, the blocks by their
// (aspect-normalised) distance from the image centre; list-scheduling the measured costs of the shipped scenes puts
// this 6-15 % below image order (border-in: 5-13 % above; DESIGN.md "First launch").  One thread per tile.
__global__ __launch_bounds__(256) void first_order(uint32_t *__restrict__ order, uint32_t tiles_x, uint32_t tiles_y, uint32_t n_frames) {
  const uint32_t tiles_frame = tiles_x * tiles_y;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tiles_frame) return;
  constexpr uint32_t B = 16;
  const uint32_t nbx = (tiles_x + B - 1) / B, nby = (tiles_y + B - 1) / B;
  const uint32_t tx = t % tiles_x, ty = t / tiles_x, bx = tx / B, by = ty / B, me = by * nbx + bx;
  const float aspect = (float)nbx / (float)nby;
  auto key = [&](uint32_t x, uint32_t y) -> float {
    const float dx = ((float)x + 0.5f) - 0.5f * (float)nbx, dy = (((float)y + 0.5f) - 0.5f * (float)nby) * aspect;
    return dx * dx + dy * dy;
  };
  const float km = key(bx, by);
  uint32_t before = 0;  // tiles of the blocks that come first
  for (uint32_t y = 0; y < nby; y++)
    for (uint32_t x = 0; x < nbx; x++) {
      const float k = key(x, y);
      if (k < km || (k == km && y * nbx + x < me)) {
        const uint32_t cw = tiles_x - x * B < B ? tiles_x - x * B : B, ch = tiles_y - y * B < B ? tiles_y - y * B : B;
        before += cw * ch;
      }
    }
  const uint32_t bw = tiles_x - bx * B < B ? tiles_x - bx * B : B;
  const uint32_t slot = before + (ty - by * B) * bw + (tx - bx * B);
  for (uint32_t f = 0; f < n_frames; f++) order[f * tiles_frame + slot] = f * tiles_frame + t;
}

static_assert(CTR_SHARDS == CTR_COST_BINS, "after_render uses one block size for both jobs");
__global__ __launch_bounds__(CTR_SHARDS) void after_render(unsigned long long *__restrict__ shards,
                                                           unsigned long long *__restrict__ counters,
                                                           const uint32_t *__restrict__ cost,
                                                           uint32_t *__restrict__ order, uint32_t n, uint32_t group_tiles_x) {
  if (blockIdx.x == 0) {
    if (shards) fold_block(shards, counters);
  } else {
    if (cost) order_block(cost, order, n, group_tiles_x);
  }
}

uint64_t launch_waves(const RenderLaunch &L) {
  const uint32_t tiles_x = (L.w + TW - 1) / TW, tiles_y = (L.rows.n_rows + TH - 1) / TH;
  return (uint64_t)tiles_x * tiles_y * L.n_frames;
}

template <uint32_t KV>
int launch(const RenderLaunch &L, hipStream_t stream) {
  KArgs A;
  A.objs = (const CADDR DObj *)L.objs;
  A.oloop = (const CADDR DObj *)L.oloop;
  A.meshes = (const CADDR DObj *)L.meshes;
  A.n_mesh = L.n_mesh;
  A.tlas_root = L.tlas_root;
  A.tlas_root2 = L.tlas_root_regular;
  A.tlas_begin = L.tlas_begin;
  for (int q = 0; q < 3; q++) { A.tl_mn[q] = L.tl_mn[q]; A.tl_mx[q] = L.tl_mx[q]; }
  A.planes = (const CADDR DPlanePair *)L.planes;
  A.n_oloop = L.n_oloop;
  A.n_plane_recs = L.n_plane_recs;
  A.n_axis_recs = L.n_axis_recs;
  A.tris = (const CADDR DTri *)L.tris;
  A.nodes = (const CADDR DNode *)L.nodes;
  A.nodes4 = (const CADDR DNode4 *)L.nodes4;
  A.gnorm = (const CADDR float *)L.gnorm;
  A.lights = (const CADDR DLight *)L.lights;
  A.mats = (const CADDR DMat *)L.mats;
  A.n_obj = L.n_obj;
  A.n_light = L.n_light;
  A.cams = (const CADDR DCam *)L.cams;
  A.w = L.w;
  A.h = L.h;
  A.first_frame = L.first_frame;
  A.n_frames = L.n_frames;
  A.frame_stride_px = L.frame_stride_px;
  A.rows = L.rows;
  A.fudge = L.fudge;
  A.bounces = L.bounces;
  A.has_mesh = L.has_mesh;
  A.nf = L.need_cold_frames ? 10u : 4u;
  A.frames = (uint32_t)((L.bounces > 0 && L.any_bounce) ? L.bounces : 1);
  A.order = (const CADDR uint32_t *)L.order;
  A.cost = L.cost;
  const bool host_delivery = (KV & KV_HOSTOUT) != 0;
  if (host_delivery != (L.group_done != nullptr)) return (int)hipErrorInvalidValue;
  if (host_delivery && (L.n_frames != 1 || !L.host_depth || !L.host_color || !L.host_normal)) return (int)hipErrorInvalidValue;
  A.host_depth = L.host_depth;
  A.host_color = L.host_color;
  A.host_normal = L.host_normal;
  A.group_done = L.group_done;
  A.uv_out = L.uv;
  if (((KV & KV_UV) != 0) != (L.uv != nullptr)) return (int)hipErrorInvalidValue;
  size_t lds_bytes = (size_t)WAVES_PER_WG * A.frames * A.nf * 64 * sizeof(float);
  if (KV & KV_OCC6) lds_bytes += (size_t)WAVES_PER_WG * 5 * 64 * sizeof(float);  // PARK
  // diagnostic only: extra dynamic LDS per workgroup caps the waves resident per CU (occupancy sweeps)
  if (const char *pad = getenv("CUTRACE_LDS_PAD")) lds_bytes += (size_t)atol(pad);
  const uint64_t waves = launch_waves(L);
  if (waves > 0x7FFFFFFFull) return (int)hipErrorInvalidValue;
  if (waves == 0) return 0;
  const uint32_t grid = (uint32_t)((waves + WAVES_PER_WG - 1) / WAVES_PER_WG);
  // counters go through the scene's shard buffer and are folded into the caller's words afterwards
  unsigned long long *shards = L.counters ? L.shards : nullptr;
  if (L.order_init && L.order) {
    const uint32_t tiles_x = (L.w + TW - 1) / TW, tiles_y = (L.rows.n_rows + TH - 1) / TH;
    hipLaunchKernelGGL(first_order, dim3((tiles_x * tiles_y + 255) / 256), dim3(256), 0, stream, const_cast<uint32_t *>(L.order),
                       tiles_x, tiles_y, L.n_frames);
  }
  hipLaunchKernelGGL(render_kernel<KV>, dim3(grid), dim3(WG_THREADS), lds_bytes, stream, A, L.depth, L.color, L.normal,
                     shards);
  const bool reorder = L.cost && L.order_next;
  if (shards || reorder)
    hipLaunchKernelGGL(after_render, dim3(reorder ? 2 : 1), dim3(CTR_SHARDS), 0, stream, shards, L.counters,
                       reorder ? L.cost : nullptr, L.order_next, (uint32_t)waves,
                       host_delivery ? (L.w + TW - 1) / TW : 0u);
  return (int)hipGetLastError();
}

template <uint32_t BASE>
int launch_main(const RenderLaunch &L, hipStream_t s) {
  switch (L.variant & (KV_PREFILTER | KV_ANYHIT | KV_BVH)) {
    case 0: return launch<BASE>(L, s);
    case KV_PREFILTER: return launch<BASE | KV_PREFILTER>(L, s);
    case KV_ANYHIT: return launch<BASE | KV_ANYHIT>(L, s);
    case KV_PREFILTER | KV_ANYHIT: return launch<BASE | KV_PREFILTER | KV_ANYHIT>(L, s);
    case KV_BVH: return launch<BASE | KV_BVH>(L, s);
    case KV_BVH | KV_PREFILTER: return launch<BASE | KV_BVH | KV_PREFILTER>(L, s);
    case KV_BVH | KV_ANYHIT: return launch<BASE | KV_BVH | KV_ANYHIT>(L, s);
    default: return launch<BASE | KV_BVH | KV_PREFILTER | KV_ANYHIT>(L, s);
  }
}

}  // namespace

// ---- self-test of norm_and_inverse against the library's correctly rounded sqrtf and division ----
namespace {
__global__ __launch_bounds__(256) void selftest_exact_math(unsigned long long *bad) {
  // every mantissa (2^23) x both exponent parities, at exponents -79/-78, -1/0 and 78/79: thread t takes 6 values
  const uint32_t m = blockIdx.x * blockDim.x + threadIdx.x;  // 0 .. 2^23-1
  unsigned long long wrong = 0;
  const int exps[6] = {127 - 79, 127 - 78, 126, 127, 127 + 78, 127 + 79};
  for (int k = 0; k < 6; k++) {
    float x = __uint_as_float(((uint32_t)exps[k] << 23) | m);
    asm volatile("" : "+v"(x));
    float n, inv;
    norm_and_inverse(x, n, inv);
    const float n_ref = sqrtf(x);
    const float inv_ref = 1.0f / n_ref;
    if (__float_as_uint(n) != __float_as_uint(n_ref) || __float_as_uint(inv) != __float_as_uint(inv_ref)) wrong++;
  }
  for (int off = 32; off > 0; off >>= 1) wrong += __shfl_xor(wrong, off);
  if ((threadIdx.x & 63) == 0 && wrong) atomicAdd(bad, wrong);
}
}  // namespace

#if defined(CTR_PROFILE)
// profile builds only (scripts/dynamic_mix.py): the segment counters, optionally cleared
extern "C" int ctr_debug_profile_read(uint64_t *out128, int reset) {
  unsigned int h[128];
  if (hipDeviceSynchronize() != hipSuccess) return CTR_E_INVALID;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof(h)) != hipSuccess) return CTR_E_INVALID;
  for (int q = 0; q < 128; q++) out128[q] = h[q];
  if (reset) {
    for (int q = 0; q < 128; q++) h[q] = 0u;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof(h)) != hipSuccess) return CTR_E_INVALID;
  }
  return CTR_OK;
}
#endif

// live-lane statistics of the CTR_VAR_STATS launches since the last reset (g_lane_stats above): 96 words
extern "C" int ctr_debug_lane_stats(uint64_t *out80, int reset) {
  if (!out80) return CTR_E_INVALID;
  unsigned long long h[96];
  if (hipDeviceSynchronize() != hipSuccess) return CTR_E_INVALID;
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_lane_stats), sizeof(h)) != hipSuccess) return CTR_E_INVALID;
  for (int q = 0; q < 96; q++) out80[q] = h[q];
  if (reset) {
    for (int q = 0; q < 96; q++) h[q] = 0ull;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_lane_stats), h, sizeof(h)) != hipSuccess) return CTR_E_INVALID;
  }
  return CTR_OK;
}

extern "C" int ctr_selftest_exact_math(uint64_t *n_mismatch) {
  if (!n_mismatch) return CTR_E_INVALID;
  unsigned long long *d = nullptr;
  hipError_t e = hipMalloc((void **)&d, sizeof(unsigned long long));
  if (e != hipSuccess) return CTR_E_HIP_BASE + (int)e;
  (void)hipMemset(d, 0, sizeof(unsigned long long));
  hipLaunchKernelGGL(selftest_exact_math, dim3((1u << 23) / 256), dim3(256), 0, nullptr, d);
  unsigned long long h = ~0ull;
  e = hipMemcpy(&h, d, sizeof(h), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (e != hipSuccess) return CTR_E_HIP_BASE + (int)e;
  *n_mismatch = h;
  return CTR_OK;
}

uint64_t ctr_launch_waves(const RenderLaunch &L) { return launch_waves(L); }
uint64_t ctr_staging_pixels(const RenderLaunch &L) { return launch_waves(L) * 64u; }
uint64_t ctr_staging_groups(const RenderLaunch &L) {
  const uint64_t tiles_x = (L.w + TW - 1) / TW, tiles_y = (L.rows.n_rows + TH - 1) / TH;
  return ((tiles_x + GROUP_TILES - 1) / GROUP_TILES) * tiles_y;
}

uint32_t ctr_group_tile_count(const RenderLaunch &L, uint64_t group) {
  const uint32_t tiles_x = (L.w + TW - 1) / TW, groups_x = (tiles_x + GROUP_TILES - 1) / GROUP_TILES;
  const uint32_t gx = (uint32_t)(group % groups_x);
  return tiles_x - gx * GROUP_TILES < GROUP_TILES ? tiles_x - gx * GROUP_TILES : GROUP_TILES;
}
uint64_t ctr_staging_index(const RenderLaunch &L, uint32_t x, uint32_t k_row) {
  const uint64_t tiles_x = (L.w + TW - 1) / TW;
  return ((uint64_t)(k_row / TH) * tiles_x + x / TW) * 64u + (k_row % TH) * TW + x % TW;
}

// The 6-waves-per-SIMD build parks 1280 bytes of shading state per wave in LDS (render_kernel PARK); LDS is handed out
// in granules of 1280 bytes (160 KB / 128), and 24 waves per CU need at most 5 granules each
static bool occ6_fits(size_t stack_bytes) { return (stack_bytes + 1280u + 1279u) / 1280u <= 5u; }

// Host delivery exists for the variants ctr_api.cpp picks by itself (not for the ablation / diagnostic builds)
bool ctr_host_delivery_available(uint32_t variant) {
  constexpr uint32_t DEF = KV_PREFILTER | KV_BVH | KV_FASTPOW;
  return (variant & ~(KV_ANYHIT | KV_OCC6)) == DEF;
}

int ctr_launch_render(const RenderLaunch &L, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  if (L.group_done) {
    if (!ctr_host_delivery_available(L.variant)) return (int)hipErrorInvalidValue;
    constexpr uint32_t DEF = KV_PREFILTER | KV_BVH | KV_FASTPOW | KV_HOSTOUT;
    const size_t sb = (size_t)((L.bounces > 0 && L.any_bounce) ? L.bounces : 1) * (L.need_cold_frames ? 10u : 4u) * 64 * sizeof(float);
    if (!(L.variant & KV_ANYHIT)) return launch<DEF>(L, s);
    if ((L.variant & KV_OCC6) && occ6_fits(sb)) return launch<DEF | KV_ANYHIT | KV_OCC6>(L, s);
    return launch<DEF | KV_ANYHIT>(L, s);
  }
  if ((L.variant & (KV_UV | KV_IGNTR)) == (KV_UV | KV_IGNTR)) {  // ... with the kernel.hpp:52 cast ignoring transparent objects
    constexpr uint32_t U = KV_PREFILTER | KV_BVH | KV_UV | KV_IGNTR;
    if (L.variant & KV_FASTPOW) return (L.variant & KV_ANYHIT) ? launch<U | KV_FASTPOW | KV_ANYHIT>(L, s) : launch<U | KV_FASTPOW>(L, s);
    return (L.variant & KV_ANYHIT) ? launch<U | KV_ANYHIT>(L, s) : launch<U>(L, s);
  }
  if (L.variant & KV_UV) {  // the fourth output: the shipped walk only (BVH + prefilter), any-hit and pow as the scene / caller say
    constexpr uint32_t U = KV_PREFILTER | KV_BVH | KV_UV;
    if (L.variant & KV_FASTPOW) return (L.variant & KV_ANYHIT) ? launch<U | KV_FASTPOW | KV_ANYHIT>(L, s) : launch<U | KV_FASTPOW>(L, s);
    return (L.variant & KV_ANYHIT) ? launch<U | KV_ANYHIT>(L, s) : launch<U>(L, s);
  }
  if (L.variant & KV_COUNT) return launch<KV_PREFILTER | KV_COUNT>(L, s);
  if (L.variant & KV_MERGE) {
    // the merged walk (CTR_VAR_MERGE, scenes with several meshes): the shipped walk's variants only
    constexpr uint32_t M = KV_PREFILTER | KV_BVH | KV_MERGE;
    const bool any = (L.variant & KV_ANYHIT) != 0;
    if (L.variant & KV_STATS) return any ? launch<M | KV_ANYHIT | KV_FASTPOW | KV_STATS>(L, s) : launch<M | KV_FASTPOW | KV_STATS>(L, s);
    if (!(L.variant & KV_FASTPOW)) return any ? launch<M | KV_ANYHIT>(L, s) : launch<M>(L, s);
    const size_t sb = (size_t)((L.bounces > 0 && L.any_bounce) ? L.bounces : 1) * (L.need_cold_frames ? 10u : 4u) * 64 * sizeof(float);
    if (any && (L.variant & KV_OCC6) && occ6_fits(sb)) return launch<M | KV_FASTPOW | KV_ANYHIT | KV_OCC6>(L, s);
    return any ? launch<M | KV_FASTPOW | KV_ANYHIT>(L, s) : launch<M | KV_FASTPOW>(L, s);
  }
  if (L.variant & KV_STATS)
    return (L.variant & KV_ANYHIT) ? launch<KV_BVH | KV_PREFILTER | KV_ANYHIT | KV_FASTPOW | KV_STATS>(L, s)
                                   : launch<KV_BVH | KV_PREFILTER | KV_FASTPOW | KV_STATS>(L, s);
  // KV_OCC6: the same kernel compiled for 6 waves per SIMD (80 VGPRs; five cold dwords parked in LDS, see PARK) instead
  // of 5 (85): worth it where the mesh data exceed the scalar cache many times over and a wave mostly waits for L2
  // (ctr_api.cpp picks it by triangle count); only the shipped default variant has it (and only when 24 waves' recursion
  // stacks plus the parking granule fit the CU's 160 KB of LDS: bounces <= 5 without cold frames, occ6_fits)
  const size_t stack_bytes = (size_t)((L.bounces > 0 && L.any_bounce) ? L.bounces : 1) * (L.need_cold_frames ? 10u : 4u) * 64 * sizeof(float);
  if ((L.variant & KV_OCC6) && occ6_fits(stack_bytes) &&
      (L.variant & (KV_PREFILTER | KV_ANYHIT | KV_BVH | KV_FASTPOW)) == (KV_PREFILTER | KV_ANYHIT | KV_BVH | KV_FASTPOW))
    return launch<KV_OCC6 | KV_PREFILTER | KV_ANYHIT | KV_BVH | KV_FASTPOW>(L, s);
  return (L.variant & KV_FASTPOW) ? launch_main<KV_FASTPOW>(L, s) : launch_main<0>(L, s);
}
