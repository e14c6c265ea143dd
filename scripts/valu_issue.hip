// valu_issue.hip — gfx950 microbenchmark: what does one wave64 VALU instruction cost to issue?
//
// MI355X_MICROARCH.md says a wave64 v_fma_f32 occupies the SIMD-32 for 2 cycles once
// more than one wave shares the SIMD ("one wave alone: 4"); DESIGN r01 charged 4.  This measures it on
// the box for the instruction KINDS the render kernel is made of: VALU with VGPR operands only, VALU
// reading an SGPR operand (how the kernel consumes its wave-uniform scene records), compares that
// write an SGPR lane mask, packed f32, transcendental, lane <-> SGPR moves, SALU, and the latency of a
// dependent chain of scalar loads (cache-resident and not).  1..8 waves resident per SIMD, timed
// inside the kernel with s_memtime (shader clock) and s_memrealtime (100 MHz).  Every wave records
// where it ran (HW_ID, XCC_ID), so the host checks how many waves really shared each SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o scripts/bin/valu_issue scripts/valu_issue.hip
//   scripts/bin/valu_issue [mode-substring] [k,k,...]      (results: profiles/r02/valu_issue.txt)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#define HIP_OK(x)                                                                      \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                          \
      exit(1);                                                                         \
    }                                                                                  \
  } while (0)

struct Rec {
  uint64_t cycles;    // s_memtime delta
  uint64_t realtime;  // s_memrealtime delta (100 MHz ticks)
  uint32_t hw_id, xcc_id;
};

typedef float f2 __attribute__((ext_vector_type(2)));

// one body = 8 instructions; operands (the same list for every body):
//   %0..%7  a0..a7   VGPR accumulators        %8..%11 p0..p3  VGPR-pair accumulators
//   %12     sacc     SGPR accumulator         %13 m, %14 c    VGPR inputs
//   %15     sm       SGPR input               %16 sp          SGPR-pair input      %17 pm  VGPR-pair input
#define BODY(str)                                                                                          \
  asm volatile(str                                                                                         \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(p0),  \
                 "+v"(p1), "+v"(p2), "+v"(p3), "+s"(sacc)                                                  \
               : "v"(m), "v"(c), "s"(sm), "s"(sp), "v"(pm)                                                 \
               : "scc", "vcc", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", \
                 "s51", "s52", "s53", "s54", "s55")

#define X8(pre, post) pre "%0" post pre "%1" post pre "%2" post pre "%3" post pre "%4" post pre "%5" post pre "%6" post pre "%7" post

struct ModeInfo {
  const char *name;
  int valu, insts;  // VALU instructions / all instructions per body of 8
};
enum {
  M_FMA_V = 0, M_FMA_S, M_FMAC_V, M_FMAC_S, M_MUL_S, M_ADD_V, M_MAX3_V, M_FMA_INL, M_PK_V, M_PK_S, M_PKMUL_S,
  M_CMP_VCC, M_CMP_S64, M_CMP_SOP, M_MOV_S, M_READLANE, M_CNDMASK_S, M_RCP, M_DEP, M_SALU, M_SALU64, M_FMA_SALU,
  M_MIX, M_KMIX, M_KMIX_NOSALU, M_COUNT
};
static const ModeInfo modes[M_COUNT] = {
    {"fma_v       v_fma_f32 v,v,v,v            (VGPR operands only)", 8, 8},
    {"fma_s       v_fma_f32 v,s,v,v            (one SGPR operand)", 8, 8},
    {"fmac_v      v_fmac_f32_e32 v,v,v         (VOP2, VGPR only)", 8, 8},
    {"fmac_s      v_fmac_f32_e32 v,s,v         (VOP2, SGPR src0)", 8, 8},
    {"mul_s       v_mul_f32_e32 v,s,v          (VOP2, SGPR src0)", 8, 8},
    {"add_v       v_add_f32_e32 v,v,v", 8, 8},
    {"max3_v      v_max3_f32 v,v,v,v", 8, 8},
    {"fma_inl     v_fma_f32 v,v,1.0,v          (inline constant)", 8, 8},
    {"pk_v        v_pk_fma_f32, VGPR pairs only", 8, 8},
    {"pk_s        v_pk_fma_f32 with an SGPR-pair operand", 8, 8},
    {"pkmul_s     v_pk_mul_f32 with an SGPR-pair operand", 8, 8},
    {"cmp_vcc     v_cmp_lt_f32_e32 vcc,v,v", 8, 8},
    {"cmp_s64     v_cmp_lt_f32_e64 s[..],v,v   (lane mask into an SGPR pair)", 8, 8},
    {"cmp_sop     v_cmp_lt_f32_e64 s[..],s,v   (SGPR operand and SGPR-pair result)", 8, 8},
    {"mov_s       v_mov_b32 v,s", 8, 8},
    {"readlane    v_readlane_b32 s,v,imm", 8, 8},
    {"cndmask_s   v_cndmask_b32_e64 v,v,v,s[..]", 8, 8},
    {"rcp         v_rcp_f32 v,v", 8, 8},
    {"dep         v_fma_f32 dependent chain (one accumulator)", 8, 8},
    {"salu        s_add_u32 (independent)", 0, 8},
    {"salu64      s_and_b64 / s_or_b64 (independent)", 0, 8},
    {"fma_salu    v_fma_f32 (VGPR only) alternating with s_add_u32", 4, 8},
    {"mix         6 fma_s + 2 pk_s + 2 cmp_s64 + 3 SALU (render-like, 13 instructions)", 10, 13},
    // round 3: a stream with the render kernel's EXECUTED mix (profiles/r03/valu_mix_dynamic_bunny.json: F 0.525, H 0.460,
    // Q 0.015 of the VALU instructions; 0.64 SALU per VALU): 40 VALU = 20 plain + 19 half-rate (9 with an SGPR operand, 4
    // packed with an SGPR pair, 4 compares into SGPR pairs, 2 max3) + 1 transcendental, and 26 SALU between them.  The
    // additive model prices it (20 x 2.2 + 19 x 4.1 + 8.1) / 40 = 3.25 cycles per VALU; what does the SIMD take?
    {"kmix        the render kernel's executed mix: 20 F + 19 H + 1 Q VALU + 26 SALU", 40, 66},
    {"kmix_nosalu the same 40 VALU without the scalar instructions", 40, 40},
};

template <int MODE>
__global__ void bench(Rec *out, int iters, float seed) {
  float a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  f2 p0 = {seed, seed}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f;
  const float m = 0.999f, c = 0.001f;
  const f2 pm = {0.999f, 0.998f};
  float sm = __builtin_amdgcn_readfirstlane(m);
  f2 sp;
  sp.x = __builtin_amdgcn_readfirstlane(pm.x);
  sp.y = __builtin_amdgcn_readfirstlane(pm.y);
  uint32_t sacc = 0;
  asm volatile("s_mov_b64 s[40:41], -1\n s_mov_b64 s[42:43], 0\n s_mov_b64 s[44:45], -1\n s_mov_b64 s[46:47], 0\n"
               "s_mov_b64 s[48:49], -1\n s_mov_b64 s[50:51], 0\n s_mov_b64 s[52:53], -1\n s_mov_b64 s[54:55], 0" ::
                   : "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55");
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#define R8(X) X; X; X; X; X; X; X; X
    if (MODE == M_FMA_V) { R8(BODY(X8("v_fma_f32 ", ", %13, %14, %14\n"))); }
    if (MODE == M_FMA_S) {
      R8(BODY("v_fma_f32 %0, %15, %0, %14\n v_fma_f32 %1, %15, %1, %14\n v_fma_f32 %2, %15, %2, %14\n v_fma_f32 %3, %15, %3, %14\n"
              "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %15, %5, %14\n v_fma_f32 %6, %15, %6, %14\n v_fma_f32 %7, %15, %7, %14\n"));
    }
    if (MODE == M_FMAC_V) { R8(BODY(X8("v_fmac_f32_e32 ", ", %13, %14\n"))); }
    if (MODE == M_FMAC_S) { R8(BODY(X8("v_fmac_f32_e32 ", ", %15, %14\n"))); }
    if (MODE == M_MUL_S) { R8(BODY(X8("v_mul_f32_e32 ", ", %15, %14\n"))); }
    if (MODE == M_ADD_V) { R8(BODY(X8("v_add_f32_e32 ", ", %13, %14\n"))); }
    if (MODE == M_MAX3_V) { R8(BODY(X8("v_max3_f32 ", ", %13, %14, %14\n"))); }
    if (MODE == M_FMA_INL) { R8(BODY(X8("v_fma_f32 ", ", %13, 1.0, %14\n"))); }
    if (MODE == M_PK_V) {
      R8(BODY("v_pk_fma_f32 %8, %8, %17, %17\n v_pk_fma_f32 %9, %9, %17, %17\n v_pk_fma_f32 %10, %10, %17, %17\n v_pk_fma_f32 %11, %11, %17, %17\n"
              "v_pk_fma_f32 %8, %8, %17, %17\n v_pk_fma_f32 %9, %9, %17, %17\n v_pk_fma_f32 %10, %10, %17, %17\n v_pk_fma_f32 %11, %11, %17, %17\n"));
    }
    if (MODE == M_PK_S) {
      R8(BODY("v_pk_fma_f32 %8, %16, %8, %17\n v_pk_fma_f32 %9, %16, %9, %17\n v_pk_fma_f32 %10, %16, %10, %17\n v_pk_fma_f32 %11, %16, %11, %17\n"
              "v_pk_fma_f32 %8, %16, %8, %17\n v_pk_fma_f32 %9, %16, %9, %17\n v_pk_fma_f32 %10, %16, %10, %17\n v_pk_fma_f32 %11, %16, %11, %17\n"));
    }
    if (MODE == M_PKMUL_S) {
      R8(BODY("v_pk_mul_f32 %8, %16, %17\n v_pk_mul_f32 %9, %16, %17\n v_pk_mul_f32 %10, %16, %17\n v_pk_mul_f32 %11, %16, %17\n"
              "v_pk_mul_f32 %8, %16, %17\n v_pk_mul_f32 %9, %16, %17\n v_pk_mul_f32 %10, %16, %17\n v_pk_mul_f32 %11, %16, %17\n"));
    }
    if (MODE == M_CMP_VCC) { R8(BODY(X8("v_cmp_lt_f32_e32 vcc, %13, ", "\n"))); }
    if (MODE == M_CMP_S64) {
      R8(BODY("v_cmp_lt_f32_e64 s[40:41], %0, %1\n v_cmp_lt_f32_e64 s[42:43], %1, %2\n v_cmp_lt_f32_e64 s[44:45], %2, %3\n v_cmp_lt_f32_e64 s[46:47], %3, %0\n"
              "v_cmp_lt_f32_e64 s[48:49], %0, %2\n v_cmp_lt_f32_e64 s[50:51], %1, %3\n v_cmp_lt_f32_e64 s[52:53], %2, %0\n v_cmp_lt_f32_e64 s[54:55], %3, %1\n"));
    }
    if (MODE == M_CMP_SOP) {
      R8(BODY("v_cmp_lt_f32_e64 s[40:41], %15, %1\n v_cmp_lt_f32_e64 s[42:43], %15, %2\n v_cmp_lt_f32_e64 s[44:45], %15, %3\n v_cmp_lt_f32_e64 s[46:47], %15, %0\n"
              "v_cmp_lt_f32_e64 s[48:49], %15, %2\n v_cmp_lt_f32_e64 s[50:51], %15, %3\n v_cmp_lt_f32_e64 s[52:53], %15, %0\n v_cmp_lt_f32_e64 s[54:55], %15, %1\n"));
    }
    if (MODE == M_MOV_S) { R8(BODY(X8("v_mov_b32_e32 ", ", %15\n"))); }
    if (MODE == M_READLANE) {
      R8(BODY("v_readlane_b32 s40, %0, 1\n v_readlane_b32 s41, %1, 2\n v_readlane_b32 s42, %2, 3\n v_readlane_b32 s43, %3, 4\n"
              "v_readlane_b32 s44, %4, 5\n v_readlane_b32 s45, %5, 6\n v_readlane_b32 s46, %6, 7\n v_readlane_b32 s47, %7, 8\n"));
    }
    if (MODE == M_CNDMASK_S) { R8(BODY(X8("v_cndmask_b32_e64 ", ", %13, %14, s[40:41]\n"))); }
    if (MODE == M_RCP) {
      R8(BODY("v_rcp_f32_e32 %0, %0\n v_rcp_f32_e32 %1, %1\n v_rcp_f32_e32 %2, %2\n v_rcp_f32_e32 %3, %3\n"
              "v_rcp_f32_e32 %4, %4\n v_rcp_f32_e32 %5, %5\n v_rcp_f32_e32 %6, %6\n v_rcp_f32_e32 %7, %7\n"));
    }
    if (MODE == M_DEP) {
      R8(BODY("v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n"
              "v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %0, %0, %13, %14\n"));
    }
    if (MODE == M_SALU) {
      R8(BODY("s_add_u32 s40, s41, 1\n s_add_u32 s42, s43, 1\n s_add_u32 s44, s45, 1\n s_add_u32 s46, s47, 1\n"
              "s_add_u32 s48, s49, 1\n s_add_u32 s50, s51, 1\n s_add_u32 s52, s53, 1\n s_add_u32 s54, s55, 1\n"));
    }
    if (MODE == M_SALU64) {
      R8(BODY("s_and_b64 s[40:41], s[42:43], s[44:45]\n s_or_b64 s[46:47], s[48:49], s[50:51]\n s_and_b64 s[52:53], s[42:43], s[44:45]\n s_or_b64 s[54:55], s[48:49], s[50:51]\n"
              "s_and_b64 s[40:41], s[42:43], s[44:45]\n s_or_b64 s[46:47], s[48:49], s[50:51]\n s_and_b64 s[52:53], s[42:43], s[44:45]\n s_or_b64 s[54:55], s[48:49], s[50:51]\n"));
    }
    if (MODE == M_FMA_SALU) {
      R8(BODY("v_fma_f32 %0, %0, %13, %14\n s_add_u32 %12, %12, 1\n v_fma_f32 %1, %1, %13, %14\n s_add_u32 s40, s41, 3\n"
              "v_fma_f32 %2, %2, %13, %14\n s_add_u32 s42, s43, 5\n v_fma_f32 %3, %3, %13, %14\n s_add_u32 s44, s45, 7\n"));
    }
    if (MODE == M_MIX) {
      R8(BODY("v_fma_f32 %0, %15, %0, %14\n v_fma_f32 %1, %15, %1, %14\n v_pk_fma_f32 %8, %16, %8, %17\n s_add_u32 %12, %12, 1\n"
              "v_fma_f32 %2, %15, %2, %14\n v_fma_f32 %3, %15, %3, %14\n v_pk_fma_f32 %9, %16, %9, %17\n s_and_b32 s40, %12, 7\n"
              "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %15, %5, %14\n v_cmp_lt_f32_e64 s[42:43], %0, %1\n"
              "v_cmp_lt_f32_e64 s[44:45], %2, %3\n s_or_b64 s[46:47], s[42:43], s[44:45]\n"));
    }
#define KM_V1 "v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %1, %15, %1, %14\n v_fma_f32 %2, %2, %13, %14\n v_pk_fma_f32 %8, %16, %8, %17\n v_fma_f32 %3, %3, %13, %14\n"
#define KM_V2 "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %5, %13, %14\n v_cmp_lt_f32_e64 s[42:43], %0, %1\n v_fma_f32 %6, %6, %13, %14\n v_fma_f32 %7, %15, %7, %14\n"
#define KM_V3 "v_fma_f32 %0, %0, %13, %14\n v_max3_f32 %1, %1, %13, %14\n v_fma_f32 %2, %2, %13, %14\n v_pk_fma_f32 %9, %16, %9, %17\n v_fma_f32 %3, %3, %13, %14\n"
#define KM_V4 "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %5, %13, %14\n v_cmp_lt_f32_e64 s[44:45], %2, %3\n v_fma_f32 %6, %6, %13, %14\n v_fma_f32 %7, %15, %7, %14\n"
#define KM_V5 "v_fma_f32 %0, %0, %13, %14\n v_fma_f32 %1, %15, %1, %14\n v_fma_f32 %2, %2, %13, %14\n v_pk_fma_f32 %10, %16, %10, %17\n v_fma_f32 %3, %3, %13, %14\n"
#define KM_V6 "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %5, %13, %14\n v_cmp_lt_f32_e64 s[46:47], %4, %5\n v_fma_f32 %6, %6, %13, %14\n v_rcp_f32_e32 %7, %7\n"
#define KM_V7 "v_fma_f32 %0, %0, %13, %14\n v_max3_f32 %1, %1, %13, %14\n v_fma_f32 %2, %2, %13, %14\n v_pk_fma_f32 %11, %16, %11, %17\n v_fma_f32 %3, %3, %13, %14\n"
#define KM_V8 "v_fma_f32 %4, %15, %4, %14\n v_fma_f32 %5, %5, %13, %14\n v_cmp_lt_f32_e64 s[48:49], %6, %7\n v_fma_f32 %6, %6, %13, %14\n v_fma_f32 %7, %15, %7, %14\n"
#define KM_S3a "s_add_u32 %12, %12, 1\n s_and_b64 s[50:51], s[42:43], s[44:45]\n s_and_b32 s40, %12, 7\n"
#define KM_S3b "s_or_b64 s[52:53], s[46:47], s[48:49]\n s_add_u32 s41, s40, 3\n s_andn2_b64 s[54:55], s[50:51], s[52:53]\n"
#define KM_S4 "s_lshl_b32 s40, s41, 2\n s_cmp_lg_u64 s[54:55], 0\n s_cselect_b32 s41, s40, 5\n s_add_u32 %12, %12, s41\n"
    if (MODE == M_KMIX) {  // 8 groups of 5 VALU, 26 SALU spread between them
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
      BODY(KM_V1 KM_S3a KM_V2 KM_S3b KM_V3 KM_S4 KM_V4 KM_S3a KM_V5 KM_S3b KM_V6 KM_S4 KM_V7 KM_S3a KM_V8 KM_S3b);
    }
    if (MODE == M_KMIX_NOSALU) { R8(BODY(KM_V1 KM_V2 KM_V3 KM_V4 KM_V5 KM_V6 KM_V7 KM_V8)); }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  float sink = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + (float)sacc;
  if (sink == 123.456f) out[0].cycles = 1;  // keep the accumulators alive
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.cycles = t1 - t0;
    r.realtime = r1 - r0;
    r.hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    r.xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = r;
  }
}

// ---- dependent chain of scalar loads: every wave chases its own cycle through a table of 64-byte
//      records (record i holds the index of the next one), like the BVH walk of the render kernel ----
__global__ void chase(const uint32_t *__restrict__ table, Rec *out, int steps, uint32_t n_rec) {
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t cur = __builtin_amdgcn_readfirstlane((wave * 977u) % n_rec);
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  const uint64_t r0 = __builtin_amdgcn_s_memrealtime();
  const __attribute__((address_space(4))) uint32_t *tab = (const __attribute__((address_space(4))) uint32_t *)table;
  for (int s = 0; s < steps; ++s) {
    cur = tab[(size_t)cur * 16];  // s_load_dword, address depends on the previous load
    asm volatile("" : "+s"(cur));
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  const uint64_t r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) {
    Rec r;
    r.cycles = t1 - t0;
    r.realtime = r1 - r0;
    r.hw_id = cur;
    r.xcc_id = 0;
    out[wave] = r;
  }
}

static std::vector<int> g_ks = {1, 2, 3, 4, 5, 6, 8};

template <int MODE>
static void run_mode(Rec *d_out, int n_cu, const char *filter) {
  if (filter && !strstr(modes[MODE].name, filter)) return;
  printf("\n== %s ==\n", modes[MODE].name);
  printf("%-11s %-20s %-24s %-22s %-10s\n", "waves/SIMD", "wave cyc per inst", "SIMD cyc per VALU inst", "SIMD cyc per any inst",
         "clock GHz");
  const int iters = 3000;
  for (int k : g_ks) {
    // blocks of 256 threads = 4 waves (one per SIMD of a CU); k blocks per CU
    const int blocks = n_cu * k;
    const int waves = blocks * 4;
    HIP_OK(hipMemset(d_out, 0, sizeof(Rec) * waves));
    for (int rep = 0; rep < 2; rep++) {  // first repetition warms the instruction cache / clocks
      hipLaunchKernelGGL(bench<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 1.0f);
      HIP_OK(hipDeviceSynchronize());
    }
    std::vector<Rec> h(waves);
    HIP_OK(hipMemcpy(h.data(), d_out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
    // how many waves shared each SIMD: key = (xcc, se, sh, cu, simd) from HW_ID
    std::map<uint64_t, std::vector<uint64_t>> per_simd;
    double clock_sum = 0;
    for (const Rec &r : h) {
      const uint64_t key = ((uint64_t)(r.xcc_id & 0xF) << 32) | (r.hw_id & 0xFFF0u & ~0xC0u);  // drop wave_id and pipe_id
      per_simd[key].push_back(r.cycles);
      clock_sum += (double)r.cycles / ((double)r.realtime * 10e-9) / 1e9;
    }
    double sum_cpi = 0, sum_simd_valu = 0, sum_simd_any = 0;
    int n_ok = 0, n_other = 0;
    const double body = 8.0 * iters;  // bodies executed per wave
    for (auto &kv : per_simd) {
      if ((int)kv.second.size() != k) { n_other++; continue; }  // only SIMDs that really held k waves
      uint64_t mx = 0;
      double mean = 0;
      for (uint64_t cc : kv.second) { mx = std::max(mx, cc); mean += (double)cc; }
      mean /= k;
      sum_cpi += mean / (body * modes[MODE].insts);
      if (modes[MODE].valu) sum_simd_valu += (double)mx / (body * modes[MODE].valu * k);
      sum_simd_any += (double)mx / (body * modes[MODE].insts * k);
      n_ok++;
    }
    if (n_ok == 0) { printf("%-11d no SIMD held exactly %d waves (%d SIMDs other)\n", k, k, n_other); continue; }
    printf("%-11d %-20.3f %-24.3f %-22.3f %-10.3f (%d SIMDs with %d waves, %d with another count)\n", k, sum_cpi / n_ok,
           sum_simd_valu / n_ok, sum_simd_any / n_ok, clock_sum / waves, n_ok, k, n_other);
  }
}

static void run_chase(Rec *d_out, int n_cu, const char *filter) {
  if (filter && !strstr("chase", filter)) return;
  printf("\n== dependent s_load chain (64-byte records, random cycle), cycles per load ==\n");
  printf("%-14s %-11s %-18s\n", "table bytes", "waves/SIMD", "cycles per load");
  for (uint32_t n_rec : {64u, 192u, 1024u, 4096u, 65536u}) {  // 4 KB, 12 KB, 64 KB, 256 KB, 4 MB
    std::vector<uint32_t> perm(n_rec), tab((size_t)n_rec * 16, 0);
    for (uint32_t i = 0; i < n_rec; i++) perm[i] = i;
    uint64_t st = 88172645463325252ull;
    for (uint32_t i = n_rec - 1; i > 0; i--) {  // one cycle through all records (Sattolo)
      st ^= st << 13; st ^= st >> 7; st ^= st << 17;
      std::swap(perm[i], perm[st % i]);
    }
    for (uint32_t i = 0; i < n_rec; i++) tab[(size_t)perm[i] * 16] = perm[(i + 1) % n_rec];
    uint32_t *d_tab;
    HIP_OK(hipMalloc(&d_tab, tab.size() * 4));
    HIP_OK(hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice));
    for (int k : {1, 4}) {
      const int blocks = n_cu * k, waves = blocks * 4, steps = 20000;
      for (int rep = 0; rep < 2; rep++) {
        hipLaunchKernelGGL(chase, dim3(blocks), dim3(256), 0, 0, d_tab, d_out, steps, n_rec);
        HIP_OK(hipDeviceSynchronize());
      }
      std::vector<Rec> h(waves);
      HIP_OK(hipMemcpy(h.data(), d_out, sizeof(Rec) * waves, hipMemcpyDeviceToHost));
      double s = 0;
      for (const Rec &r : h) s += (double)r.cycles / steps;
      printf("%-14u %-11d %-18.1f\n", n_rec * 64u, k, s / waves);
    }
    HIP_OK(hipFree(d_tab));
  }
}

int main(int argc, char **argv) {
  const char *filter = argc > 1 && strcmp(argv[1], "all") ? argv[1] : nullptr;
  if (argc > 2) {
    g_ks.clear();
    for (char *t = strtok(argv[2], ","); t; t = strtok(nullptr, ",")) g_ks.push_back(atoi(t));
  }
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s (%s), %d CUs, clock %d kHz\n", prop.name, prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
  printf("columns: 'wave cyc per inst'      = a wave's own cycles per instruction of its stream (s_memtime)\n"
         "         'SIMD cyc per VALU inst' = longest wave on the SIMD / VALU instructions issued by all its waves:\n"
         "                                    the SIMD's issue cost per wave64 VALU instruction of this kind\n");
  const int n_cu = prop.multiProcessorCount;
  Rec *d_out;
  HIP_OK(hipMalloc(&d_out, sizeof(Rec) * n_cu * 8 * 4));
  run_mode<M_FMA_V>(d_out, n_cu, filter);
  run_mode<M_FMA_S>(d_out, n_cu, filter);
  run_mode<M_FMAC_V>(d_out, n_cu, filter);
  run_mode<M_FMAC_S>(d_out, n_cu, filter);
  run_mode<M_MUL_S>(d_out, n_cu, filter);
  run_mode<M_ADD_V>(d_out, n_cu, filter);
  run_mode<M_MAX3_V>(d_out, n_cu, filter);
  run_mode<M_FMA_INL>(d_out, n_cu, filter);
  run_mode<M_PK_V>(d_out, n_cu, filter);
  run_mode<M_PK_S>(d_out, n_cu, filter);
  run_mode<M_PKMUL_S>(d_out, n_cu, filter);
  run_mode<M_CMP_VCC>(d_out, n_cu, filter);
  run_mode<M_CMP_S64>(d_out, n_cu, filter);
  run_mode<M_CMP_SOP>(d_out, n_cu, filter);
  run_mode<M_MOV_S>(d_out, n_cu, filter);
  run_mode<M_READLANE>(d_out, n_cu, filter);
  run_mode<M_CNDMASK_S>(d_out, n_cu, filter);
  run_mode<M_RCP>(d_out, n_cu, filter);
  run_mode<M_DEP>(d_out, n_cu, filter);
  run_mode<M_SALU>(d_out, n_cu, filter);
  run_mode<M_SALU64>(d_out, n_cu, filter);
  run_mode<M_FMA_SALU>(d_out, n_cu, filter);
  run_mode<M_MIX>(d_out, n_cu, filter);
  run_mode<M_KMIX>(d_out, n_cu, filter);
  run_mode<M_KMIX_NOSALU>(d_out, n_cu, filter);
  run_chase(d_out, n_cu, filter);
  HIP_OK(hipFree(d_out));
  return 0;
}
