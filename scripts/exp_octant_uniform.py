"""Experiment (round 4): OCTANT-UNIFORM BOX TEST.  When every lane that needs a mesh has the same three direction signs (85-97 % of all
node visits, scripts/gpu_oct_stats.py), which face of a box is the entry and which the exit plane is the same for all of them, so the
per-axis min(t1, t2) / max(t1, t2) of the slab test are not needed: 32 instead of 56 vector instructions per four-wide node visit, same
values.  This file is the form with eight statically routed copies of the box test behind a switch; the first form fetched the planes
through wave-uniform byte offsets (six s_load_dwordx4 with an SGPR offset).  Parity-exact (113 GPU tests with the library), and SLOWER:
bunny +3 ... +6 %, 64 000 triangles -0.7 ... +4 %, C3-deep +4 % (profiles/r04/exp_octant_uniform_ab.txt) — not adopted.
The shipped kernel source is NOT touched: this script patches a COPY under /tmp/defer and builds build_variants/<name>.so from it.
usage: exp_octant_uniform.py <name> [--asm]"""
import os, subprocess, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
from cutrace_amd import build
src = open(os.path.join(ROOT, "cutrace_amd/csrc/render_kernel.hip")).read()

def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old[:60])
    src = src.replace(old, new)

rep("""            const float2_ c_rxy = {ria.x, ria.y}, c_rzk = {ria.z, ka.x}, c_kyz = {ka.y, ka.z};
            const float2_ c_bxy = {kb.x, kb.y}, c_bz = {kb.z, 0.0f};
""", """            // Octant-uniform waves: when every lane that needs the mesh has the same three direction signs, which face of a box
            // is the entry and which the exit plane of an axis is the same for all of them — the node's planes are then FETCHED as
            // (entry, exit) instead of (min, max) (wave-uniform byte offsets into the record) and min(t1, t2) / max(t1, t2) per axis
            // are not computed at all: the same twelve products, then max3 / max and min3 / min per box.  The values are the
            // same as the general form's (rounding is monotonic: the entry product is never above the exit product), a NaN
            // still drops its axis.  bb_m only shrinks during the walk, so a wave uniform here stays uniform.
            const mask_t sg_x = BALLOT((int)__float_as_uint(rd.x) < 0) & bb_m, sg_y = BALLOT((int)__float_as_uint(rd.y) < 0) & bb_m,
                         sg_z = BALLOT((int)__float_as_uint(rd.z) < 0) & bb_m;
            const bool uni = (sg_x == 0ull || sg_x == bb_m) && (sg_y == 0ull || sg_y == bb_m) && (sg_z == 0ull || sg_z == bb_m);
            const bool fl_x = uni && nb_x != 0u, fl_y = uni && nb_y != 0u, fl_z = uni && nb_z != 0u;
            uint32_t octw = uni ? (nb_x | (nb_y << 1) | (nb_z << 2)) : 8u;
            asm volatile("" : "+s"(octw));
            const V3 kn = mk(fl_x ? kb.x : ka.x, fl_y ? kb.y : ka.y, fl_z ? kb.z : ka.z);
            const V3 kf = mk(fl_x ? ka.x : kb.x, fl_y ? ka.y : kb.y, fl_z ? ka.z : kb.z);
            const float2_ c_rxy = {ria.x, ria.y}, c_rzk = {ria.z, kn.x}, c_kyz = {kn.y, kn.z};
            const float2_ c_bxy = {kf.x, kf.y}, c_bz = {kf.z, 0.0f};
            typedef float f32x4_ __attribute__((ext_vector_type(4)));
            typedef uint32_t u32x8_ __attribute__((ext_vector_type(8)));
""")
rep("""            auto box_hits2 = [&](const CADDR DNode4 &N, int c, mask_t &ha, mask_t &hb) {
              float2_ t1x, t1y, t1z, t2x, t2y, t2z;
              PKFMA(t1x, ldpair2(&N.lo[0][c]), c_rxy, 0, c_rzk, 1);
              PKFMA(t1y, ldpair2(&N.lo[1][c]), c_rxy, 1, c_kyz, 0);
              PKFMA(t1z, ldpair2(&N.lo[2][c]), c_rzk, 0, c_kyz, 1);
              PKFMA(t2x, ldpair2(&N.hi[0][c]), c_rxy, 0, c_bxy, 0);
              PKFMA(t2y, ldpair2(&N.hi[1][c]), c_rxy, 1, c_bxy, 1);
              PKFMA(t2z, ldpair2(&N.hi[2][c]), c_rzk, 0, c_bz, 0);
              const float lo_a = slab_lo4(t1x.x, t2x.x, t1y.x, t2y.x, t1z.x, t2z.x, min_t);
              const float hi_a = slab_hi4(t1x.x, t2x.x, t1y.x, t2y.x, t1z.x, t2z.x, lim);
              const float lo_b = slab_lo4(t1x.y, t2x.y, t1y.y, t2y.y, t1z.y, t2z.y, min_t);
              const float hi_b = slab_hi4(t1x.y, t2x.y, t1y.y, t2y.y, t1z.y, t2z.y, lim);
              ha = bb_m & ~FCMP(lo_a, hi_a, FC_OGT);
              hb = bb_m & ~FCMP(lo_b, hi_b, FC_OGT);
            };
""", """            auto box_hits2 = [&](float2_ nx, float2_ ny, float2_ nz, float2_ fx, float2_ fy, float2_ fz, const bool uni_, mask_t &ha, mask_t &hb) {
              float2_ t1x, t1y, t1z, t2x, t2y, t2z;
              PKFMA(t1x, nx, c_rxy, 0, c_rzk, 1);
              PKFMA(t1y, ny, c_rxy, 1, c_kyz, 0);
              PKFMA(t1z, nz, c_rzk, 0, c_kyz, 1);
              PKFMA(t2x, fx, c_rxy, 0, c_bxy, 0);
              PKFMA(t2y, fy, c_rxy, 1, c_bxy, 1);
              PKFMA(t2z, fz, c_rzk, 0, c_bz, 0);
              float lo_a, hi_a, lo_b, hi_b;
              if (uni_) {
                CTR_MARK(64);
                float m;
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(t1x.x), "v"(t1y.x), "v"(t1z.x));
                asm("v_max_f32 %0, %1, %2" : "=v"(lo_a) : "v"(m), "v"(min_t));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(t2x.x), "v"(t2y.x), "v"(t2z.x));
                asm("v_min_f32 %0, %1, %2" : "=v"(hi_a) : "v"(m), "v"(lim));
                asm("v_max3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(t1x.y), "v"(t1y.y), "v"(t1z.y));
                asm("v_max_f32 %0, %1, %2" : "=v"(lo_b) : "v"(m), "v"(min_t));
                asm("v_min3_f32 %0, %1, %2, %3" : "=v"(m) : "v"(t2x.y), "v"(t2y.y), "v"(t2z.y));
                asm("v_min_f32 %0, %1, %2" : "=v"(hi_b) : "v"(m), "v"(lim));
              } else {
                CTR_MARK(65);
                lo_a = slab_lo4(t1x.x, t2x.x, t1y.x, t2y.x, t1z.x, t2z.x, min_t);
                hi_a = slab_hi4(t1x.x, t2x.x, t1y.x, t2y.x, t1z.x, t2z.x, lim);
                lo_b = slab_lo4(t1x.y, t2x.y, t1y.y, t2y.y, t1z.y, t2z.y, min_t);
                hi_b = slab_hi4(t1x.y, t2x.y, t1y.y, t2y.y, t1z.y, t2z.y, lim);
              }
              CTR_MARK(66);
              ha = bb_m & ~FCMP(lo_a, hi_a, FC_OGT);
              hb = bb_m & ~FCMP(lo_b, hi_b, FC_OGT);
            };
""")
rep("""              const CADDR DNode4 &N = nodes4[cur];
              mask_t h0, h1, h2, h3;
              box_hits2(N, 0, h0, h1);
              box_hits2(N, 2, h2, h3);
""", """              const CADDR DNode4 &N = nodes4[cur];
              mask_t h0, h1, h2, h3;
#define CTR_PL(a, c) ldpair2(&N.lo[a][c])
#define CTR_PH(a, c) ldpair2(&N.hi[a][c])
#define CTR_BOX4(FX, FY, FZ)                                                                                              \\
  box_hits2(FX ? CTR_PH(0, 0) : CTR_PL(0, 0), FY ? CTR_PH(1, 0) : CTR_PL(1, 0), FZ ? CTR_PH(2, 0) : CTR_PL(2, 0),          \\
            FX ? CTR_PL(0, 0) : CTR_PH(0, 0), FY ? CTR_PL(1, 0) : CTR_PH(1, 0), FZ ? CTR_PL(2, 0) : CTR_PH(2, 0), true, h0, h1); \\
  box_hits2(FX ? CTR_PH(0, 2) : CTR_PL(0, 2), FY ? CTR_PH(1, 2) : CTR_PL(1, 2), FZ ? CTR_PH(2, 2) : CTR_PL(2, 2),          \\
            FX ? CTR_PL(0, 2) : CTR_PH(0, 2), FY ? CTR_PL(1, 2) : CTR_PH(1, 2), FZ ? CTR_PL(2, 2) : CTR_PH(2, 2), true, h2, h3);
              switch (octw) {
                case 0: { CTR_BOX4(0, 0, 0) } break;
                case 1: { CTR_BOX4(1, 0, 0) } break;
                case 2: { CTR_BOX4(0, 1, 0) } break;
                case 3: { CTR_BOX4(1, 1, 0) } break;
                case 4: { CTR_BOX4(0, 0, 1) } break;
                case 5: { CTR_BOX4(1, 0, 1) } break;
                case 6: { CTR_BOX4(0, 1, 1) } break;
                case 7: { CTR_BOX4(1, 1, 1) } break;
                default:
                  box_hits2(CTR_PL(0, 0), CTR_PL(1, 0), CTR_PL(2, 0), CTR_PH(0, 0), CTR_PH(1, 0), CTR_PH(2, 0), false, h0, h1);
                  box_hits2(CTR_PL(0, 2), CTR_PL(1, 2), CTR_PL(2, 2), CTR_PH(0, 2), CTR_PH(1, 2), CTR_PH(2, 2), false, h2, h3);
              }
""")
os.makedirs("/tmp/defer/osrc4", exist_ok=True)
open("/tmp/defer/osrc4/render_kernel.hip", "w").write(src)
name = sys.argv[1]
flags = sys.argv[2:]
srcs = ["/tmp/defer/osrc4/render_kernel.hip"] + build.HIP_SRCS[1:]
out = os.path.join(ROOT, "build_variants", name + ".so")
cmd = [build.hipcc(), *build.HIP_FLAGS, *flags, "-shared", "-o", out, *srcs, "-ldl"]
if "--asm" in flags:
    flags.remove("--asm")
    cmd = [build.hipcc(), *build.HIP_FLAGS, *flags, "-S", "--cuda-device-only", "-o", "/tmp/defer/" + name + ".s", srcs[0]]
subprocess.check_call(cmd)
print("built", name)
