"""Experiment (round 4): "touch" loads in the node visit — a pushed node's two lines requested when it is pushed (t1), the first record of
every child that is a leaf requested as soon as the node's descriptors are there (t2) — into xnack_mask_lo, a register nothing reads, so
no wait is needed.  Both lose (profiles/r04/exp_scalar_diet_ab.txt): every scalar instruction added to a node visit costs.
The shipped kernel source is NOT touched: a COPY is patched under /tmp/defer and built into build_variants/<name>.so.
usage: exp_touch.py name [t1] [t2]"""
import os, subprocess, sys
ROOT = "/root/repo"
sys.path.insert(0, ROOT)
from cutrace_amd import build
src = open(os.path.join(ROOT, "cutrace_amd/csrc/render_kernel.hip")).read()
name = sys.argv[1]
opts = set(sys.argv[2:])
def rep(old, new, count=1):
    global src
    assert src.count(old) == count, (src.count(old), old[:60])
    src = src.replace(old, new)
if "t1" in opts:   # a pushed node's two lines are requested when it is pushed
    rep("""            auto push = [&](uint32_t d) {
""", """            auto push = [&](uint32_t d) {
              asm volatile("s_load_dword xnack_mask_lo, %0, %1\\n\\ts_load_dword xnack_mask_lo, %0, %1 offset:0x40" :: "s"(nodes4), "s"(d << 7));
""")
if "t2" in opts:   # the first triangle record of every child that is a leaf is requested as soon as the descriptors are there
    rep("""              const uint32_t d0 = N.child[0], d1 = N.child[1], d2 = N.child[2], d3 = N.child[3], n_axis_ = N.axis;
""", """              const uint32_t d0 = N.child[0], d1 = N.child[1], d2 = N.child[2], d3 = N.child[3], n_axis_ = N.axis;
              asm volatile("s_load_dword xnack_mask_lo, %0, %1\\n\\ts_load_dword xnack_mask_lo, %0, %2\\n\\ts_load_dword xnack_mask_lo, %0, %3\\n\\ts_load_dword xnack_mask_lo, %0, %4"
                           :: "s"(A.tris + beg), "s"((d0 & 0xFFFFFFu) << 6), "s"((d1 & 0xFFFFFFu) << 6), "s"((d2 & 0xFFFFFFu) << 6), "s"((d3 & 0xFFFFFFu) << 6));
""")
d = "/tmp/defer/tsrc_" + name
os.makedirs(d, exist_ok=True)
open(d + "/render_kernel.hip", "w").write(src)
srcs = [d + "/render_kernel.hip"] + build.HIP_SRCS[1:]
out = os.path.join(ROOT, "build_variants", name + ".so")
subprocess.check_call([build.hipcc(), *build.HIP_FLAGS, "-shared", "-o", out, *srcs, "-ldl"])
subprocess.check_call([build.hipcc(), *build.HIP_FLAGS, "-S", "--cuda-device-only", "-o", "/tmp/defer/" + name + ".s", srcs[0]])
print("built", name)
