"""How far is "the reference compiled for the host with every operation rounded once" from a build that may fuse a*b+c?
(VERDICT r02 "What's weak" 1(i): parity against a real CUDA run, nvcc default --fmad=true, is undefined here; CPU only.)

The parity target is the reference's headers compiled for the host with -ffp-contract=off.  nvcc contracts
multiply-add pairs into FMAs by default; WHICH pairs is the compiler's choice, so no host build reproduces a CUDA binary.
This script renders every config with the parity target and with a second build of the same headers where g++ may
contract (-ffp-contract=fast -mfma, and fminf/fmaxf semantics for min/max) and reports how the two frames differ: it
bounds what a maintainer should expect between this library's output and their CUDA binary's, it does not move the target.
usage: python scripts/cuda_fmad_gap.py > profiles/r03/cuda_fmad_gap.txt"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
import oracle
from cutrace_amd import scenes

NT = os.cpu_count() or 4
d = tempfile.mkdtemp()
todo = [("C0 triangle.json@128x128", "scene/triangle.json", (128, 128), 5),
        ("C1 sphere_plane.json@960x540", "scene/sphere_plane.json", (960, 540), 5),
        ("C2 bunny.json@480x270", "scene/bunny.json", (480, 270), 5),
        ("C3 mirror.json@960x540 b8", "scene/mirror.json", (960, 540), 8),
        ("C3-deep @240x135 b8", scenes.make_mirror_deep(d), (240, 135), 8),
        ("C4 4x4 bunny grid @192x192", scenes.make_bunny_grid(d), (192, 192), 5)]
print("parity target (reference headers, host, every operation rounded once) against the same headers with contraction allowed")
print("(g++ -ffp-contract=fast -mfma: 324 fused multiply-adds in the library; fminf/fmaxf semantics for min/max):")
for name, path, size, b in todo:
    s = ca.HostScene.load(path)
    s.set_size(*size)
    a = oracle.ref_render(s, bounces=b, threads=NT)
    c = oracle.ref_fmad_render(s, bounces=b, threads=NT)
    w, h = s.size
    bits = (a["depth"].view(np.uint32) != c["depth"].view(np.uint32)) | \
           (a["normal"].view(np.uint32) != c["normal"].view(np.uint32)).any(-1) | \
           (a["color"].view(np.uint32) != c["color"].view(np.uint32)).any(-1)
    fin = np.isfinite(a["depth"]) & np.isfinite(c["depth"])
    rel = np.zeros_like(a["depth"], dtype=np.float64)
    rel[fin] = np.abs(a["depth"][fin].astype(np.float64) - c["depth"][fin]) / np.maximum(np.abs(a["depth"][fin]), 1e-30)
    dcol = np.abs(a["color"].astype(np.float64) - c["color"]).max(-1)
    other = a["hit_ids"] != c["hit_ids"] if "hit_ids" in a and a["hit_ids"] is not None else np.zeros((h, w), bool)
    print(f"  {name:32s} {int(bits.sum()):7d} of {w * h:7d} pixels differ in some bit; depth: max relative {rel.max():.2e}, "
          f"{int((rel > 1e-5).sum())} pixels > 1e-5; colour: max {dcol.max():.2e}, {int((dcol > 1e-4).sum())} pixels > 1e-4, "
          f"{int((dcol > 1e-2).sum())} > 1e-2; another object hit first: {int(other.sum())}; rays {a['ray_count']} vs {c['ray_count']}", flush=True)
