"""Renders bunny.json x30, the 64 000-triangle mesh x16, C3-deep x16 and the C4 grid x8 through ctr_render (after 4 warm-up frames each) — the workload of
scripts/gpu_ab_cycles.sh, which counts the kernel's CYCLES per dispatch (a box's clock moves by several % within a call; its cycle count much less)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
for path, b, n in ((os.path.join(ROOT, "scene/bunny.json"), 5, 30), (scenes.make_dense_bunny(d, 3), 5, 16), (scenes.make_mirror_deep(d), 8, 16), (scenes.make_bunny_grid(d), 5, 8)):
    ds = ca.DeviceScene(ca.HostScene.load(path))
    for _ in range(4 + n): ds.render(bounces=b)
    ds.close()
