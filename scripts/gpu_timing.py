"""Where a wave spends its life: run with CUTRACE_AMD_LIB=<lib built with -DCTR_TIMING>."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5)]
d = tempfile.mkdtemp()
todo.append(("dense64k", scenes.make_dense_bunny(d, 3), 5))
if "--more" in sys.argv:
    todo += [("c3deep", scenes.make_mirror_deep(d), 8), ("c4", scenes.make_bunny_grid(d), 5)]
for name, path, b in todo:
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    for _ in range(3):
        r = ds.render(bounces=b)
    print(name, "kernel_ms", round(r["kernel_ms"], 3), flush=True)
