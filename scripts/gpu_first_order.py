"""First launch of a shape: tiles dispatched centre-out (default) against image order (CTR_VAR_IMAGE_ORDER_FIRST), kernel ms,
fresh scene handle each time, device buffers + DMA so that kernel_ms is the render kernel (+ the order kernel) alone."""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5),
        ("dense64k", scenes.make_dense_bunny(d, 3), 5), ("c3deep", scenes.make_mirror_deep(d), 8)]
if "--c4" in sys.argv:
    todo.append(("c4", scenes.make_bunny_grid(d), 5))
for name, path, b in todo:
    s = ca.HostScene.load(path)
    res = {}
    for label, var in (("image order", ca.VAR_IMAGE_ORDER_FIRST), ("centre-out", 0)):
        ts = []
        for rep in range(5):
            ds = ca.DeviceScene(s)
            ds.set_variant(var | ca.VAR_NO_DIRECT)
            ds.render(bounces=b, rows=(0, 8))      # another shape: code object, clocks
            ts.append(ds.render(bounces=b, pinned=True)["kernel_ms"])
            ds.close()
        res[label] = statistics.median(ts)
    ds = ca.DeviceScene(s)
    for _ in range(4):
        ds.render(bounces=b)
    steady = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(5))
    print(f"{name:13s} first launch: image order {res['image order']:.3f} ms, centre-out {res['centre-out']:.3f} ms ({100 * (res['centre-out'] / res['image order'] - 1):+.1f} %); steady {steady:.3f}", flush=True)
