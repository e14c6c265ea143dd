"""Tile scheduling on/off on every config: kernel ms of the 1st..5th launch (order feedback kicks in at the 2nd)."""
import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
cfgs = [("bunny", "scene/bunny.json", 5, None), ("sphere_plane", "scene/sphere_plane.json", 5, None),
        ("mirror", "scene/mirror.json", 8, None), ("deep", scenes.make_mirror_deep(gen), 8, None),
        ("dense64k", scenes.make_dense_bunny(gen, 3), 5, None), ("grid4096", scenes.make_bunny_grid(gen, 4, 4), 5, (4096, 4096))]
for name, path, b, size in cfgs:
    s = ca.HostScene.load(path)
    if size: s.set_size(*size)
    ref = None
    for var in (ca.VAR_NO_REORDER, 0):
        ds = ca.DeviceScene(s)
        ds.set_variant(var)
        ts = []
        for i in range(6):
            r = ds.render(bounces=b)
            ts.append(round(r["kernel_ms"], 3))
        if ref is None: ref = r
        else:
            same = all(np.array_equal(ref[k].view(np.uint32), r[k].view(np.uint32)) for k in ("depth", "color", "normal")) and ref["ray_count"] == r["ray_count"]
            print("   bitwise equal to image order:", same, flush=True)
        print(f"{name:14s} {'image-order' if var else 'cost-order '} kernel_ms {ts}", flush=True)
