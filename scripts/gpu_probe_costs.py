"""For the study of a first-launch tile order from a low-resolution probe render: per-wave costs of the frame at 1/8 of
the resolution (one ray per full-size tile; a wave = a block of 8x8 tiles) at several bounce counts, and the per-tile
costs of the full frame, saved to gpurun_out/probe_<scene>.npz."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5),
        ("dense64k", scenes.make_dense_bunny(d, 3), 5), ("c3deep", scenes.make_mirror_deep(d), 8)]
for name, path, b in todo:
    s = ca.HostScene.load(path)
    w, h = s.size
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_NO_DIRECT)
    out = {}
    ds.render(bounces=b)
    ds.render(bounces=b)
    out["full"] = ds.tile_costs()
    out["full_ms"] = ds.render(bounces=b)["kernel_ms"]
    ds.set_size((w + 7) // 8, (h + 7) // 8)
    for pb in (0, 1, 2, b):
        r = ds.render(bounces=pb)          # first launch of this (shape, bounces)... same shape: order from the previous one
        r = ds.render(bounces=pb)
        out[f"probe_b{pb}"] = ds.tile_costs()
        out[f"probe_b{pb}_ms"] = r["kernel_ms"]
        print(name, "probe bounces", pb, "kernel_ms", round(r["kernel_ms"], 4), "waves", out[f"probe_b{pb}"].size, flush=True)
    np.savez(f"gpurun_out/probe_{name}.npz", w=w, h=h, **out)
    ds.close()
