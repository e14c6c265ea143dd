"""The build for 6 waves per SIMD (KV_OCC6) against the default 5, per config, on one box: which scenes gain?"""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny 1000 tris", "scene/bunny.json", 5), ("mirror 924 tris b8", "scene/mirror.json", 8),
        ("dense 4k tris", scenes.make_dense_bunny(d, 1), 5), ("dense 16k tris", scenes.make_dense_bunny(d, 2), 5),
        ("dense 64k tris", scenes.make_dense_bunny(d, 3), 5), ("c3deep b8", scenes.make_mirror_deep(d), 8),
        ("C4 16x1000 @4096^2", scenes.make_bunny_grid(d), 5)]
for name, path, b in todo:
    s = ca.HostScene.load(path)
    out = []
    for label, var in (("5 waves", ca.VAR_NO_OCC6), ("6 waves", 0)):
        os.environ["CUTRACE_OCC6_MIN_TRIS"] = "1"
        ds = ca.DeviceScene(s)
        ds.set_variant(var)
        for _ in range(4):
            ds.render(bounces=b)
        out.append(statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(9)))
        ds.close()
    print(f"{name:22s} 5 waves {out[0]:.3f} ms, 6 waves {out[1]:.3f} ms  ({(out[1] / out[0] - 1) * 100:+.1f} %)", flush=True)
