"""Quick GPU check used while iterating on the kernel: parity of the three shipped scenes against the golden
fixtures / oracle at small size, default kernel vs. all shortcuts off at 1080p (bitwise), then kernel ms."""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
import oracle
from cutrace_amd import scenes
from tests.util import assert_parity, same_bits

d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5),
        ("dense64k", scenes.make_dense_bunny(d, 3), 5)]
if "--more" in sys.argv:
    todo += [("c3deep", scenes.make_mirror_deep(d), 8), ("c4", scenes.make_bunny_grid(d), 5)]
bad = 0
for name, path, b in todo:
    s = ca.HostScene.load(path)
    w, h = s.size
    if name != "c4":
        s.set_size(96, 54)
        ds = ca.DeviceScene(s)
        r = ds.render(bounces=b)
        o = oracle.oracle_render(s, bounces=b, threads=os.cpu_count() or 4)
        try:
            assert_parity(r, o, what=name)
            assert r["ray_count"] == o["ray_count"]
        except AssertionError as e:
            bad += 1
            print("PARITY FAIL", name, str(e)[:300], flush=True)
        ds.close()
        s.set_size(w, h)
    ds = ca.DeviceScene(s)
    r0 = ds.render(bounces=b)
    first = r0["kernel_ms"]
    for _ in range(3):
        ds.render(bounces=b)
    t = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(7))
    msg = f"{name:13s} {w}x{h} b{b}: first launch {first:.3f} ms, steady {t:.3f} ms, rays {r0['ray_count']}"
    if "--bitwise" in sys.argv and name != "c4":
        ds.set_variant(ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT | ca.VAR_EXACT_POW)
        plain = ds.render(bounces=b)
        ds.set_variant(ca.VAR_EXACT_POW)
        fast = ds.render(bounces=b)
        ok = all(same_bits(fast[k], plain[k]) for k in ("depth", "normal", "color")) and fast["ray_count"] == plain["ray_count"]
        msg += "  accel on/off bitwise: " + ("ok" if ok else "MISMATCH")
        bad += 0 if ok else 1
    print(msg, flush=True)
print("BAD" if bad else "ALL OK", flush=True)
sys.exit(1 if bad else 0)
