"""Full-size timings of every BASELINE.json config on one GPU (median of 5 kernel_ms), for DESIGN.md."""
import sys, os, statistics, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
cases = [
    ("C0 triangle@128x128", "scene/triangle.json", 128, 128, 5),
    ("C1 sphere_plane@1080p", "scene/sphere_plane.json", 1920, 1080, 5),
    ("C2 bunny@1080p (1000 tris)", "scene/bunny.json", 1920, 1080, 5),
    ("C2-dense bunny 64k tris@1080p", scenes.make_dense_bunny(gen, 3), 1920, 1080, 5),
    ("C3 mirror@1080p b8", "scene/mirror.json", 1920, 1080, 8),
    ("C3-deep mirror (walls reflect 0.5) b8", scenes.make_mirror_deep(gen), 1920, 1080, 8),
    ("C4 bunny grid 4x4 @4096x4096", scenes.make_bunny_grid(gen, 4), 4096, 4096, 5),
]
for name, path, w, h, b in cases:
    s = ca.HostScene.load(path)
    assert s.ok
    s.set_size(w, h)
    ds = ca.DeviceScene(s)
    r = ds.render(bounces=b)
    t = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(5))
    print(json.dumps({"config": name, "kernel_ms": round(t, 3), "rays": r["ray_count"], "mrays_s": round(r["ray_count"] / t / 1e3, 1),
                      "total_ms_incl_d2h": round(r["total_ms"], 1)}), flush=True)
