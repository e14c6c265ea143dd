"""How many of a wave's 64 lanes are alive, trip by trip and by what the lane casts for:
CTR_VAR_STATS renders of every single-GPU config + C4 -> one JSON line per config (ctr_debug_lane_stats)."""
import sys, os, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
cases = [
    ("C1 sphere_plane@1080p", "scene/sphere_plane.json", 1920, 1080, 5),
    ("C2 bunny@1080p", "scene/bunny.json", 1920, 1080, 5),
    ("C3 mirror@1080p b8", "scene/mirror.json", 1920, 1080, 8),
    ("C3-deep b8", scenes.make_mirror_deep(gen), 1920, 1080, 8),
]
if "--c4" in sys.argv:
    cases.append(("C4 grid @4096x4096", scenes.make_bunny_grid(gen, 4), 4096, 4096, 5))
for name, path, w, h, b in cases:
    s = ca.HostScene.load(path)
    assert s.ok
    s.set_size(w, h)
    ds = ca.DeviceScene(s)
    t = ds.render(bounces=b)["kernel_ms"]
    t = min(ds.render(bounces=b)["kernel_ms"] for _ in range(3))
    ds.set_variant(ca.VAR_STATS)
    ca.DeviceScene.lane_stats(reset=True)
    ds.render(bounces=b)
    st = ca.DeviceScene.lane_stats(reset=True)
    st["config"] = name
    st["kernel_ms_default_build"] = round(t, 4)
    for v in st["by_kind"].values():
        v["lanes_per_trip"] = round(v["lanes_per_trip"], 2)
    print(json.dumps(st), flush=True)
    ds.close()
