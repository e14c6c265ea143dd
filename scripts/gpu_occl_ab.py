"""Same-box A/B of the occluder maps (default) against CTR_VAR_NO_OCCLUDER_MAP: kernel ms, median of alternating runs, bitwise check."""
import sys, os, statistics, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
cases = [("C2 bunny@1080p", "scene/bunny.json", 1920, 1080, 5), ("C2-dense", scenes.make_dense_bunny(gen, 3), 1920, 1080, 5),
         ("C3 mirror b8", "scene/mirror.json", 1920, 1080, 8), ("C3-deep b8", scenes.make_mirror_deep(gen), 1920, 1080, 8),
         ("C4 grid @4096x4096", scenes.make_bunny_grid(gen, 4), 4096, 4096, 5)]
for name, path, w, h, b in cases:
    s = ca.HostScene.load(path); s.set_size(w, h)
    res, frames, scene = {}, {}, {}
    for tag, var in (("map", 0), ("no_map", ca.VAR_NO_OCCLUDER_MAP)):
        ds = ca.DeviceScene(s); ds.set_variant(var)
        frames[tag] = ds.render(bounces=b)
        for _ in range(3): ds.render(bounces=b)
        scene[tag] = ds; res[tag] = []
    for rep in range(9):
        for tag in ("map", "no_map"):
            res[tag].append(scene[tag].render(bounces=b)["kernel_ms"])
    same = all(np.array_equal(frames["map"][k].view(np.uint32), frames["no_map"][k].view(np.uint32)) for k in ("depth", "normal", "color"))
    a, n = statistics.median(res["map"]), statistics.median(res["no_map"])
    print(json.dumps({"config": name, "map_ms": round(a, 4), "no_map_ms": round(n, 4), "ratio": round(a / n, 4), "bitwise_equal": same}), flush=True)
