"""Same-box A/B of the merged walk (ONE tree over all meshes' triangles) against the two-level walk (CTR_VAR_NO_MERGE) on the
multi-mesh configs: C4 (16 meshes @4096x4096), mirror.json b8 (3 meshes), C3-deep b8; kernel ms, median of alternating runs."""
import sys, os, statistics, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
cases = [("C4 grid 4x4 @4096x4096", scenes.make_bunny_grid(gen, 4), 4096, 4096, 5),
         ("C3 mirror@1080p b8", "scene/mirror.json", 1920, 1080, 8),
         ("C3-deep b8", scenes.make_mirror_deep(gen), 1920, 1080, 8)]
for name, path, w, h, b in cases:
    s = ca.HostScene.load(path)
    s.set_size(w, h)
    res = {}
    frames = {}
    scene = {}
    for tag, var in (("merged", ca.VAR_MERGE), ("two_level", 0)):
        ds = ca.DeviceScene(s)
        ds.set_variant(var)
        frames[tag] = ds.render(bounces=b)
        for _ in range(3):
            ds.render(bounces=b)
        scene[tag] = ds
        res[tag] = []
    for rep in range(7):
        for tag in ("merged", "two_level"):
            res[tag].append(scene[tag].render(bounces=b)["kernel_ms"])
    same = all(np.array_equal(frames["merged"][k].view(np.uint32), frames["two_level"][k].view(np.uint32)) for k in ("depth", "normal", "color"))
    out = {"config": name, "merged_ms": round(statistics.median(res["merged"]), 4), "two_level_ms": round(statistics.median(res["two_level"]), 4),
           "bitwise_equal": same, "rays": frames["merged"]["ray_count"]}
    out["ratio"] = round(out["merged_ms"] / out["two_level_ms"], 4)
    ds = scene["merged"]
    ds.set_variant(ca.VAR_STATS | ca.VAR_MERGE)
    ca.DeviceScene.lane_stats(reset=True)
    ds.render(bounces=b)
    st = ca.DeviceScene.lane_stats(reset=True)
    c = [int(x) for x in ds.last_counters()]
    out["merged_stats"] = {"wave_casts": c[4], "bvh_nodes": c[5], "tri_prefilters": c[6], "tri_exact": c[7], "mesh_entries": c[8],
                           "merged_walks": st["merged_walks"], "redone": st["merged_walks_redone"]}
    ds = scene["two_level"]
    ds.set_variant(ca.VAR_STATS)
    ds.render(bounces=b)
    c = [int(x) for x in ds.last_counters()]
    out["two_level_stats"] = {"wave_casts": c[4], "bvh_nodes": c[5], "tri_prefilters": c[6], "tri_exact": c[7], "mesh_entries": c[8]}
    print(json.dumps(out), flush=True)
