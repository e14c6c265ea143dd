"""How many shadow wave casts a per-light occluder-depth map could spare the mesh phase, at best (STATS counters [80..83])."""
import sys, os, json, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
for name, path, w, h, b in (("C2 bunny", "scene/bunny.json", 1920, 1080, 5), ("C2-dense", scenes.make_dense_bunny(gen, 3), 1920, 1080, 5),
                            ("C3 mirror b8", "scene/mirror.json", 1920, 1080, 8), ("C4", scenes.make_bunny_grid(gen, 4), 4096, 4096, 5)):
    s = ca.HostScene.load(path); s.set_size(w, h)
    ds = ca.DeviceScene(s); ds.set_variant(ca.VAR_STATS)
    ca.DeviceScene.lane_stats(reset=True)
    ds.render(bounces=b)
    st = ca.DeviceScene.lane_stats(reset=True)
    c = [int(x) for x in ds.last_counters()]
    print(json.dumps({"config": name, "wave_casts": st["wave_trips"], "shadow": st["shadow_casts_at_meshes"], "mesh_entries": c[8], "nodes": c[5], "prefilters": c[6]}), flush=True)
