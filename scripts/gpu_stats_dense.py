import sys, os
sys.path.insert(0, os.getcwd())
import cutrace_amd as ca
for path in ("scene/bunny.json", "build_variants/scenes/bunny_dense3.json"):
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    ds.render(bounces=5)
    ds.set_variant(ca.VAR_STATS)
    ds.render(bounces=5)
    c = [int(x) for x in ds.last_counters()]
    print(path, dict(casts=c[4], nodes=c[5], prefilters=c[6], exact=c[7], mesh_entries=c[8], active_lanes=c[9], node_lanes=c[10], pf_lanes=c[11], exact_lanes=c[12]))
