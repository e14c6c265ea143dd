import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
for name, b in (("bunny", 5), ("mirror", 8), ("sphere_plane", 5)):
    s = ca.HostScene.load(f"scene/{name}.json")
    ds = ca.DeviceScene(s)
    ds.render(bounces=b)
    for v in (0, 8, 16):
        ds.set_variant(v)
        r = ds.render(bounces=b)
        print(name, "variant", v, "kernel_ms", round(r["kernel_ms"], 3), "rays", r["ray_count"], flush=True)
