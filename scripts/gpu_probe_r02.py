"""Round-2 probes: occupancy sensitivity (extra LDS caps waves per CU) and the frame without its mesh."""
import json, os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca

def med(ds, b=5, n=7):
    ds.render(bounces=b); ds.render(bounces=b); ds.render(bounces=b)
    return round(statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(n)), 4)

s = ca.HostScene.load("scene/bunny.json")
for pad, label in ((0, "16 waves/CU (default)"), (3000, "15"), (4000, "13"), (6000, "11->12?"), (8500, "10"), (13000, "8"), (20000, "5-6")):
    os.environ["CUTRACE_LDS_PAD"] = str(pad)
    ds = ca.DeviceScene(s)
    lds = 7680 + pad
    print(f"bunny pad={pad:6d} lds/wave={lds:6d} waves/CU<={min(16, 163840 // lds):2d}  kernel_ms={med(ds)}", flush=True)
os.environ["CUTRACE_LDS_PAD"] = "0"
# the same room without the bunny: what every cast pays for planes + continuation + shading
j = json.load(open("scene/bunny.json"))
j["objects"] = [o for o in j["objects"] if o["type"] != "mesh"]
d = tempfile.mkdtemp()
p = os.path.join(d, "room.json")
json.dump(j, open(p, "w"))
s2 = ca.HostScene.load(p)
ds2 = ca.DeviceScene(s2)
print("room without mesh: kernel_ms", med(ds2), "rays", ds2.render()["ray_count"], flush=True)
ds2.set_variant(ca.VAR_NO_REORDER)
print("room without mesh, image order: kernel_ms", med(ds2), flush=True)
