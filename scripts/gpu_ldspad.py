"""Where does extra LDS per wave start to cost occupancy?  (CUTRACE_LDS_PAD sweep on bunny.json: 5120 B of stack per wave.)"""
import os, statistics, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
s = ca.HostScene.load("scene/bunny.json")
ds = ca.DeviceScene(s)
for pad in (0, 256, 512, 768, 1024, 1280, 1536, 1792, 2048, 2560, 3072, 0):
    os.environ["CUTRACE_LDS_PAD"] = str(pad)
    for _ in range(4): ds.render(bounces=5)
    t = statistics.median(ds.render(bounces=5)["kernel_ms"] for _ in range(9))
    print(f"pad {pad:5d} -> {5120 + pad} B/wave: {t:.4f} ms", flush=True)
