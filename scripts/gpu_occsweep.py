"""Occupancy sweep of the shipped kernel: extra dynamic LDS per wave (CUTRACE_LDS_PAD) caps the waves resident per CU.
How the frame time moves with 2..6 waves per SIMD says how much of it is latency that more waves would hide."""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("dense64k", scenes.make_dense_bunny(d, 3), 5)]
if "--c4" in sys.argv: todo.append(("c4", scenes.make_bunny_grid(d), 5))
for name, path, b in todo:
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    base = 5 * 4 * 256  # stack bytes per wave at bounces 5, 4 dwords per frame
    for waves_cu in (24, 20, 16, 12, 8):
        per = 160 * 1024 // waves_cu
        per -= per % 256
        os.environ["CUTRACE_LDS_PAD"] = str(max(per - base, 0))
        for _ in range(4): ds.render(bounces=b)
        t = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(7))
        print(f"{name:10s} <= {waves_cu} waves/CU ({waves_cu / 4:.0f}/SIMD): {t:.3f} ms", flush=True)
    os.environ.pop("CUTRACE_LDS_PAD")
