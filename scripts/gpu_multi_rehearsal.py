"""ctr_render_multi on ONE GPU: groups that list device 0 several times (peer-copy transport).  Says nothing about
multi-GPU speed — every part runs on the same device — but shows what the extra steps cost there: kernel ms of the
slowest part, and the whole call (render + gather copies + re-interleave + one D2H) against ctr_render."""
import os, statistics, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
for name, path, b in (("bunny@1080p", "scene/bunny.json", 5), ("C4 4x4 grid@4096^2", scenes.make_bunny_grid(d), 5)):
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    for _ in range(3):
        ds.render(bounces=b)
    one = statistics.median(ds.render(bounces=b)["total_ms"] for _ in range(5))
    k1 = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(5))
    ds.close()
    print(f"{name}: ctr_render kernel {k1:.2f} ms, call {one:.2f} ms (pageable destination)", flush=True)
    for n in (1, 2, 4, 8):
        m = ca.MultiScene(s, [0] * n)
        for _ in range(3):
            m.render(bounces=b)
        rr = [m.render(bounces=b) for _ in range(5)]
        tot = statistics.median(r["total_ms"] for r in rr)
        per = [round(x, 2) for x in rr[-1]["kernel_ms_per_device"]]
        print(f"   group of {n} x device 0 ({m.transport}): call {tot:.2f} ms; per-part kernel ms (all on one GPU, overlapping) {per}", flush=True)
        m.close()
