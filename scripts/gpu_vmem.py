import sys, os, tempfile, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
paths = [("bunny1k", "scene/bunny.json")] + [(f"sub{r}", scenes.make_dense_bunny(gen, r)) for r in (1, 2, 3)] + [("grid4096", scenes.make_bunny_grid(gen, 4))]
for name, p in paths:
    s = ca.HostScene.load(p)
    ds = ca.DeviceScene(s)
    res = {}
    for vname, v in (("smem", ca.VAR_SMEM), ("vmem", ca.VAR_VMEM), ("auto", 0)):
        ds.set_variant(v)
        r0 = ds.render()
        t = statistics.median(ds.render()["kernel_ms"] for _ in range(3))
        res[vname] = (round(t, 3), r0)
    same = all(np.array_equal(res["smem"][1][k].view(np.uint32), res["vmem"][1][k].view(np.uint32)) for k in ("depth", "normal", "color"))
    print(name, s.desc.contents.n_triangles, {k: v[0] for k, v in res.items()}, "bitwise smem==vmem:", same, flush=True)
