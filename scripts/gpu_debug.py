import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
p = scenes.make_mirror_deep(tempfile.mkdtemp(), width=160, height=90)
s = ca.HostScene.load(p)
import oracle
o = oracle.oracle_render(s, bounces=8, threads=16)
ds = ca.DeviceScene(s)
for name, v in (("auto", 0), ("no_bvh", 8), ("no_anyhit", 4), ("no_prefilter", 2), ("exactpow", 32), ("none", 2 | 4 | 8 | 32)):
    ds.set_variant(v)
    r = ds.render(bounces=8)
    d = np.abs(r["color"].astype(np.float64) - o["color"].astype(np.float64)).max(-1)
    bad = np.argwhere(d > 1e-4)
    print(name, "bad px:", bad.tolist()[:5], "max", d.max(), "rays", r["ray_count"], o["ray_count"], flush=True)
