import sys, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
gen = tempfile.mkdtemp()
for rounds in (1, 2, 3):
    s = ca.HostScene.load(scenes.make_dense_bunny(gen, rounds))
    ds = ca.DeviceScene(s)
    ds.render()
    for v in (0, 16):
        ds.set_variant(v)
        r = ds.render()
        print("rounds", rounds, "tris", s.desc.contents.n_triangles, "variant", v, "kernel_ms", round(r["kernel_ms"], 3), flush=True)
