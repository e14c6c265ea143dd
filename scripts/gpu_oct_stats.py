"""How many mesh entries / BVH node visits are made by waves whose lanes all share the three direction signs (experiment build with
counters 84..87 of ctr_debug_lane_stats; scripts/exp_octant_uniform.py).  usage: CUTRACE_AMD_LIB=build_variants/octs.so gpu_oct_stats.py"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cutrace_amd as ca
from cutrace_amd import scenes, _lib
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5, None), ("dense64k", scenes.make_dense_bunny(d, 3), 5, None), ("c3deep", scenes.make_mirror_deep(d), 8, None),
        ("c4", scenes.make_bunny_grid(d), 5, (4096, 4096))]
for name, path, b, size in todo:
    s = ca.HostScene.load(os.path.join(ROOT, path) if not os.path.isabs(path) else path)
    ds = ca.DeviceScene(s)
    ds.set_variant(ca.VAR_STATS)
    raw = np.zeros(96, np.uint64)
    _lib.hip_lib().ctr_debug_lane_stats(raw.ctypes.data, 1)
    ds.render(bounces=b)
    _lib.hip_lib().ctr_debug_lane_stats(raw.ctypes.data, 1)
    e_u, e_n, v_u, v_n = (int(raw[i]) for i in (84, 85, 86, 87))
    print(f"{name}: mesh entries uniform {e_u} / mixed {e_n} ({e_u / max(1, e_u + e_n):.3f}); node visits uniform {v_u} / mixed {v_n} ({v_u / max(1, v_u + v_n):.3f})", flush=True)
    ds.close()
