"""Segment execution counts of the render kernel (-DCTR_PROFILE build, CUTRACE_AMD_LIB=build_variants/profile.so) for the
bench workloads -> gpurun_out/profile_counts_<name>.json (input of scripts/dynamic_mix.py).
usage: CUTRACE_AMD_LIB=build_variants/profile.so python scripts/gpu_profile_mix.py [--c4]"""
import ctypes as C, json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes, _lib
L = _lib.hip_lib()
L.ctr_debug_profile_read.argtypes = [C.c_void_p, C.c_int]
L.ctr_debug_profile_read.restype = C.c_int
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5, "bunny.json@1920x1080b5"), ("dense64k", scenes.make_dense_bunny(d, 3), 5, "bunny_dense3.json@1920x1080b5")]
if "--c4" in sys.argv:
    todo.append(("c4", scenes.make_bunny_grid(d), 5, "bunny_grid4x4.json@4096x4096b5"))
for name, path, b, wl in todo:
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    ds.render(bounces=b)
    out = np.zeros(128, np.uint64)
    assert L.ctr_debug_profile_read(out.ctypes.data, 1) == 0
    N = 3
    rays = 0
    for _ in range(N):
        rays = ds.render(bounces=b)["ray_count"]
    assert L.ctr_debug_profile_read(out.ctypes.data, 1) == 0
    per = {str(i): float(out[i]) / N for i in range(128) if out[i]}
    json.dump({"workload": wl, "launches": N, "rays": rays, "per_launch": per}, open(os.path.join(ROOT, "gpurun_out", f"profile_counts_{name}.json"), "w"), indent=1)
    print(name, "waves", per.get("0"), "trips", per.get("1"), "nodes", per.get("32"), "prefilters", per.get("23"), "exact", per.get("28"), "unwind", per.get("53"), flush=True)
    ds.close()
