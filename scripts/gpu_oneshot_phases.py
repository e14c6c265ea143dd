"""What a one-shot process pays, call by call (fresh process): HIP initialisation, ctr_scene_create, the page-locked frame, the first ctr_render."""
import sys, os, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t = [time.perf_counter()]
import numpy as np
import cutrace_amd as ca
from cutrace_amd import _lib
t.append(time.perf_counter())
L = _lib.hip_lib()
t.append(time.perf_counter())
n = L.ctr_device_count()
t.append(time.perf_counter())
s = ca.HostScene.load("scene/bunny.json")
t.append(time.perf_counter())
ds = ca.DeviceScene(s)
t.append(time.perf_counter())
px = 1920 * 1080
d, c, m = C.POINTER(C.c_float)(), C.POINTER(C.c_float)(), C.POINTER(C.c_float)()
L.ctr_frame_alloc(px, C.byref(d), C.byref(c), C.byref(m))
t.append(time.perf_counter())
st = _lib.RenderStats()
r = _lib.Rows(0, 1080, 1080, 0, 1)
L.ctr_render(ds._h, C.c_float(1e-3), 5, C.byref(r), d, c, m, C.byref(st))
t.append(time.perf_counter())
k1 = st.kernel_ms
L.ctr_render(ds._h, C.c_float(1e-3), 5, C.byref(r), d, c, m, C.byref(st))
t.append(time.perf_counter())
names = ["import numpy + package", "dlopen libcutrace_amd (+ libamdhip64)", "ctr_device_count (HIP initialisation)", "scene JSON + STL", "ctr_scene_create",
         "ctr_frame_alloc (58 MB page-locked)", "first ctr_render (kernel %.2f ms)" % k1, "second ctr_render (kernel %.2f ms)" % st.kernel_ms]
for nm, a, b in zip(names, t, t[1:]):
    print("%-45s %8.2f ms" % (nm, (b - a) * 1e3))
# the same frame into pageable memory, fresh process state aside
dep = np.empty((1080, 1920), np.float32); col = np.empty((1080, 1920, 3), np.float32); nor = np.empty((1080, 1920, 3), np.float32)
t0 = time.perf_counter()
L.ctr_render(ds._h, C.c_float(1e-3), 5, C.byref(r), dep.ctypes.data, col.ctypes.data, nor.ctypes.data, C.byref(st))
print("%-45s %8.2f ms" % ("ctr_render into fresh pageable numpy buffers", (time.perf_counter() - t0) * 1e3))
