"""One-off stress of the kernel's delivery into page-locked memory (render_kernel.hip "Host delivery"): N frames of changing
content (moving camera, changing bounce count, three image sizes) through ONE scene handle, each compared bitwise with
the same frame through device buffers + DMA (CTR_VAR_NO_DIRECT) on a second handle.  usage: gpu_hostdel_stress.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cutrace_amd as ca
from cutrace_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
bad = 0
for scene, sizes in (("scene/bunny.json", [(1920, 1080), (1000, 563), (1916, 1076)]), ("scene/mirror.json", [(1920, 1080)]),
                     ("scene/sphere_plane.json", [(1920, 1080), (333, 777)])):
    s = ca.HostScene.load(scene)
    cam0 = s.desc.contents.cam
    for (w, h) in sizes:
        s.set_size(w, h)
        direct, plain = ca.DeviceScene(s), ca.DeviceScene(s)
        plain.set_variant(ca.VAR_NO_DIRECT)
        for i in range(n):
            c = ca.Camera()
            C.memmove(C.byref(c), C.byref(s.desc.contents.cam), C.sizeof(ca.Camera))
            _lib.host_lib().ctr_camera_look_at(C.byref(c), _lib.Vec3(1.0 - 0.004 * i, 0.3 * np.sin(i / 7.0), 2.0 - 0.002 * i),
                                               _lib.Vec3(0, 1, 0), _lib.Vec3(-0.92388 + 0.001 * i, 0.0, -0.38268))
            direct.set_cameras([c])
            plain.set_cameras([c])
            b = (5, 1, 3, 0, 4, 2)[i % 6]
            got = direct.render(bounces=b, pinned=True)
            want = plain.render(bounces=b, pinned=True)
            ok = all(np.array_equal(got[k].view(np.uint32), want[k].view(np.uint32)) for k in ("depth", "color", "normal")) \
                and got["ray_count"] == want["ray_count"] and got["max_depth"] == want["max_depth"]
            if not ok:
                bad += 1
                print("MISMATCH", scene, w, h, "frame", i, flush=True)
        print(scene, f"{w}x{h}: {n} frames, {bad} bad so far", flush=True)
        direct.close(); plain.close()
print("done:", bad, "bad")
sys.exit(1 if bad else 0)
