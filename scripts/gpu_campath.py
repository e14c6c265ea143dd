"""Tile scheduling on a MOVING camera: 90 frames of the bunny room, the eye moving ~1.5 cm and the
view direction turning ~0.35 deg per frame; every frame is one launch whose tile order comes from the
previous (different) frame.  Prints mean kernel ms per frame with the scheduler on and off."""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutrace_amd as ca
from cutrace_amd import _lib

s = ca.HostScene.load("scene/bunny.json")
w, h = s.size
cam0 = s.desc.contents.cam
N = 90
cams = []
for k in range(N):
    c = ca.Camera()
    C.memmove(C.byref(c), C.byref(cam0), C.sizeof(ca.Camera))
    a = math.radians(-22.5 + 0.35 * k)     # look direction swings through the shipped one (-22.5 deg + ...)
    eye = _lib.Vec3(1.0 - 0.015 * k * 0.5, 0.1 * math.sin(k / 15.0), 2.0)
    look = _lib.Vec3(-math.cos(a) * 0.92388 / math.cos(math.radians(22.5)) if False else -math.sin(math.radians(67.5) - a + math.radians(-22.5)), 0.0, -math.cos(math.radians(67.5) - a + math.radians(-22.5)))
    _lib.host_lib().ctr_camera_look_at(C.byref(c), eye, _lib.Vec3(0, 1, 0), look)
    cams.append(c)
dev = torch.device("cuda:0")
depth = torch.zeros(h * w, dtype=torch.float32, device=dev)
color = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
normal = torch.zeros(h * w * 3, dtype=torch.float32, device=dev)
counters = torch.zeros(16, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream()
for var, name in ((ca.VAR_NO_REORDER, "image order"), (0, "cost order from the previous frame")):
    ds = ca.DeviceScene(s)
    ds.set_cameras(cams)
    ds.set_variant(var)
    for rep in range(2):
        evs = []
        counters.zero_()
        for k in range(N):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            ds.render_device_batch(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), n_frames=1, frame_stride_px=h * w,
                                   first_frame=k, d_counters=counters.data_ptr(), stream=stream.cuda_stream)
            e1.record(stream)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = [a.elapsed_time(b) for a, b in evs]
    print(f"{name:36s} mean {sum(ms) / N:.3f} ms/frame  (min {min(ms):.3f} max {max(ms):.3f}), rays/frame {int(counters[0]) // N}", flush=True)
