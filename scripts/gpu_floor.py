import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cutrace_amd as ca
dev = torch.device("cuda:0")
def timeit(ds, w, h, bounces, counters):
    depth = torch.empty(w*h, dtype=torch.float32, device=dev)
    color = torch.empty(w*h*3, dtype=torch.float32, device=dev)
    normal = torch.empty(w*h*3, dtype=torch.float32, device=dev)
    cnt = torch.zeros(16, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream()
    def go():
        ds.render_device(depth.data_ptr(), color.data_ptr(), normal.data_ptr(), cnt.data_ptr() if counters else 0, st.cuda_stream, bounces=bounces)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/10
empty = {"camera": {"eye": [0, 0, -5], "up": [0, 1, 0], "look": [0, 0, 0], "near_plane": 0.1, "far_plane": 10, "width": 1920, "height": 1080, "ambient": 0.2}, "lights": [], "materials": [], "objects": []}
s = ca.HostScene.parse(json.dumps(empty)); ds = ca.DeviceScene(s)
print("empty with counters", timeit(ds, 1920, 1080, 5, True), "without", timeit(ds, 1920, 1080, 5, False))
for name, b in (("sphere_plane", 5), ("mirror", 8), ("bunny", 5), ("triangle", 5)):
    s = ca.HostScene.load(f"scene/{name}.json"); s.set_size(1920, 1080); ds = ca.DeviceScene(s)
    print(name, "with counters", round(timeit(ds, 1920, 1080, b, True), 3), "without", round(timeit(ds, 1920, 1080, b, False), 3))
