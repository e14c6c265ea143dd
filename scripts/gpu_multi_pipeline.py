"""ctr_multi_submit / ctr_multi_wait against ctr_render_multi, ms per frame, on the ONE GPU of the box: groups that list
device 0 several times (every part its own scene handle and streams, parts moved by peer copies).  Not a scaling number —
all parts share one device — but it shows what the pipeline hides: gather, re-interleave and copy-out of frame k under the
kernels of frame k+1.  usage: gpu_multi_pipeline.py"""
import os, statistics, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny@1080p", "scene/bunny.json", (1, 2, 4, 8)), ("C4 grid@4096^2", scenes.make_bunny_grid(d), (1, 4, 8))]
for name, path, groups in todo:
    s = ca.HostScene.load(path)
    for n in groups:
        m = ca.MultiScene(s, [0] * n)
        frames = [m.alloc_frame() for _ in range(3)]
        for _ in range(3):
            m.submit(frames[0]); m.wait()
        N = 12
        t0 = time.perf_counter()
        for k in range(N):
            m.submit(frames[k % 3]); m.wait()
        sync = (time.perf_counter() - t0) / N * 1e3
        t0 = time.perf_counter()
        m.submit(frames[0])
        for k in range(1, N):
            m.submit(frames[k % 3])
            st = m.wait()
        m.wait()
        pipe = (time.perf_counter() - t0) / N * 1e3
        print(f"{name:16s} group of {n}: one frame at a time {sync:7.3f} ms/frame, two in flight {pipe:7.3f} ms/frame "
              f"(kernel of the slowest part {st['kernel_ms']:.3f} ms, transport {m.transport})", flush=True)
        for f in frames:
            m.free_frame(f)
        m.close()
