"""The three image writers on a 1080p frame (host only): ms with one and with all encoder threads (CUTRACE_JPEG_THREADS), sequential and in parallel."""
import sys, os, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import numpy as np, ctypes as C, threading, tempfile
    from cutrace_amd import _lib
    import cutrace_amd as ca, oracle
    L = _lib.host_lib()
    s = ca.HostScene.load("scene/bunny.json"); s.set_size(1920, 1080)
    r = oracle.oracle_render(s, bounces=0, threads=os.cpu_count() or 4, hit_ids=False)   # a real frame's content (primary hit + shading)
    d = tempfile.mkdtemp().encode()
    w, h = 1920, 1080
    jobs = [lambda: L.ctr_write_depth_map(d + b"/d.jpg", r["depth"].ctypes.data, w, h, C.c_float(6.0)),
            lambda: L.ctr_write_normal_map(d + b"/n.jpg", r["normal"].ctypes.data, w, h),
            lambda: L.ctr_write_colorized(d + b"/c.jpg", r["color"].ctypes.data, w, h)]
    for rep in range(2):
        t0 = time.perf_counter()
        for j in jobs: j()
        t1 = time.perf_counter()
        th = [threading.Thread(target=j) for j in jobs]
        for t in th: t.start()
        for t in th: t.join()
        t2 = time.perf_counter()
    print("threads per encoder %-4s: one after the other %.1f ms, the three at once %.1f ms" % (os.environ.get("CUTRACE_JPEG_THREADS", "all"), (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
else:
    for thr in ("1", "4", ""):
        env = dict(os.environ)
        if thr: env["CUTRACE_JPEG_THREADS"] = thr
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, cwd=ROOT)
