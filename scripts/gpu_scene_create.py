"""ctr_scene_create by phase (CUTRACE_DEBUG_CREATE=1) for C2, C2-dense, C4, with 1 and all builder threads (CUTRACE_BUILD_THREADS)."""
import sys, os, time, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import cutrace_amd as ca
    from cutrace_amd import scenes
    d = tempfile.mkdtemp()
    for name, path in (("C2 bunny.json", "scene/bunny.json"), ("C2-dense 64000", scenes.make_dense_bunny(d, 3)), ("C4 grid", scenes.make_bunny_grid(d))):
        hs = ca.HostScene.load(path)
        ca.DeviceScene(hs).close()   # warm: HIP context, allocator
        print("==", name, "threads", os.environ.get("CUTRACE_BUILD_THREADS", "all"), flush=True)
        sys.stderr.flush()
        t0 = time.perf_counter()
        ds = ca.DeviceScene(hs)
        print("   total %.3f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
        ds.close()
else:
    for thr in ("1", ""):
        env = dict(os.environ, CUTRACE_DEBUG_CREATE="1")
        if thr:
            env["CUTRACE_BUILD_THREADS"] = thr
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env, cwd=ROOT, stderr=subprocess.STDOUT)
