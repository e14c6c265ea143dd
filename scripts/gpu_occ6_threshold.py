"""Which build (5 or 6 waves per SIMD, KV_OCC6) suits which scene: every config rendered with CUTRACE_OCC6_MIN_TRIS = 0 (always the 6-wave build),
the default (1000 mesh triangles) and a huge value (never).  Same box, fresh process per setting, median of 9 after 3 warm-up frames; 3 rounds."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, statistics, tempfile
sys.path.insert(0, %r)
import cutrace_amd as ca
from cutrace_amd import scenes
out = {}
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("mirror", "scene/mirror.json", 8), ("sphere_plane", "scene/sphere_plane.json", 5),
        ("dense64k", scenes.make_dense_bunny(d, 3), 5), ("c3deep", scenes.make_mirror_deep(d), 8)]
for name, path, b in todo:
    ds = ca.DeviceScene(ca.HostScene.load(path))
    for _ in range(3): ds.render(bounces=b)
    out[name] = round(statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(9)), 4)
print(json.dumps(out))
''' % ROOT
best = {}
for r in range(3):
    for tag, val in (("always6", "0"), ("default", None), ("never6", "1000000000")):
        env = dict(os.environ)
        if val is not None: env["CUTRACE_OCC6_MIN_TRIS"] = val
        q = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        if q.returncode: print(tag, "FAILED", q.stderr[-300:]); continue
        o = json.loads(q.stdout.strip().splitlines()[-1])
        print(tag, o, flush=True)
        b = best.setdefault(tag, dict(o))
        for k in o: b[k] = min(b[k], o[k])
print("---- best of rounds ----")
for t, b in best.items(): print(t, b)
