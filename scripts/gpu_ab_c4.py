"""A/B of library builds on the multi-mesh scenes (C4: 4x4 bunny grid @4096x4096, 16 meshes; mirror.json: 3 meshes), same box,
kernel ms by ctr_render.  usage: gpu_ab_c4.py name=lib.so ..."""
import json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, statistics, tempfile
sys.path.insert(0, %r)
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
out = {}
for name, path, b in (("c4", scenes.make_bunny_grid(d), 5), ("mirror", "scene/mirror.json", 8)):
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    for _ in range(3): ds.render(bounces=b)
    out[name] = round(statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(5)), 4)
print(json.dumps(out))
''' % ROOT
libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
for r in range(2):
    for n, p in libs:
        q = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=dict(os.environ, CUTRACE_AMD_LIB=os.path.join(ROOT, p)), cwd=ROOT, timeout=600)
        print(n, q.stdout.strip().splitlines()[-1] if q.returncode == 0 else "FAILED " + q.stderr[-300:], flush=True)
