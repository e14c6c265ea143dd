"""A/B timing of several builds of libcutrace_amd.so on the SAME box in one gpurun call (devices differ by
several % — never compare numbers from different calls).  usage: gpu_ab.py name=path.so ... [--rounds N] [--stats] [--host]"""
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
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    first = ds.render(bounces=b)["kernel_ms"]
    for _ in range(3): ds.render(bounces=b)
    t = [ds.render(bounces=b)["kernel_ms"] for _ in range(9)]
    out[name] = round(statistics.median(t), 4)
    out[name + "_first"] = round(first, 4)
    if "--host" in sys.argv:   # ctr_render into a page-locked block, whole call
        for _ in range(3): ds.render(bounces=b, pinned=True)
        out[name + "_host"] = round(statistics.median(ds.render(bounces=b, pinned=True)["total_ms"] for _ in range(9)), 4)
    if "--stats" in sys.argv:
        ds.set_variant(ca.VAR_STATS)
        ds.render(bounces=b)
print(json.dumps(out))
''' % ROOT
libs = [a.split("=", 1) for a in sys.argv[1:] if "=" in a]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
extra = [f for f in ("--stats", "--host") if f in sys.argv]
res = {n: [] for n, _ in libs}
for r in range(rounds):
    for n, p in libs:
        env = dict(os.environ, CUTRACE_AMD_LIB=os.path.join(ROOT, p))
        q = subprocess.run([sys.executable, "-c", CHILD] + extra, capture_output=True, text=True, env=env, cwd=ROOT, timeout=600)
        if q.returncode:
            print(n, "FAILED", q.stderr[-400:], flush=True)
            continue
        res[n].append(json.loads(q.stdout.strip().splitlines()[-1]))
        for line in q.stderr.splitlines():
            if "stats:" in line or "timing" in line:
                print("  ", n, line, flush=True)
        print(n, res[n][-1], flush=True)
print("---- best of rounds ----")
for n, rs in res.items():
    if rs:
        print(n, {k: min(x[k] for x in rs) for k in rs[0]})
