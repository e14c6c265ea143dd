"""XCD bands (CUTRACE_XCD_BANDS, ctr_api.cpp build_xcd_order): each XCD's tiles kept in one band of the image, against the
plain cost-sorted order; kernel ms per config, outputs compared bitwise.  usage: gpu_xcd.py [--c4]"""
import hashlib, json, os, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, os, json, statistics, tempfile, hashlib
sys.path.insert(0, %r)
import numpy as np
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
todo = [("bunny", "scene/bunny.json", 5), ("dense64k", scenes.make_dense_bunny(d, 3), 5), ("mirror", "scene/mirror.json", 8)]
if "--c4" in sys.argv: todo.append(("c4", scenes.make_bunny_grid(d), 5))
out = {}
for name, path, b in todo:
    s = ca.HostScene.load(path)
    ds = ca.DeviceScene(s)
    for _ in range(5): r = ds.render(bounces=b)
    t = [ds.render(bounces=b)["kernel_ms"] for _ in range(9)]
    r = ds.render(bounces=b)
    h = hashlib.sha1(np.ascontiguousarray(r["depth"]).tobytes() + np.ascontiguousarray(r["color"]).tobytes() + np.ascontiguousarray(r["normal"]).tobytes()).hexdigest()[:12]
    out[name] = round(statistics.median(t), 4)
    out[name + "_sha"] = h
print(json.dumps(out))
''' % ROOT
extra = [a for a in sys.argv[1:] if a.startswith("--")]
for label, env in (("sorted", {}), ("bands8_rows", {"CUTRACE_XCD_BANDS": "8", "CUTRACE_XCD_MODE": "0"}),
                   ("bands8_cols", {"CUTRACE_XCD_BANDS": "8", "CUTRACE_XCD_MODE": "1"}),
                   ("bands8_blocks", {"CUTRACE_XCD_BANDS": "8", "CUTRACE_XCD_MODE": "2"}),
                   ("bands16_rows", {"CUTRACE_XCD_BANDS": "16", "CUTRACE_XCD_MODE": "0"}),
                   ("sorted_again", {})):
    q = subprocess.run([sys.executable, "-c", CHILD] + extra, capture_output=True, text=True, env=dict(os.environ, **env), cwd=ROOT, timeout=900)
    print(label, q.stdout.strip().splitlines()[-1] if q.returncode == 0 else "FAILED " + q.stderr[-600:], flush=True)
