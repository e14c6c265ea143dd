"""Build kernel variants with different -D knobs ON THE GPU BOX and time them.
usage: python scripts/tune.py "NAME:-DX=1 -DY=2" ...   (run through gpurun)"""
import json, os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cutrace_amd import build

CHILD = r'''
import sys, os, json, statistics
sys.path.insert(0, %r)
import cutrace_amd as ca
out = {}
for name, b in (("bunny", 5), ("mirror", 8), ("sphere_plane", 5)):
    s = ca.HostScene.load("scene/%%s.json" %% name)
    ds = ca.DeviceScene(s)
    ds.render(bounces=b)
    t = [ds.render(bounces=b)["kernel_ms"] for _ in range(7)]
    out[name] = round(statistics.median(t), 3)
import tempfile
from cutrace_amd import scenes
s = ca.HostScene.load(scenes.make_dense_bunny(tempfile.mkdtemp(), 2))
ds = ca.DeviceScene(s)
ds.render()
out["dense16k"] = round(statistics.median([ds.render()["kernel_ms"] for _ in range(7)]), 3)
print(json.dumps(out))
''' % ROOT

def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    build.build_host()
    results = {}
    for spec in sys.argv[1:]:
        name, _, flags = spec.partition(":")
        lib = os.path.join(ROOT, "gpurun_out", f"libtune_{name}.so")
        cmd = [build.hipcc(), *build.HIP_FLAGS, *flags.split(), "-shared", "-o", lib, *build.HIP_SRCS]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode:
            print(name, "BUILD FAILED", r.stderr[-500:], flush=True)
            continue
        env = dict(os.environ, CUTRACE_AMD_LIB=lib)
        r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, env=env, cwd=ROOT, timeout=300)
        line = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        for el in r.stderr.splitlines():
            if "timing" in el:
                print("   ", el, flush=True)
        print(f"{name:28s} {flags:60s} {line}", flush=True)
        os.remove(lib)
if __name__ == "__main__":
    main()
