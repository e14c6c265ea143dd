"""Parity under extreme magnitudes (one-off): seeded random scenes with every length scaled by 1e-15 ... 1e19 — overflow to
inf, underflow to denormals and zero, NaN from inf - inf all travel through both the kernel and the oracle — the default
kernel against the oracle (parity bar) and against the kernel without shortcuts (bitwise).
usage: python scripts/gpu_fuzz_scale.py [first_seed] [count]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cutrace_amd as ca
import oracle
from cutrace_amd import scenes
from tests.test_gpu_parity import _random_scene
from tests.util import assert_parity, same_bits

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
d = tempfile.mkdtemp()
skull = scenes.read_stl(os.path.join(ROOT, "scene", "skull.stl"))
bad = 0
for seed in range(first, first + count):
    for scale in (1e-15, 1e-6, 1e6, 1e15, 1e19):
        sc = json.loads(_random_scene(seed, w=64, h=40, opaque_mesh=seed % 2 == 0, extra_planes=seed % 3 != 0))
        f = float(scale)
        for k in ("eye", "look"):
            sc["camera"][k] = [v * f for v in sc["camera"][k]]
        for L in sc["lights"]:
            if L["type"] == "point":
                L["point"] = [v * f for v in L["point"]]
        for o in sc["objects"]:
            if o["type"] == "sphere":
                o["center"] = [v * f for v in o["center"]]; o["radius"] *= f
            elif o["type"] == "triangle":
                for k in ("p1", "p2", "p3"):
                    o[k] = [v * f for v in o[k]]
            elif o["type"] == "plane":
                o["point"] = [v * f for v in o["point"]]
            elif o["type"] == "mesh":
                p = os.path.join(d, f"skull_{scale:g}.stl")
                if not os.path.exists(p):
                    scenes.write_stl(p, (skull.astype(np.float64) * f).astype(np.float32))
                o["file"] = p
        s = ca.HostScene.parse(json.dumps(sc))
        if not s.ok:
            print("seed", seed, "scale", scale, "rejected:", s.error); continue
        b = [0, 2, 3, 5][seed % 4]
        for fudge in (1e-3, float(1e-3 * f)):
            o = oracle.oracle_render(s, bounces=b, fudge=fudge, threads=os.cpu_count() or 4)
            ds = ca.DeviceScene(s)
            r = ds.render(bounces=b, fudge=fudge)
            ds.set_variant(ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT)
            plain = ds.render(bounces=b, fudge=fudge)
            ok = all(same_bits(r[k], plain[k]) for k in ("depth", "normal", "color")) and r["ray_count"] == plain["ray_count"] == o["ray_count"]
            try:
                assert_parity(r, o, what=f"seed {seed} scale {scale:g} fudge {fudge:g}")
            except AssertionError as e:
                ok = False
                print(str(e)[:300])
            if not ok:
                bad += 1
                print("BAD seed", seed, "scale", scale, "fudge", fudge, flush=True)
            ds.close()
    if seed % 5 == 4:
        print(seed - first + 1, "seeds,", bad, "bad", flush=True)
print("done:", count, "seeds x 5 scales x 2 fudges,", bad, "bad")
