"""Differential fuzz (one-off, larger than the pytest suite): N seeded random scenes, the default kernel
vs. the kernel with every shortcut off (bitwise), vs. delivery into page-locked memory by the kernel (bitwise) and
vs. the CPU oracle (parity bar).
usage: python scripts/gpu_fuzz.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cutrace_amd as ca
import oracle
from tests.test_gpu_parity import _random_scene
from tests.util import assert_parity, same_bits

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    opaque = seed % 2 == 0
    w, h = [(88, 56), (61, 37), (130, 24), (64, 64)][(seed // 6) % 4]   # ragged right / bottom tiles and tile groups
    s = ca.HostScene.parse(_random_scene(seed, w=w, h=h, opaque_mesh=opaque, extra_planes=seed % 3 != 0))  # (walls of all kinds: the axis-aligned plane path)
    assert s.ok
    b = [0, 1, 2, 3, 5, 7][seed % 6]
    ds = ca.DeviceScene(s)
    r = ds.render(bounces=b)
    ds.set_variant(ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT)
    plain = ds.render(bounces=b)
    ok = all(same_bits(r[k], plain[k]) for k in ("depth", "normal", "color")) and r["ray_count"] == plain["ray_count"]
    ds.set_variant(0)
    for rep in range(2):  # page-locked destination: delivered by the kernel itself (second call: measured tile order)
        hd = ds.render(bounces=b, pinned=True)
        if not (all(same_bits(r[k], hd[k]) for k in ("depth", "normal", "color")) and r["ray_count"] == hd["ray_count"]
                and r["max_depth"] == hd["max_depth"]):
            ok = False
            print("HOST DELIVERY MISMATCH seed", seed, "call", rep, flush=True)
        hd["depth"][:] = -1.0
    try:
        o = oracle.oracle_render(s, bounces=b, threads=os.cpu_count() or 4)
        assert_parity(r, o, what=f"seed {seed}")
        assert r["ray_count"] == o["ray_count"]
    except AssertionError as e:
        ok = False
        print("ORACLE MISMATCH", str(e)[:300], flush=True)
    if not ok:
        bad += 1
        print("FAIL seed", seed, "opaque", opaque, "bounces", b, flush=True)
    ds.close()
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} scenes, {bad} bad, {time.time() - t0:.0f} s", flush=True)
print("done:", count, "scenes,", bad, "bad")
sys.exit(1 if bad else 0)
