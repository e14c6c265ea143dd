"""Differential fuzz of what round 4 added (one-off, larger than the pytest suite), N seeded random scenes each:
  * scenes with 2-6 meshes (tests/test_gpu_merged.py generator): the two-level walk (default) vs the merged walk (CTR_VAR_MERGE)
    vs the kernel with every shortcut off, bitwise, and vs the oracle; STATS counts how many casts the merged walk handed back;
  * CTR_VAR_IGNORE_TRANSPARENT (the kernel.hpp:52 cast with ray_cast's ignore_transparent = true) vs the oracle's restatement.
usage: python scripts/gpu_fuzz_r04.py [first_seed] [count]"""
import os, sys, time, tempfile, pathlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import cutrace_amd as ca
import oracle
from tests.test_gpu_parity import _random_scene
from tests.test_gpu_merged import _multi_mesh_scene
from tests.util import assert_parity, same_bits

first = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tmp = pathlib.Path(tempfile.mkdtemp())
bad = redone = walks = 0
t0 = time.time()
thr = os.cpu_count() or 4
for seed in range(first, first + count):
    ok = True
    w, h = [(88, 56), (61, 37), (130, 24), (64, 64)][(seed // 6) % 4]
    b = [0, 1, 2, 3, 5][seed % 5]
    try:
        # ---- several meshes: two-level vs merged vs plain vs oracle ----
        s = ca.HostScene.parse(_multi_mesh_scene(tmp, seed, w=w, h=h, opaque=(seed % 2 == 0), n_mesh=2 + seed % 5))
        assert s.ok
        o = oracle.oracle_render(s, bounces=b, threads=thr)
        ds = ca.DeviceScene(s)
        r = ds.render(bounces=b)
        assert_parity(r, o, what=f"multi-mesh seed {seed}")
        assert r["ray_count"] == o["ray_count"]
        for var in (ca.VAR_MERGE, ca.VAR_NO_CLUSTER | ca.VAR_NO_PREFILTER | ca.VAR_NO_ANYHIT):
            ds.set_variant(var)
            x = ds.render(bounces=b)
            assert all(same_bits(r[k], x[k]) for k in ("depth", "normal", "color")) and x["ray_count"] == r["ray_count"], f"variant {var} differs"
        ds.set_variant(ca.VAR_STATS | ca.VAR_MERGE)
        ca.DeviceScene.lane_stats(reset=True)
        sys.stderr.flush()
        ds.render(bounces=b)
        st = ca.DeviceScene.lane_stats(reset=True)
        walks += st["merged_walks"]
        redone += st["merged_walks_redone"]
        ds.close()
        # ---- ignore_transparent ----
        s2 = ca.HostScene.parse(_random_scene(seed, w=w, h=h, opaque_mesh=False, extra_planes=seed % 3 != 0))
        assert s2.ok
        o2 = oracle.oracle_render(s2, bounces=b, threads=thr, uv=True, ignore_transparent_primary=True)
        d2 = ca.DeviceScene(s2)
        d2.set_variant(ca.VAR_IGNORE_TRANSPARENT)
        r2 = d2.render_uv(bounces=b)
        assert_parity(r2, o2, what=f"ignore_transparent seed {seed}")
        assert r2["ray_count"] == o2["ray_count"]
        d2.close()
    except AssertionError as e:
        ok = False
        print("MISMATCH seed", seed, str(e)[:300], flush=True)
    if not ok:
        bad += 1
    if (seed - first) % 50 == 49:
        print(f"{seed - first + 1} scene pairs, {bad} bad, merged walks {walks} ({redone} handed back), {time.time() - t0:.0f} s", flush=True)
print("done:", count, "scene pairs,", bad, "bad; merged walks", walks, "handed back to the two-level walk", redone)
sys.exit(1 if bad else 0)
