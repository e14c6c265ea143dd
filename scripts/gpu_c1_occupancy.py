"""Is C1 (sphere_plane.json) held back by its LDS stack?  Its one material that both reflects (0.05) and transmits (0.6) makes every stack frame 10 dwords
(12.8 KB per wave at bounces 5: 12 waves per CU).  The same scene with that material's reflection set to 0 needs 4-dword frames (5 KB: 20+ waves): a different
image, nearly the same rays (the reflection child of a 0.05-reflective glass sphere is one of ~3 rays per hit) — if the frame time barely moves, occupancy is not
what C1 waits for.  Also: the real scene with extra LDS padding (fewer waves), the other direction."""
import sys, os, json, statistics, tempfile, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT)
    import cutrace_amd as ca
    d = json.load(open(os.path.join(ROOT, "scene", "sphere_plane.json")))
    out = {}
    for tag, refl in (("as shipped (reflect 0.05 + transparency 0.6: 10-dword frames)", None), ("reflect 0 on that material (4-dword frames)", 0.0)):
        dd = json.loads(json.dumps(d))
        if refl is not None:
            dd["materials"][1]["reflect"] = refl
        s = ca.HostScene.parse(json.dumps(dd))
        ds = ca.DeviceScene(s)
        r = ds.render(bounces=5)
        for _ in range(3): ds.render(bounces=5)
        out[tag] = (round(statistics.median(ds.render(bounces=5)["kernel_ms"] for _ in range(9)), 4), r["ray_count"])
    print(os.environ.get("CUTRACE_LDS_PAD", "0"), out, flush=True)
else:
    for pad in ("0", "3200", "8000"):
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, CUTRACE_LDS_PAD=pad), cwd=ROOT)
