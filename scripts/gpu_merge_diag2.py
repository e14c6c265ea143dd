"""Diagnostic: time buckets (CUTRACE_AMD_LIB=build_variants/timing.so) of the merged and the two-level walk on C4 / mirror / C3-deep."""
import sys, os, tempfile, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
names = ["cast setup", "planes", "object loop", "TLAS + AABB", "mesh entry setup", "BVH walk w/o leaves", "leaves", "radiance cont", "rest of cont", "whole wave"]
for label, path, b in (("c4", scenes.make_bunny_grid(d), 5), ("c3deep", scenes.make_mirror_deep(d), 8), ("mirror", "scene/mirror.json", 8)):
    s = ca.HostScene.load(path)
    row = []
    for var in (ca.VAR_MERGE, 0):
        ds = ca.DeviceScene(s)
        ds.set_variant(var)
        for _ in range(3):
            ds.render(bounces=b)
        t = statistics.median(ds.render(bounces=b)["kernel_ms"] for _ in range(5))
        c = [int(x) for x in ds.last_counters()]
        row.append((t, c))
        ds.close()
    print(label, "bounces", b, "merged %.4f two-level %.4f" % (row[0][0], row[1][0]), flush=True)
    if os.environ.get("CUTRACE_AMD_LIB"):
        for q, nm in enumerate(names):
            print("    %-22s merged %14d   two-level %14d" % (nm, row[0][1][4 + q], row[1][1][4 + q]))
