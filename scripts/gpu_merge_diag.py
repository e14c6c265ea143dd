"""Diagnostic: where the merged walk spends its time on C3-deep (run with CUTRACE_AMD_LIB=build_variants/timing.so for buckets)."""
import sys, os, tempfile, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cutrace_amd as ca
from cutrace_amd import scenes
d = tempfile.mkdtemp()
names = ["cast setup", "planes", "object loop", "TLAS + AABB", "mesh entry setup", "BVH walk w/o leaves", "leaves", "radiance cont", "rest of cont", "whole wave"]
for label, path in (("c3deep", scenes.make_mirror_deep(d)), ("mirror", "scene/mirror.json")):
    s = ca.HostScene.load(path)
    for b in (0, 1, 2, 8):
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
                print("    %-22s merged %12d   two-level %12d" % (nm, row[0][1][4 + q], row[1][1][4 + q]))
