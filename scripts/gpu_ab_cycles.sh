# usage: bash scripts/gpu_ab_cycles.sh name=lib.so ...   (through gpurun) — A/B in kernel CYCLES (GRBM_GUI_ACTIVE per dispatch, a PMC pass of its own per
# library, never mixed with tracing), two rounds; the per-scene figure is the median over the steady dispatches (the first 4 of each scene dropped).
REPO=$(pwd)
mkdir -p gpurun_out/abc
cd /tmp && export TMPDIR=/tmp
for round in 1 2; do
  for spec in "$@"; do
    name=${spec%%=*}; lib=${spec#*=}
    export CUTRACE_AMD_LIB=$REPO/$lib
    rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $REPO/gpurun_out/abc/${name}_$round -o pmc -- python3 $REPO/scripts/render_loop.py > $REPO/gpurun_out/abc/${name}_$round.log 2>&1 || tail -3 $REPO/gpurun_out/abc/${name}_$round.log
  done
done
cd $REPO
python3 - "$@" <<'PY'
import csv, glob, statistics, sys, collections
names = [a.split("=")[0] for a in sys.argv[1:]]
# render_loop.py's order: bunny 34 dispatches, 64 000 triangles 20, C3-deep 20 (all 1920x1080), C4 12 (4096x4096); the first 4 of each are warm-up
PLAN = (("bunny", 34), ("dense64k", 20), ("c3deep", 20), ("c4", 12))
res = collections.defaultdict(dict)
for rnd in (1, 2):
    for n in names:
        rows = []
        for f in glob.glob(f"gpurun_out/abc/{n}_{rnd}/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                if "render_kernel" in row["Kernel_Name"] and row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    rows.append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
        v = [c for _, c in sorted(rows)]
        out, at = [], 0
        for scene, cnt in PLAN:
            part = v[at + 4:at + cnt]; at += cnt
            if part:
                m = statistics.median(part) / 1e6
                res[scene].setdefault(n, []).append(m)
                out.append(f"{scene} {m:.4f}")
        print(f"round {rnd} {n}: " + "  ".join(out) + "   (Mcycles, GRBM_GUI_ACTIVE summed over the XCDs, median of the steady dispatches)", flush=True)
print("---- relative to the first library (mean of the rounds) ----")
for scene, _ in PLAN:
    base = statistics.mean(res[scene][names[0]])
    print(scene, "  ".join(f"{n} {100 * (statistics.mean(res[scene][n]) / base - 1):+.2f}%" for n in names if n in res[scene]))
PY
