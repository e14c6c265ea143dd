# usage: bash scripts/gpu_pmc_calib.sh — what do the SQ VALU counters read for instruction streams of KNOWN issue cost?
# (scripts/bin/valu_issue: fma_v = 2.2 cyc/inst, fma_s / pk_s / cmp_s64 = 4.1, rcp = 8.1 at 8 waves per SIMD)
set -x
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for m in fma_v fma_s pk_s cmp_s64 rcp mix; do
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE \
    --output-format csv -d $REPO/gpurun_out/pmc_calib_$m -o pmc -- $REPO/scripts/bin/valu_issue $m 8 > $REPO/gpurun_out/pmc_calib_$m.log 2>&1 || tail -5 $REPO/gpurun_out/pmc_calib_$m.log
done
cd $REPO
python3 - <<'PY'
import csv, glob, collections
for m in ("fma_v", "fma_s", "pk_s", "cmp_s64", "rcp", "mix"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/pmc_calib_{m}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "bench" in row["Kernel_Name"]:
                acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
    if not acc:
        print(m, "no data"); continue
    v = {k: x[-1] for k, x in acc.items()}   # last dispatch = the measured repetition
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    print(f"{m:8s} INSTS_VALU={v['SQ_INSTS_VALU']:.4g} ACTIVE_INST_VALU={v['SQ_ACTIVE_INST_VALU']:.4g} (x4/INSTS = {4*v['SQ_ACTIVE_INST_VALU']/v['SQ_INSTS_VALU']:.3f} cyc/inst) "
          f"kernel cycles={cyc:.4g}  VALUBusy=4*ACTIVE/(1024*cyc)={4*v['SQ_ACTIVE_INST_VALU']/(1024*cyc):.3f}  "
          f"THREAD_CYCLES/INSTS={v['SQ_THREAD_CYCLES_VALU']/v['SQ_INSTS_VALU']:.2f} WAVE_CYCLES={v['SQ_WAVE_CYCLES']:.4g} BUSY_CYCLES={v['SQ_BUSY_CYCLES']:.4g} "
          f"ACTIVE_ANY={v['SQ_ACTIVE_INST_ANY']:.4g} WAIT_INST_ANY={v['SQ_WAIT_INST_ANY']:.4g}")
PY
