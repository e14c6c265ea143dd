"""counters.json for bench.py's roofline from the PMC passes of scripts/gpu_pmc.sh.

Per launch of the render kernel (mean over the dispatches of the steady state; the first dispatch of a run — the
scene's first launch, in image order — is left out):
  valu/salu/smem instructions  SQ_INSTS_*           (wave-level instruction counts)
  cycles                       GRBM_GUI_ACTIVE / 8  (rocprofv3 sums the 8 XCDs; MI355X_MICROARCH.md "DVFS")
  effective clock              cycles / (End - Start timestamp of the same dispatches)
  HBM bytes                    2 x FETCH_SIZE + WRITE_SIZE KiB (gfx950: FETCH_SIZE counts 64 B per 128-B request;
                               MI355X_MICROARCH.md §HBM), each from its own pass
usage: pmc_counters.py <dir-prefix> <kernel-substring> <workload> <out.json>"""
import csv, glob, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
prefix, kern, workload, out = sys.argv[1:5]


def load(name):
    global kernel_name
    rows = []
    for f in glob.glob(os.path.join(f"{prefix}_{name}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if kern in r["Kernel_Name"]:
                rows.append(r)
                kernel_name = r["Kernel_Name"]
    per = {}
    for r in rows:
        d = per.setdefault(int(r["Dispatch_Id"]), {"t": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)[1:] or sorted(per)   # drop the first (image-order) launch
    return [per[i] for i in ids]


kernel_name = ""


def mean(rows, key):
    v = [r[key] for r in rows if key in r]
    return sum(v) / len(v) if v else None


sq1, grbm, fetch, write = load("sq1"), load("grbm"), load("fetch"), load("write")
sq2, sqc = load("sq2"), load("sqc")
cyc = mean(grbm, "GRBM_GUI_ACTIVE") / 8.0
t_ns = mean(grbm, "t")
fk, wk = mean(fetch, "FETCH_SIZE"), mean(write, "WRITE_SIZE")
res = {
    "workload": workload, "kernel": (re.search(r"render_kernel<\d+u>", kernel_name) or [kern])[0] if kernel_name else kern,
    "dispatches_averaged": len(sq1),
    "valu_insts_per_launch": mean(sq1, "SQ_INSTS_VALU"), "salu_insts_per_launch": mean(sq1, "SQ_INSTS_SALU"),
    "smem_insts_per_launch": mean(sq1, "SQ_INSTS_SMEM"), "waves_per_launch": mean(sq1, "SQ_WAVES"),
    "wave_quadcycles_per_launch": mean(sq1, "SQ_WAVE_CYCLES"), "wait_any_quadcycles": mean(sq1, "SQ_WAIT_ANY"),
    "wait_inst_any_quadcycles": mean(sq1, "SQ_WAIT_INST_ANY"),
    "active_inst_valu_quadcycles": mean(sq2, "SQ_ACTIVE_INST_VALU"), "active_inst_any_quadcycles": mean(sq2, "SQ_ACTIVE_INST_ANY"),
    "thread_cycles_valu": mean(sq2, "SQ_THREAD_CYCLES_VALU"), "valu_trans_f32_insts": mean(sq2, "SQ_INSTS_VALU_TRANS_F32"),
    "sqc_dcache_req": mean(sqc, "SQC_DCACHE_REQ"), "sqc_dcache_hits": mean(sqc, "SQC_DCACHE_HITS"),
    "sqc_dcache_misses": mean(sqc, "SQC_DCACHE_MISSES"), "sqc_dcache_misses_duplicate": mean(sqc, "SQC_DCACHE_MISSES_DUPLICATE"),
    "cycles_per_launch": cyc, "kernel_ns_in_pmc_pass": t_ns, "effective_clock_ghz": cyc / t_ns,
    "FETCH_SIZE_KiB_raw": fk, "WRITE_SIZE_KiB_raw": wk, "fetch_correction": 2.0,
    "hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024),
    "kernel_source_sha256": __import__("bench").kernel_source_hash(),
    "source": "scripts/gpu_pmc.sh -> scripts/pmc_counters.py",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
