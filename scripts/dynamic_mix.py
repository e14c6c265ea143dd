"""Execution-weighted VALU instruction mix of the render kernel (the dynamic counterpart of valu_mix.py).

The kernel source is cut into straight-line segments by CTR_MARK(n) (render_kernel.hip).  Two builds of the same source:
  -DCTR_MARKS    every mark is a comment in the ISA: the instructions that follow a mark in layout order, up to the next
                 mark, are that segment's (a segment inlined k times — tri_test x4, plane_test x8 — is averaged over its copies)
  -DCTR_PROFILE  every mark counts how often a wave runs its segment (scripts/gpu_profile_mix.py -> counts JSON)
so   dynamic instructions of a class = sum over segments  executions x instructions of that class in the segment,
priced with the issue costs scripts/valu_issue.hip measured (profiles/r02/valu_issue.txt): F 2.2, H 4.1, Q 8.1 cycles.
Cross-checks against PMC passes of the shipped build (same scene): the VALU total against SQ_INSTS_VALU, the Q count
against SQ_ACTIVE_INST_VALU - SQ_INSTS_VALU (a transcendental holds the pipe two quad-cycles), SALU against SQ_INSTS_SALU.

usage: dynamic_mix.py <marks.s> <variant> <counts.json> [--pmc counters.json] [--out mix.json]"""
import json, re, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from valu_mix import classify, F, H, Q


def kernel_body(path, kv):
    name = "_ZN12_GLOBAL__N_113render_kernelILj%sEEE" % kv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(name) and ":" in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith("; Occupancy"))
    return lines[start:end]


def segments(body):
    """-> {id: {"copies": n, "F":.., "H":.., "Q":.., "salu":.., "smem":.., "vmem":.., "lds":.., "nop":..}} (totals over copies)

    Attribution works on the compiler's basic blocks (a label or a '; %bb.N:' line opens one): a mark is a volatile
    comment, and the scheduler may sink it below arithmetic of its own block, so a block that holds marks belongs to
    them entirely — what precedes the first mark of the block goes to that mark; a block without a mark continues the
    segment of the block before it in layout order."""
    blocks, cur = [], []
    for raw in body:
        line = raw.strip()
        if re.match(r"^(\.LBB\d+_\d+:|; %bb\.\d+:)", line):
            if cur:
                blocks.append(cur)
            cur = []
            continue
        if line:
            cur.append(line)
    if cur:
        blocks.append(cur)
    seg = {}
    last = None

    def add(mark, line):
        d = seg[mark]
        if line.startswith("v_"):
            d[classify(line)] += 1
        elif line.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            d["smem"] += 1
        elif line.startswith("s_nop"):
            d["nop"] += 1
        elif line.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
            d["branch"] += 1
        elif line.startswith("s_") and not line.startswith("s_waitcnt"):
            d["salu"] += 1
        elif line.startswith(("global_", "flat_", "buffer_", "scratch_")):
            d["vmem"] += 1
        elif line.startswith("ds_"):
            d["lds"] += 1

    for blk in blocks:
        marks = [(i, int(m.group(1), 0)) for i, l in enumerate(blk)
                 for m in [re.match(r";\s*CTR_MARK (0x[0-9a-fA-F]+|\d+)", l)] if m]
        for _, k in marks:
            seg.setdefault(k, dict(copies=0, F=0, H=0, Q=0, salu=0, branch=0, smem=0, vmem=0, lds=0, nop=0))["copies"] += 1
        cur_mark = marks[0][1] if marks else last
        nxt = 1
        for i, l in enumerate(blk):
            if nxt < len(marks) and i >= marks[nxt][0]:
                cur_mark = marks[nxt][1]
                nxt += 1
            if l.startswith((";", ".")) or cur_mark is None:
                continue
            add(cur_mark, l)
        if marks:
            last = marks[-1][1]
    return seg


def main():
    path, kv, counts_path = sys.argv[1:4]
    pmc = json.load(open(sys.argv[sys.argv.index("--pmc") + 1])) if "--pmc" in sys.argv else None
    out = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else None
    cj = json.load(open(counts_path))
    counts = {int(k): float(v) for k, v in cj["per_launch"].items()}
    seg = segments(kernel_body(path, kv))
    tot = dict(F=0.0, H=0.0, Q=0.0, salu=0.0, branch=0.0, smem=0.0, vmem=0.0, lds=0.0, nop=0.0)
    rows = []
    for i in sorted(seg):
        d = seg[i]
        n = counts.get(i, 0.0)
        per = {k: d[k] / d["copies"] for k in tot}
        for k in tot:
            tot[k] += n * per[k]
        rows.append(dict(mark=i, copies=d["copies"], executions=n, **{k: round(per[k], 2) for k in tot},
                         valu_share=None))
    valu = tot["F"] + tot["H"] + tot["Q"]
    cyc = F * tot["F"] + H * tot["H"] + Q * tot["Q"]
    for r in rows:
        r["valu_share"] = round(r["executions"] * (r["F"] + r["H"] + r["Q"]) / valu, 4) if valu else 0
    res = {
        "workload": cj.get("workload"), "kernel": "render_kernel<%su>" % kv,
        "method": "segment executions (CTR_PROFILE build) x static class counts of the segment (CTR_MARKS build)",
        "issue_cycles": {"F": F, "H": H, "Q": Q},
        "dynamic_per_launch": {k: round(v) for k, v in tot.items()}, "valu_insts_modelled": round(valu),
        "mean_issue_cycles_per_valu": round(cyc / valu, 4),
        "class_shares": {k: round(tot[k] / valu, 4) for k in ("F", "H", "Q")},
        "segments": rows,
    }
    if pmc:
        res["pmc_check"] = {
            "SQ_INSTS_VALU": pmc["valu_insts_per_launch"], "modelled_over_measured_valu": round(valu / pmc["valu_insts_per_launch"], 4),
            "SQ_INSTS_SALU": pmc["salu_insts_per_launch"], "modelled_over_measured_salu": round((tot["salu"] + tot["nop"]) / pmc["salu_insts_per_launch"], 4),
            "modelled_over_measured_salu_with_branches": round((tot["salu"] + tot["nop"] + tot["branch"]) / pmc["salu_insts_per_launch"], 4),
            "SQ_INSTS_SMEM": pmc["smem_insts_per_launch"], "modelled_over_measured_smem": round(tot["smem"] / pmc["smem_insts_per_launch"], 4),
        }
        if pmc.get("valu_trans_f32_insts"):
            res["pmc_check"]["SQ_INSTS_VALU_TRANS_F32"] = pmc["valu_trans_f32_insts"]
            res["pmc_check"]["modelled_over_measured_transcendental"] = round(tot["Q"] / pmc["valu_trans_f32_insts"], 4)
        # what the miscount of the VALU total can do to the mean: were all of it plain (F) or all of it half-rate (H) instructions
        over = valu - pmc["valu_insts_per_launch"]
        res["mean_issue_cycles_bracket"] = sorted(round((cyc - over * c) / pmc["valu_insts_per_launch"], 4) for c in (F, H))
        if pmc.get("active_inst_valu_quadcycles"):
            qm = pmc["active_inst_valu_quadcycles"] - pmc["valu_insts_per_launch"]
            res["pmc_check"]["Q_measured(ACTIVE_INST_VALU-INSTS_VALU)"] = qm
            res["pmc_check"]["modelled_over_measured_Q"] = round(tot["Q"] / qm, 4) if qm else None
    s = json.dumps(res, indent=1)
    if out:
        open(out, "w").write(s)
    brief = {k: v for k, v in res.items() if k != "segments"}
    print(json.dumps(brief, indent=1))
    top = sorted(rows, key=lambda r: -r["valu_share"])[:12]
    for r in top:
        print("  mark %2d x%d  exec %12.0f  F %5.1f H %5.1f Q %4.1f salu %5.1f  share %.3f" % (r["mark"], r["copies"], r["executions"], r["F"], r["H"], r["Q"], r["salu"], r["valu_share"]))


if __name__ == "__main__":
    main()
