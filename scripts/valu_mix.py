"""Static VALU instruction mix of a kernel's gfx950 ISA, priced with the issue costs measured by
scripts/valu_issue.hip (profiles/r02/valu_issue.txt): SIMD cycles per wave64 instruction with >= 2 waves resident.

  F  2.2  plain VALU (fma/mul/add/sub/min/max/logic ...) whose operands are VGPRs or inline constants
  H  4.1  reads an SGPR or a literal, any compare, packed f32, min3/max3/med3, cndmask, readlane/writelane,
          v_mov from an SGPR, 64-bit integer ops, div_scale/div_fmas/div_fixup
  Q  8.1  transcendental (rcp, rsq, sqrt, exp, log)
usage: valu_mix.py <kernel.s> [mangled-name-substring]   -> JSON on stdout"""
import json, re, sys

F, H, Q = 2.2, 4.1, 8.1
TRANS = ("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")
HALF_OPS = ("v_cmp", "v_cmpx", "v_pk_", "v_max3", "v_min3", "v_med3", "v_cndmask", "v_readlane", "v_writelane",
            "v_readfirstlane", "v_div_scale", "v_div_fmas", "v_div_fixup", "v_mad_u64", "v_lshl_add_u64", "v_mov_b64",
            "v_lshlrev_b64", "v_add_co", "v_addc_co", "v_sub_co", "v_subb_co", "v_mul_lo", "v_mul_hi", "v_mad_", "v_bfe",
            "v_perm", "v_alignbit", "v_ldexp", "v_frexp", "v_class")


def classify(line):
    op = line.split()[0]
    if op.startswith(TRANS):
        return "Q"
    if op.startswith(HALF_OPS):
        return "H"
    ops = line[len(op):]
    if re.search(r"(?<![a-z\[])s\d+|s\[\d+:\d+\]|vcc|exec|0x[0-9a-f]+|m0", ops):  # SGPR / literal operand
        return "H"
    return "F"


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    out, cur, counts = {}, None, None
    for raw in open(path):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = m.group(1)
            counts = {"F": 0, "H": 0, "Q": 0, "salu": 0, "smem": 0, "vmem": 0, "lds": 0}
            out[cur] = counts
            continue
        if cur is None or not line or line.startswith((";", ".")):
            continue
        if line.startswith("s_endpgm"):
            cur = None
            continue
        if line.startswith("v_"):
            counts[classify(line)] += 1
        elif line.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime")):
            counts["smem"] += 1
        elif line.startswith("s_") and not line.startswith(("s_waitcnt", "s_nop")):
            counts["salu"] += 1
        elif line.startswith(("global_", "flat_", "buffer_", "scratch_")):
            counts["vmem"] += 1
        elif line.startswith("ds_"):
            counts["lds"] += 1
    res = {}
    for k, c in out.items():
        if want not in k:
            continue
        n = c["F"] + c["H"] + c["Q"]
        if n == 0:
            continue
        c = dict(c)
        c["valu"] = n
        c["mean_issue_cycles"] = round((c["F"] * F + c["H"] * H + c["Q"] * Q) / n, 3)
        res[k] = c
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
