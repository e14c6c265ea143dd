"""Extract one render_kernel instantiation from the gfx950 assembly of render_kernel.hip and print its resource
summary.  usage: isa_extract.py <kernel.s> <variant number> [out.s]   (make the .s with scripts/make_valu_mix.sh's flags)"""
import re, sys
path, kv = sys.argv[1], sys.argv[2]
out = sys.argv[3] if len(sys.argv) > 3 else None
name = "_ZN12_GLOBAL__N_113render_kernelILj%sEEE" % kv
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(name) and l.rstrip().endswith(("Py", "Py:")) or (l.startswith(name) and ":" in l))
end = next(i for i in range(start, len(lines)) if lines[i].startswith("; Occupancy"))
body = lines[start:end + 1]
if out:
    open(out, "w").write("\n".join(body))
keys = ("codeLenInByte", "TotalNumSgprs", "NumVgprs:", "ScratchSize", "Occupancy")
for l in body:
    if any(k in l for k in keys):
        print(l.strip())
print("spill/reload lines:", sum(1 for l in body if "Spill" in l or "Reload" in l), " s_nop:", sum(1 for l in body if l.strip().startswith("s_nop")))
