# usage: bash scripts/prepare_profile_r04.sh  (build container) — the libraries and the marked assembly scripts/gpu_profile_r04.sh needs
set -e
python -c "from cutrace_amd import build; build.build_all()"
python scripts/build_variant.py profile:-DCTR_PROFILE
hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -munsafe-fp-atomics \
  -fno-slp-vectorize -Iinclude -Icutrace_amd/csrc -DCTR_MARKS -S --cuda-device-only -o build_variants/marks_all.s cutrace_amd/csrc/render_kernel.hip 2>/dev/null
python3 scripts/isa_extract.py build_variants/marks_all.s 107 build_variants/marks_107_body.s
# (dynamic_mix.py looks the instantiation up by its mangled name: keep the whole file, trimmed to that kernel)
python3 - <<'PY'
import re
src = open("build_variants/marks_all.s").read().split("\n")
name = "_ZN12_GLOBAL__N_113render_kernelILj107EEE"
a = next(i for i, l in enumerate(src) if l.startswith(name) and ":" in l)
b = next(i for i in range(a, len(src)) if src[i].startswith("; Occupancy"))
open("build_variants/marks_107.s", "w").write("\n".join(src[a:b + 1]) + "\n")
PY
rm -f build_variants/marks_all.s build_variants/marks_107_body.s
ls -la build_variants/marks_107.s build_variants/profile.so
