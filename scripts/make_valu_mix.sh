# usage: bash scripts/make_valu_mix.sh [out.json]  — static VALU instruction mix of the shipped render kernel
# (every instantiation; the shipped default is render_kernel<43u> = BVH | PREFILTER | ANYHIT | FASTPOW, and
# render_kernel<107u> = the same compiled for 6 waves per SIMD, used for scenes with >= 1000 mesh triangles), priced with the issue costs scripts/valu_issue.hip
# measured on the box (profiles/r02/valu_issue.txt).  bench.py reads the result for roofline.frac.
set -e
OUT=${1:-profiles/r02/valu_mix.json}
TMP=$(mktemp -d)
hipcc --offload-arch=gfx950 -std=c++17 -O3 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt \
  -munsafe-fp-atomics -fno-slp-vectorize -Iinclude -Icutrace_amd/csrc -S --cuda-device-only -o $TMP/rk.s cutrace_amd/csrc/render_kernel.hip 2>/dev/null
python3 scripts/valu_mix.py $TMP/rk.s render_kernelILj > $OUT
rm -rf $TMP
cat $OUT
