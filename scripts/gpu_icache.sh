# usage: bash scripts/gpu_icache.sh [lib.so]   (through gpurun) — instruction-cache counters of the render kernel on bunny.json and the
# 64 000-triangle mesh (PMC pass of its own, never mixed with tracing) -> gpurun_out/icache_summary.txt
set -x
REPO=$(pwd)
LIB=${1:-}
[ -n "$LIB" ] && export CUTRACE_AMD_LIB=$REPO/$LIB
mkdir -p gpurun_out/icache
python3 -c "
import sys; sys.path.insert(0, '$REPO')
from cutrace_amd import scenes
scenes.make_dense_bunny('$REPO/build_variants/scenes', 3)"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $REPO/gpurun_out/icache/avail.txt 2>&1
grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQ_INST_LEVEL[A-Z_]*\|SQ_WAIT_IFETCH\|SQC_TC_INST[A-Z_]*" $REPO/gpurun_out/icache/avail.txt | sort -u > $REPO/gpurun_out/icache/names.txt
cat $REPO/gpurun_out/icache/names.txt
for t in "bunny " "dense --scene build_variants/scenes/bunny_dense3.json"; do set -- $t; tag=$1; shift
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $REPO/gpurun_out/icache/pmc_$tag -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe --no-extras "$@" > $REPO/gpurun_out/icache/$tag.log 2>&1 || tail -5 $REPO/gpurun_out/icache/$tag.log
  rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU --output-format csv -d $REPO/gpurun_out/icache/pmc2_$tag -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe --no-extras "$@" > $REPO/gpurun_out/icache/${tag}2.log 2>&1 || tail -5 $REPO/gpurun_out/icache/${tag}2.log
done
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/icache/pmc_bunny gpurun_out/icache/pmc2_bunny > gpurun_out/icache_summary_bunny.txt 2>&1
python3 scripts/pmc_summary.py gpurun_out/icache/pmc_dense gpurun_out/icache/pmc2_dense > gpurun_out/icache_summary_dense.txt 2>&1
grep -A12 "render_kernel" gpurun_out/icache_summary_bunny.txt gpurun_out/icache_summary_dense.txt
