# usage: bash scripts/gpu_traffic.sh <tag>  — HBM-side traffic of the render kernel per launch (FETCH_SIZE and WRITE_SIZE,
# each in its own PMC pass, MI355X_MICROARCH.md §HBM) for bunny.json and its 64 000-triangle version, plus the wait / scalar
# cache counters of the dense scene; summaries -> gpurun_out/traffic_<tag>.txt
TAG=${1:-r03}
REPO=$(pwd)
mkdir -p gpurun_out
DENSE=$(python3 -c "
import sys; sys.path.insert(0, '$REPO')
from cutrace_amd import scenes
print(scenes.make_dense_bunny('$REPO/build_variants/scenes', 3))")
cd /tmp && export TMPDIR=/tmp
run() { # name scene counters...
  name=$1; scene=$2; shift; shift
  rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/tr_${TAG}_$name -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe --no-extras --scene $scene > $REPO/gpurun_out/tr_${TAG}_$name.log 2>&1 || tail -5 $REPO/gpurun_out/tr_${TAG}_$name.log
}
run bunny_fetch $REPO/scene/bunny.json FETCH_SIZE
run bunny_write $REPO/scene/bunny.json WRITE_SIZE
run dense_fetch $DENSE FETCH_SIZE
run dense_write $DENSE WRITE_SIZE
run dense_sq $DENSE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run dense_sqc $DENSE SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
cd $REPO
{ echo "== bunny.json@1920x1080 b5"; python3 scripts/pmc_summary.py gpurun_out/tr_${TAG}_bunny_fetch gpurun_out/tr_${TAG}_bunny_write;
  echo "== bunny_dense3.json (64 000 triangles) @1920x1080 b5"; python3 scripts/pmc_summary.py gpurun_out/tr_${TAG}_dense_fetch gpurun_out/tr_${TAG}_dense_write gpurun_out/tr_${TAG}_dense_sq gpurun_out/tr_${TAG}_dense_sqc; } > gpurun_out/traffic_${TAG}.txt 2>&1
cat gpurun_out/traffic_${TAG}.txt
