# usage: bash scripts/gpu_bench.sh <tag>   (run on the GPU box through gpurun)
set -x
TAG=${1:-r01}
REPO=$(pwd)
mkdir -p gpurun_out
python bench.py --steps 10 --warmup 2 > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || { tail -20 gpurun_out/bench_$TAG.err; exit 1; }
cat gpurun_out/bench_$TAG.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_$TAG -o trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --skip-probe > $REPO/gpurun_out/prof_$TAG.log 2>&1 || { tail -20 $REPO/gpurun_out/prof_$TAG.log; exit 1; }
cd $REPO
find gpurun_out/prof_$TAG -name "*stats*" | head; 
for f in $(find gpurun_out/prof_$TAG -name "*kernel_stats.csv"); do head -5 $f; done
