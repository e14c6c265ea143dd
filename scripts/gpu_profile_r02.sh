# usage: bash scripts/gpu_profile_r02.sh — everything profiles/r02/ holds for the final kernel of round 2 (run through gpurun)
set -x
REPO=$(pwd)
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err || { tail -20 gpurun_out/bench_r02.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r02 -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras > $REPO/gpurun_out/prof_r02.log 2>&1 || { tail -20 $REPO/gpurun_out/prof_r02.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r02_dense -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras --scene build_variants/scenes/bunny_dense3.json > $REPO/gpurun_out/prof_r02_dense.log 2>&1 || { tail -20 $REPO/gpurun_out/prof_r02_dense.log; exit 1; }
cd $REPO
BENCH_ARGS="--scene build_variants/scenes/bunny_dense3.json" WORKLOAD="bunny_dense3.json@1920x1080b5" bash scripts/gpu_pmc.sh r02dense > gpurun_out/pmc_r02dense.log 2>&1
for f in $(find gpurun_out/prof_r02 gpurun_out/prof_r02_dense -name "*kernel_stats.csv"); do echo $f; head -4 $f; done
