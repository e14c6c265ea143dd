# usage: bash scripts/gpu_pmc.sh <tag> [bench args]  — PMC passes of bench.py (each counter group its own run;
# --pmc is never combined with tracing), then profiles-ready summaries under gpurun_out/:
#   pmc_<tag>_summary.txt   mean per dispatch of every counter, per kernel
#   pmc_<tag>_counters.json what bench.py's roofline reads (copy to profiles/r02/counters.json)
set -x
TAG=${1:-r02}
shift
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$name -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe --no-extras $BENCH_ARGS > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 $REPO/gpurun_out/pmc_${TAG}_$name.log; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_LDS
run sqc SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_* > gpurun_out/pmc_${TAG}_summary.txt 2>&1
python3 scripts/pmc_counters.py gpurun_out/pmc_${TAG} "render_kernel<" "${WORKLOAD:-bunny.json@1920x1080b5}" gpurun_out/pmc_${TAG}_counters.json
cat gpurun_out/pmc_${TAG}_summary.txt
