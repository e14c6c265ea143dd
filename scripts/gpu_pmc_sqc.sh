# usage: bash scripts/gpu_pmc_sqc.sh <tag> — scalar-cache / instruction-cache PMC passes of bench.py
set -x
TAG=${1:-sqc}
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$name -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 $REPO/gpurun_out/pmc_${TAG}_$name.log; }
}
run d1 SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE
run i1 SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL
run l1 SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM_NORM SQ_INSTS_SMEM SQ_IFETCH
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_* > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat gpurun_out/pmc_${TAG}_summary.txt
