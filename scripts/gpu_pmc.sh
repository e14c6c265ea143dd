# usage: bash scripts/gpu_pmc.sh <tag>  — PMC passes (each its own run; --pmc never combined with tracing)
set -x
TAG=${1:-r01}
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$name -o pmc -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --skip-probe > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1 || { tail -5 $REPO/gpurun_out/pmc_${TAG}_$name.log; }
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_FLAT
run fetch FETCH_SIZE
run write WRITE_SIZE
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_* > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat gpurun_out/pmc_${TAG}_summary.txt
