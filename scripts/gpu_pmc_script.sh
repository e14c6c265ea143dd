# usage: bash scripts/gpu_pmc_script.sh <tag> <python script + args>
set -x
TAG=$1; shift
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $REPO/gpurun_out/pmc_${TAG}_$name -o pmc -- python3 $REPO/$SCRIPT > $REPO/gpurun_out/pmc_${TAG}_$name.log 2>&1 || tail -5 $REPO/gpurun_out/pmc_${TAG}_$name.log; }
SCRIPT="$*"
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY
run sq2 SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmc_${TAG}_* > gpurun_out/pmc_${TAG}_summary.txt 2>&1
cat gpurun_out/pmc_${TAG}_summary.txt
