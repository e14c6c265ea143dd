# usage: bash scripts/gpu_pmc_one.sh <tag> <lib.so> <scene.json>  — one PMC pass (instruction counts) of bench.py with the given library build
TAG=$1; LIB=$2; SCENE=$3
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
export CUTRACE_AMD_LIB=$REPO/$LIB
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $REPO/gpurun_out/pmc1_$TAG -o pmc -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --skip-probe --no-extras --scene $REPO/$SCENE > $REPO/gpurun_out/pmc1_$TAG.log 2>&1 || tail -5 $REPO/gpurun_out/pmc1_$TAG.log
cd $REPO
echo "== $TAG"; python3 scripts/pmc_summary.py gpurun_out/pmc1_$TAG
