# usage: bash scripts/gpu_pmc_quick.sh <tag> <scene.json> — instruction counters of one scene (one PMC pass)
TAG=$1; SCENE=$2
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $REPO/gpurun_out/pmcq_$TAG -o pmc -- python3 $REPO/bench.py --steps 4 --warmup 2 --no-cpu-baseline --skip-probe --no-extras --scene $SCENE > $REPO/gpurun_out/pmcq_$TAG.log 2>&1 || tail -5 $REPO/gpurun_out/pmcq_$TAG.log
cd $REPO
python3 scripts/pmc_summary.py gpurun_out/pmcq_$TAG | grep -A9 "43u"
