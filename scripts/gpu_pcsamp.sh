set -x
REPO=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
rocprofv3 -L 2>/dev/null | grep -i -A3 "pc.sampl" | head -20
rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method host_trap --pc-sampling-unit time --pc-sampling-interval 1 --output-format csv -d $REPO/gpurun_out/pcs -o pcs -- python3 $REPO/bench.py --steps 30 --warmup 2 --no-cpu-baseline > $REPO/gpurun_out/pcs.log 2>&1
tail -5 $REPO/gpurun_out/pcs.log
ls -la $REPO/gpurun_out/pcs | head
