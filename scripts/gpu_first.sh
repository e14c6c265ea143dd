set -x
mkdir -p gpurun_out
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; echo "smoke exit $?" >> gpurun_out/smoke.log
tail -5 gpurun_out/smoke.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest exit $?" >> gpurun_out/pytest_gpu.log
tail -30 gpurun_out/pytest_gpu.log
