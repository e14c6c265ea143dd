set -x
REPO=$(pwd)
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_pair.json 2> gpurun_out/bench_pair.err || { tail -20 gpurun_out/bench_pair.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_pair -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras > $REPO/gpurun_out/prof_pair.log 2>&1
cd $REPO
for f in $(find gpurun_out/prof_pair -name "*kernel_stats.csv"); do head -3 $f; done
python3 -c "
import json
b=json.loads(open('gpurun_out/bench_pair.json').read().strip().splitlines()[-1])
print(b['value'], b['ms_per_step'], b['roofline']['kernel_ms_avg'], b['roofline']['clock_ghz_this_run'], b['roofline']['frac'])
for l in open('gpurun_out/prof_pair.log'):
    if l.startswith('{\"metric\"'):
        b=json.loads(l); print('under rocprof:', b['value'], b['ms_per_step'], b['roofline']['kernel_ms_avg'], b['roofline'].get('clock_ghz_this_run'))
"
