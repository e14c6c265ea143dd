# usage: bash scripts/gpu_profile_r03.sh   (through gpurun; the libraries and build_variants/marks_107.s are built first by
# scripts/prepare_profile_r03.sh in the build container) — everything profiles/r03/ holds for the final kernel of round 3:
#   PMC passes (one counter group per run, never mixed with tracing) of bench.py on bunny.json, the 64 000-triangle mesh
#   and the C4 grid -> counters_<tag>.json; segment execution counts (-DCTR_PROFILE build) -> valu_mix_dynamic_<tag>.json;
#   then — the fraction is now computable — the bench line and rocprofv3 --kernel-trace --stats of the same command.
set -x
REPO=$(pwd)
mkdir -p gpurun_out/p3
python3 -c "
import sys; sys.path.insert(0, '$REPO')
from cutrace_amd import scenes
scenes.make_bunny_grid('$REPO/build_variants/scenes'); scenes.make_dense_bunny('$REPO/build_variants/scenes', 3)"
bash scripts/gpu_pmc.sh r03fb > gpurun_out/p3/pmc_bunny.log 2>&1
BENCH_ARGS="--scene build_variants/scenes/bunny_dense3.json" WORKLOAD="bunny_dense3.json@1920x1080b5" bash scripts/gpu_pmc.sh r03fd > gpurun_out/p3/pmc_dense.log 2>&1
BENCH_ARGS="--scene build_variants/scenes/bunny_grid4x4.json" WORKLOAD="bunny_grid4x4.json@4096x4096b5" bash scripts/gpu_pmc.sh r03fc > gpurun_out/p3/pmc_c4.log 2>&1
CUTRACE_AMD_LIB=build_variants/profile.so python scripts/gpu_profile_mix.py --c4 > gpurun_out/p3/profile_counts.log 2>&1
mkdir -p profiles/r03
for t in "bunny r03fb" "dense64k r03fd" "c4 r03fc"; do set -- $t
  cp gpurun_out/pmc_$2_counters.json profiles/r03/counters_$1.json
  cp gpurun_out/pmc_$2_summary.txt profiles/r03/${1}_pmc_summary.txt
  cp gpurun_out/profile_counts_$1.json profiles/r03/profile_counts_$1.json
  python3 scripts/dynamic_mix.py build_variants/marks_107.s 107 gpurun_out/profile_counts_$1.json --pmc profiles/r03/counters_$1.json --out profiles/r03/valu_mix_dynamic_$1.json > gpurun_out/p3/mix_$1.txt 2>&1
done
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r03.json 2> gpurun_out/bench_r03.err || { tail -20 gpurun_out/bench_r03.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r03 -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras > $REPO/gpurun_out/prof_r03.log 2>&1 || { tail -20 $REPO/gpurun_out/prof_r03.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r03_dense -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras --scene build_variants/scenes/bunny_dense3.json > $REPO/gpurun_out/prof_r03_dense.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r03_c4 -o trace -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --skip-probe --no-extras --workload c4 --scaling strong --roots rank0 > $REPO/gpurun_out/prof_r03_c4.log 2>&1
cd $REPO
for d in prof_r03 prof_r03_dense prof_r03_c4; do for f in $(find gpurun_out/$d -name "*kernel_stats.csv"); do cp $f gpurun_out/p3/${d}_kernel_stats.csv; head -4 $f; done; done
python scripts/gpu_configs.py > gpurun_out/p3/configs.txt 2>&1; cat gpurun_out/p3/configs.txt
cp -r profiles/r03 gpurun_out/p3/profiles_r03
tail -c 1500 gpurun_out/bench_r03.json
