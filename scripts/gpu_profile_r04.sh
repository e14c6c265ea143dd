# usage: bash scripts/gpu_profile_r04.sh   (through gpurun; the libraries and build_variants/marks_107.s are built first by
# scripts/prepare_profile_r04.sh in the build container) — everything profiles/r04/ holds for the final kernel of round 4:
#   PMC passes (one counter group per run, never mixed with tracing) of bench.py on bunny.json, the 64 000-triangle mesh
#   and the C4 grid -> counters_<tag>.json; segment execution counts (-DCTR_PROFILE build) -> valu_mix_dynamic_<tag>.json;
#   then — the fraction is now computable — the bench line and rocprofv3 --kernel-trace --stats of the same command.
set -x
REPO=$(pwd)
mkdir -p gpurun_out/p4
python3 -c "
import sys; sys.path.insert(0, '$REPO')
from cutrace_amd import scenes
scenes.make_bunny_grid('$REPO/build_variants/scenes'); scenes.make_dense_bunny('$REPO/build_variants/scenes', 3)"
bash scripts/gpu_pmc.sh r04fb > gpurun_out/p4/pmc_bunny.log 2>&1
BENCH_ARGS="--scene build_variants/scenes/bunny_dense3.json" WORKLOAD="bunny_dense3.json@1920x1080b5" bash scripts/gpu_pmc.sh r04fd > gpurun_out/p4/pmc_dense.log 2>&1
BENCH_ARGS="--scene build_variants/scenes/bunny_grid4x4.json" WORKLOAD="bunny_grid4x4.json@4096x4096b5" bash scripts/gpu_pmc.sh r04fc > gpurun_out/p4/pmc_c4.log 2>&1
CUTRACE_AMD_LIB=build_variants/profile.so python scripts/gpu_profile_mix.py --c4 > gpurun_out/p4/profile_counts.log 2>&1
mkdir -p profiles/r04
for t in "bunny r04fb" "dense64k r04fd" "c4 r04fc"; do set -- $t
  cp gpurun_out/pmc_$2_counters.json profiles/r04/counters_$1.json
  cp gpurun_out/pmc_$2_summary.txt profiles/r04/${1}_pmc_summary.txt
  cp gpurun_out/profile_counts_$1.json profiles/r04/profile_counts_$1.json
  python3 scripts/dynamic_mix.py build_variants/marks_107.s 107 gpurun_out/profile_counts_$1.json --pmc profiles/r04/counters_$1.json --out profiles/r04/valu_mix_dynamic_$1.json > gpurun_out/p4/mix_$1.txt 2>&1
done
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_r04.json 2> gpurun_out/bench_r04.err || { tail -20 gpurun_out/bench_r04.err; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r04 -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras > $REPO/gpurun_out/prof_r04.log 2>&1 || { tail -20 $REPO/gpurun_out/prof_r04.log; exit 1; }
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r04_dense -o trace -- python3 $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --skip-probe --no-extras --scene build_variants/scenes/bunny_dense3.json > $REPO/gpurun_out/prof_r04_dense.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $REPO/gpurun_out/prof_r04_c4 -o trace -- python3 $REPO/bench.py --steps 8 --warmup 2 --no-cpu-baseline --skip-probe --no-extras --workload c4 --scaling strong --roots rank0 > $REPO/gpurun_out/prof_r04_c4.log 2>&1
cd $REPO
for d in prof_r04 prof_r04_dense prof_r04_c4; do for f in $(find gpurun_out/$d -name "*kernel_stats.csv"); do cp $f gpurun_out/p4/${d}_kernel_stats.csv; head -4 $f; done; done
python scripts/gpu_configs.py > gpurun_out/p4/configs.txt 2>&1; cat gpurun_out/p4/configs.txt
cp -r profiles/r04 gpurun_out/p4/profiles_r04
tail -c 1500 gpurun_out/bench_r04.json
