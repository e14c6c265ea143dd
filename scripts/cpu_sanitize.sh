# usage: bash scripts/cpu_sanitize.sh   (build container, CPU only) — the host-side code (scene loader / JSON / STL / JPEG writers,
# libcutrace_host.so) and the C oracle under AddressSanitizer + UndefinedBehaviorSanitizer, driven by the CPU tests that
# exercise them.  (The GPU library cannot be sanitised on this pool; its host half — ctr_api.cpp, bvh.cpp — needs a device.)
set -e
OUT=${TMPDIR:-/tmp}/ctr_san
mkdir -p $OUT
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
g++ -std=c++17 -ffp-contract=off -fPIC -pthread -Wall -Iinclude $SAN -shared -o $OUT/libcutrace_host_san.so cutrace_amd/host/scene_host.cpp cutrace_amd/host/images.cpp
gcc -ffp-contract=off -fPIC -Wall -Wno-unused-function -pthread $SAN -shared -o $OUT/libctr_oracle_san.so oracle/ctr_oracle.c -lm
export LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=0:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export CUTRACE_HOST_LIB=$OUT/libcutrace_host_san.so CUTRACE_ORACLE_LIB=$OUT/libctr_oracle_san.so
python -m pytest tests/test_loader.py tests/test_oracle_golden.py -x -q -m "not gpu" -p no:cacheprovider
python scripts/cpu_fuzz_loader.py 3000
# the BVH builders (host half of the GPU library that needs no device) on random and degenerate inputs
g++ -std=c++17 -O1 $SAN -Icutrace_amd/csrc -Iinclude -o $OUT/bvh_check scripts/bvh_check.cpp cutrace_amd/csrc/bvh.cpp
env -u LD_PRELOAD $OUT/bvh_check
