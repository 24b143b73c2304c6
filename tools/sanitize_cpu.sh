#!/bin/bash
# CPU-side sanitizer pass (never on the GPU box): the oracle's C restatement and the
# host side of the C++ mirror built with AddressSanitizer + UndefinedBehaviorSanitizer, then the
# whole `-m "not gpu"` suite run against that oracle build, and the mirror's host test binary.
# usage: tools/sanitize_cpu.sh [log]      (run from the repo root)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r04_sanitizers.log}
OUT=/tmp/gkomi_asan
mkdir -p $OUT
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -g"
{
echo "# $(date -u +%F) tools/sanitize_cpu.sh: gcc $(gcc -dumpversion), flags: $SAN"
gcc -O1 -std=c99 -fPIC -fopenmp -ffp-contract=off -fvisibility=hidden -Wall $SAN -shared $ROOT/oracle/*.c -o $OUT/libgko_oracle.so -lm || exit 1
echo "## oracle (oracle/*.c) under ASan + UBSan: pytest -m 'not gpu'"
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
( cd $ROOT && GKO_ORACLE_LIB=$OUT/libgko_oracle.so LD_PRELOAD="$ASAN $UBSAN" ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
  UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 OMP_NUM_THREADS=2 \
  python -m pytest tests -x -q -m "not gpu" -p no:cacheprovider 2>&1 | tail -6 )
echo "## host mirror (ginkgo.hpp host paths, examples/host_api_test.cpp) under ASan + UBSan"
g++ -O1 -std=c++14 -Wall $SAN -I$ROOT/repo-8852-ginkgo_amd/include $ROOT/repo-8852-ginkgo_amd/examples/host_api_test.cpp -o $OUT/host_api_test \
    -L$ROOT/repo-8852-ginkgo_amd/lib -lgkomi -Wl,-rpath,$ROOT/repo-8852-ginkgo_amd/lib || exit 1
ASAN_OPTIONS=detect_leaks=1:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1 $OUT/host_api_test 2>&1 | tail -5
echo "exit code of host_api_test: $?"
# round 4: the HOST side of the brick analysis (csrc/trs_bricks.hip: grid coordinates from the dependency graph, pieces of the
# level order, the schedule builder) compiled with ASan + UBSan for the host only (-fno-gpu-sanitize: no device instrumentation,
# nothing of this runs on a GPU), linked with the other objects of the normal build, under the CPU replay tests
echo "## host analysis of the brick plan (csrc/trs_bricks.hip, host code only) under ASan + UBSan: tests/test_trs_bricks_analysis.py"
PKG=$ROOT/repo-8852-ginkgo_amd
/opt/rocm/bin/hipcc -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fsanitize=address,undefined -fno-gpu-sanitize -shared-libsan \
    -c $PKG/csrc/trs_bricks.hip -o $OUT/trs_bricks.o || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -shared-libsan $(ls $PKG/build/*.o | grep -v trs_bricks.o) $OUT/trs_bricks.o \
    -o $OUT/libgkomi.so || exit 1
RT=$(/opt/rocm/lib/llvm/bin/clang -print-file-name=libclang_rt.asan-x86_64.so)
( cd $ROOT && GKOMI_LIB=$OUT/libgkomi.so LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=halt_on_error=1:print_stacktrace=1 \
  python -m pytest tests/test_trs_bricks_analysis.py -x -q -p no:cacheprovider 2>&1 | tail -4 )
} 2>&1 | tee $LOG
