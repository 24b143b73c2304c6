#!/bin/bash
# CPU-side sanitizer pass (never on the GPU box): the oracle's C restatement and the
# host side of the C++ mirror built with AddressSanitizer + UndefinedBehaviorSanitizer, then the
# whole `-m "not gpu"` suite run against that oracle build, and the mirror's host test binary.
# usage: tools/sanitize_cpu.sh [log]      (run from the repo root)
set -o pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
LOG=${1:-$ROOT/profiles/r03_sanitizers.log}
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
} 2>&1 | tee $LOG
