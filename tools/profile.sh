#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on the GPU box (run through
# gpurun).  Kernel-trace/stats and each PMC counter go in SEPARATE passes
# (MI355X_MICROARCH.md: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2; gpurun
# refuses --pmc mixed with tracing).  Output: gpurun_out/<tag>/...
set -o pipefail
TAG=${1:-prof_r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# (the legs that run the SAME kernel template on other matrices -- P3, config 3, the scattered-column classes -- are left out:
# per-kernel means would mix them with the headline matrix)
ARGS="--no-cpu-baseline --no-p3 --no-config3 --no-irregular --steps 200 --warmup 20"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 $ROOT/bench.py $ARGS --no-cg --no-config4 > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 $ROOT/bench.py $ARGS --no-cg --no-config4 > $OUT/pmc_write.log 2>&1 || exit 1
# calibration of FETCH_SIZE on a known byte count in a comparable access mix
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch_membench -- python3 $ROOT/tools/membench.py 1000 > $OUT/pmc_fetch_membench.log 2>&1 || exit 1
python3 $ROOT/tools/summarize_profile.py $OUT $OUT/traffic.json > $OUT/summary.md
cat $OUT/summary.md
