#!/bin/bash
# Why is the cold launch slower than the byte count says?  Separate rocprofv3
# --pmc passes (8 SQ slots, 4 TCC slots per pass; never mixed with tracing) over
# tools/why_spmv.py = 64 cold + 64 warm launches of ONE kernel.
# usage: profile_why.sh <tag> <strategy ...>     output: gpurun_out/<tag>/<strategy>/<pass>/
set -o pipefail
TAG=${1:-why_r2}; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
PASS[sq2]="SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS"
PASS[tcc1]="TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum"
PASS[tcc2]="TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_32B_sum TCC_REQ_sum"
PASS[grbm]="GRBM_GUI_ACTIVE GRBM_COUNT"
for strat in "$@"; do
  for p in sq1 sq2 tcc1 tcc2 grbm; do
    OUT=$ROOT/gpurun_out/$TAG/${strat//:/_}/$p
    mkdir -p $OUT
    timeout -k 10 200 rocprofv3 --pmc ${PASS[$p]} --output-format csv -d $OUT -- python3 $ROOT/tools/why_spmv.py $strat 64 > $OUT.log 2>&1 || echo "pass $p for $strat failed (see $OUT.log)"
  done
done
python3 $ROOT/tools/summarize_why.py $ROOT/gpurun_out/$TAG
