#!/bin/bash
# Where do the cycles of a brick's step loop go?  rocprofv3 --pmc passes (counters only, no tracing)
# over tools/trs_bricks_one.py.   output: gpurun_out/<tag>/<pass>/
set -o pipefail
TAG=${1:-bricks_pmc}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[sq1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS"
PASS[sq2]="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR"
PASS[sq3]="SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_INSTS_CBRANCH_NOT_TAKEN"
for p in sq1 sq2 sq3; do
  OUT=$ROOT/gpurun_out/$TAG/$p
  mkdir -p $OUT
  timeout -k 10 200 rocprofv3 --pmc ${PASS[$p]} --output-format csv -d $OUT -- python3 $ROOT/tools/trs_bricks_one.py 2 > $OUT.log 2>&1 || echo "pass $p failed (see $OUT.log)"
done
python3 - <<PY
import csv, glob, collections
for p in ("sq1", "sq2", "sq3"):
    for f in glob.glob("$ROOT/gpurun_out/$TAG/%s/**/*counter_collection.csv" % p, recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            if "pipelined" in r["Kernel_Name"]:
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
        for k, (v, c) in sorted(acc.items()):
            print(f"{p} {k:32s} {v / c:14.1f} per launch ({c} launches)")
PY
