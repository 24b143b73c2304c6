#!/bin/bash
# are the L2's fabric reads for gather misses 64-B or 128-B requests?  TCC_BUBBLE = "128-byte read requests sent to EA"
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_gather_pmc2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for case in '{"random": "uniform", "rows": 1000000, "nnz_per_row": 16}' '{"stencil": "5pt", "size": 1000}' '{"random": "powerlaw", "rows": 1000000, "nnz_per_row": 8}'; do
  i=$((i+1))
  for pass in "TCC_BUBBLE_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_RDREQ_DRAM_sum TCC_MISS_sum"; do
    tag=d${i}_$(echo $pass | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/gather_pmc.py "$case" csr_1pass > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; }
  done
done
python3 - $OUT <<'PY'
import csv, glob, statistics, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/d*_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "csr_" in k and "kernel" in k:
                acc[(k.split("<")[0].split("(")[0][-30:], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{d.rstrip('/').split('/')[-1]:50s} {k:32s} {c:26s} mean {statistics.mean(v):16.1f}  n={len(v)}")
PY
