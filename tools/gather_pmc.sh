#!/bin/bash
# L2 hit rate and fabric-side read requests of the SpMV kernels on the gather-bound matrix classes
# (separate --pmc passes, no tracing: MI355X_MICROARCH.md "rocprofv3 PMC slots").  Output: gpurun_out/r04_gather_pmc/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_gather_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1 || true
i=0
for case in '{"random": "uniform", "rows": 1000000, "nnz_per_row": 16}' '{"random": "powerlaw", "rows": 1000000, "nnz_per_row": 8}' '{"random": "local", "rows": 1000000, "nnz_per_row": 16, "bandwidth": 2000}'; do
  i=$((i+1))
  for pass in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_REQ_sum TCC_READ_sum" "FETCH_SIZE"; do
    tag=c${i}_$(echo $pass | tr ' ' '_')
    timeout -k 10 200 rocprofv3 --pmc $pass --output-format csv -d $OUT/$tag -- python3 $ROOT/tools/gather_pmc.py "$case" > $OUT/$tag.log 2>&1 || { echo "pass $tag failed"; tail -5 $OUT/$tag.log; }
  done
done
python3 - $OUT <<'PY'
import csv, glob, statistics, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(out + "/c*_*/")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "csr_" in k or "coo_" in k:
                acc[(k.split("<")[0].split("(")[0][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(acc.items()):
            print(f"{d.rstrip('/').split('/')[-1]:40s} {k:40s} {c:24s} mean {statistics.mean(v):16.1f}  n={len(v)}")
PY
