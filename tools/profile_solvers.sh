#!/bin/bash
# rocprofv3 kernel-trace stats of the preconditioner / solver kernels on the
# config-3/4 stand-ins (tools/tune_solvers.py) and of the benchmark harness
# (all formats, all solvers).  Run through gpurun; output gpurun_out/<tag>/.
set -o pipefail
TAG=${1:-prof_solvers_r1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/solvers -- python3 $ROOT/tools/tune_solvers.py > $OUT/solvers.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/solvers/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$OUT/solvers_kernel_stats.csv", "w") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "percent"])
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
            name = name[:name.index("(")] if "(" in name else name
            w.writerow([name, r["Calls"], f"{float(r['AverageNs'])/1e3:.2f}", f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", r["Percentage"]])
PY
head -40 $OUT/solvers_kernel_stats.csv
# the Krylov drivers (reference sequences and fused single-rhs drivers) on the 1M-row problems
cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/krylov -- python3 $ROOT/tools/tune_krylov.py > $OUT/krylov.log 2>&1 || exit 1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/krylov/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("$OUT/krylov_kernel_stats.csv", "w") as o:
        w = csv.writer(o)
        w.writerow(["kernel", "calls", "avg_us", "min_us", "max_us", "percent"])
        for r in rows:
            name = r["Name"].replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
            name = name[:name.index("(")] if "(" in name else name
            w.writerow([name, r["Calls"], f"{float(r['AverageNs'])/1e3:.2f}", f"{float(r['MinNs'])/1e3:.2f}", f"{float(r['MaxNs'])/1e3:.2f}", r["Percentage"]])
PY
head -30 $OUT/krylov_kernel_stats.csv
