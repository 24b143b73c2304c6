#!/bin/bash
# Kernel trace of a fused Krylov solver on the 108^3 system (tools/krylov_run.py <solver>): the launches of two
# consecutive iterations in the middle of the last solve -- which kernels, how long, what gaps.
set -o pipefail
WHICH=${1:-bicgstab}
TAG=${2:-r3_prof_$WHICH}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/krylov_run.py $WHICH > $OUT/trace.log 2> $OUT/trace.err || exit 1
cat $OUT/trace.log
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
    return name[:name.index("(")] if "(" in name else name
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr))))
spmv = [i for i, r in enumerate(rows) if "csr_" in r[2] and "make_srow" not in r[2]]
lo = spmv[len(spmv) * 3 // 4]
hi = min(lo + 28, len(rows) - 1)
with open(os.path.join(out, "timeline.md"), "w") as f:
    f.write("| kernel | start us | duration us | gap before us |\n|---|---|---|---|\n")
    t0 = rows[lo][0]
    for i in range(lo, hi):
        s, e, k = rows[i]
        f.write(f"| `{k[:70]}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - rows[i - 1][1]) / 1e3:.1f} |\n")
print(open(os.path.join(out, "timeline.md")).read())
PY
