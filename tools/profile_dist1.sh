#!/bin/bash
# Kernel trace of the distributed leg of bench.py with a world of one rank (GKOMI_BENCH_FORCE_DIST=1): the launches of a
# few CG iterations in the middle of the last solve.
set -o pipefail
TAG=${1:-r3_prof_dist1}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export GKOMI_BENCH_FORCE_DIST=1
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --p3-grid 128 --no-cpu-baseline > $OUT/trace.log 2> $OUT/trace.err || exit 1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
    return name[:name.index("(")] if "(" in name else name
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr))))
lo = len(rows) - 400
with open(os.path.join(out, "timeline.md"), "w") as f:
    f.write("| kernel | start us | duration us | gap before us |\n|---|---|---|---|\n")
    t0 = rows[lo][0]
    for i in range(lo, lo + 30):
        s, e, k = rows[i]
        f.write(f"| `{k[:80]}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - rows[i - 1][1]) / 1e3:.1f} |\n")
print(open(os.path.join(out, "timeline.md")).read())
PY
