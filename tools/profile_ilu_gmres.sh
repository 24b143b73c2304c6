#!/bin/bash
# Kernel trace of one ILU-preconditioned GMRES(30) solve (tools/ilu_gmres_run.py): the timeline of two
# consecutive iterations in the middle of the last solve -- which kernels, how long, what gaps between them.
set -o pipefail
TAG=${1:-r3_prof_ilu_gmres}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $ROOT/tools/ilu_gmres_run.py ${@:2} > $OUT/trace.log 2> $OUT/trace.err || exit 1
cat $OUT/trace.log
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)[0]
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
    return name[:name.index("(")] if "(" in name else name
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(tr))))
# the brick solves of the last solve; an iteration = from one lower solve to the next
bricks = [i for i, r in enumerate(rows) if "trs_brick_pipelined" in r[2]]
if len(bricks) >= 8:
    mid = bricks[len(bricks) * 3 // 4 // 2 * 2]
    lo, hi = mid, bricks[bricks.index(mid) + 4]
else:  # no preconditioner: a window of eight launches
    lo = len(rows) * 3 // 4
    hi = lo + 8
with open(os.path.join(out, "timeline.md"), "w") as f:
    f.write("| kernel | start us | duration us | gap before us |\n|---|---|---|---|\n")
    t0 = rows[lo][0]
    for i in range(lo, hi):
        s, e, k = rows[i]
        f.write(f"| `{k}` | {(s - t0) / 1e3:.1f} | {(e - s) / 1e3:.1f} | {(s - rows[i - 1][1]) / 1e3:.1f} |\n")
    f.write(f"\ntwo iterations: {(rows[hi][0] - t0) / 1e3:.1f} us\n")
    arn = [(e - s) / 1e3 for s, e, k in rows if "gmres_arnoldi" in k][-96:]
    f.write("\nArnoldi launches of the last solve, us: " + " ".join(f"{d:.0f}" for d in arn) + f"\nmean {sum(arn) / max(len(arn), 1):.1f} us\n")
print(open(os.path.join(out, "timeline.md")).read())
PY
