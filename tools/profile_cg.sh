#!/bin/bash
# kernel-trace --stats of the solver legs of bench.py (CG on P2 and P3): which kernels an iteration runs and
# what each costs.  Output: gpurun_out/<tag>/
set -o pipefail
TAG=${1:-r3_prof_cg}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline --steps 50 --warmup 10 ${@:2} > $OUT/trace.log 2> $OUT/trace.err || exit 1
python3 - $OUT <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
st = glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
    return name[:name.index("(")] if "(" in name else name
with open(os.path.join(out, "kernel_stats.md"), "w") as f:
    f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in csv.DictReader(open(st)):
        f.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
print(open(os.path.join(out, "kernel_stats.md")).read())
PY
