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
# the 16.7 M-row legs apart from the 1 M-row ones: launches of more than 30 us
python3 - $OUT <<'PY'
import csv, glob, os, statistics, sys
out = sys.argv[1]
tr = glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True)
if tr:
    def short(name):
        name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
        return name[:name.index("(")] if "(" in name else name
    by = {}
    for r in csv.DictReader(open(tr[0])):
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        if d > 30.0 and any(t in k for t in ("cg_fused_step", "csr_split_kernel", "compress_partials")) or (k == "compress_partials_kernel"):
            by.setdefault(k, []).append(d)
    n3, nnz3 = 256 ** 3, 7 * 256 ** 3 - 6 * 256 ** 2
    model = {"cg_fused_step2_kernel": 6 * 8 * n3, "cg_fused_step1_kernel": 3 * 8 * n3}
    with open(os.path.join(out, "p3_kernels.md"), "w") as f:
        f.write("| kernel (launches of the 256^3 legs only) | launches | median us | bytes moved (model) | TB/s | of 8 TB/s |\n|---|---|---|---|---|---|\n")
        for k, v in sorted(by.items()):
            b = model.get(k, (12 * nnz3 + 4 * (n3 + 1) + 16 * n3) if "csr_split" in k else 0)
            med = statistics.median(v)
            f.write(f"| `{k}` | {len(v)} | {med:.1f} | {b} | {b / med / 1e6 if b else 0:.2f} | {b / med / 8e6 if b else 0:.3f} |\n")
    print(open(os.path.join(out, "p3_kernels.md")).read())
PY
