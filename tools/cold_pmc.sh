#!/bin/bash
# Fabric reads in flight x latency for the headline SpMV, cold (rotating copies, nontemporal) vs warm (one matrix) vs the
# pure streaming probe: TCC_EA0_RDREQ_LEVEL / TCC_EA0_RDREQ = mean read latency in L2 clocks; requests x latency / kernel
# time = requests in flight.  One --pmc pass, no tracing.  Output: gpurun_out/r04_cold_pmc/
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/r04_cold_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-cpu-baseline --no-p3 --no-config3 --no-config4 --no-cg --steps 100 --warmup 10 --min-region-seconds 0.0"
timeout -k 10 300 rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum --output-format csv -d $OUT/level -- python3 $ROOT/bench.py $ARGS > $OUT/level.log 2>&1 || { tail -5 $OUT/level.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
python3 - $OUT <<'PY'
import csv, glob, statistics, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/level/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "csr_split_kernel" in k or "diag_stream" in k:
            acc[k[:110]][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = collections.defaultdict(list)
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "csr_split_kernel" in k or "diag_stream" in k:
            dur[k[:110]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(acc):
    req = statistics.mean(acc[k]["TCC_EA0_RDREQ_sum"]); lvl = statistics.mean(acc[k]["TCC_EA0_RDREQ_LEVEL_sum"])
    us = statistics.median(dur[k]) if dur.get(k) else float("nan")
    lat = lvl / req
    print(f"{k}\n    launches {len(acc[k]['TCC_EA0_RDREQ_sum'])}, fabric reads {req:.0f}, mean latency {lat:.0f} L2 clocks, kernel {us:.2f} us (trace run), "
          f"reads in flight ~ {req * lat / (us * 2100.0):.0f} (at 2.1 GHz)")
PY
