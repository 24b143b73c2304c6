#!/bin/bash
# FETCH_SIZE of the 256^3 SpMV for three tile -> XCD maps (no swizzle / 16 tiles per XCD / one eighth each)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for bits in 258 2562 2; do
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $ROOT/gpurun_out/r3_p3pmc_$bits -- python3 $ROOT/tools/p3_pmc.py 256 3072 $bits > $ROOT/gpurun_out/r3_p3pmc_$bits.log 2>&1 || exit 1
done
python3 - $ROOT <<'PY'
import csv, glob, statistics, sys
root = sys.argv[1]
alg_reads = 12 * 117047296 + 4 * (16777216 + 1) + 8 * 16777216
for bits, name in ((258, "no swizzle"), (2562, "16 tiles per XCD"), (2, "one eighth per XCD")):
    f = glob.glob(f"{root}/gpurun_out/r3_p3pmc_{bits}/**/*counter_collection.csv", recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if r["Counter_Name"] == "FETCH_SIZE" and "csr_split_kernel" in r["Kernel_Name"]]
    b = 2 * statistics.mean(v) * 1024   # the guide's gfx950 correction (x2), calibrated in tools/profile.sh
    print(f"{name:20s}: FETCH_SIZE mean {statistics.mean(v):.0f} KiB x 2 x 1024 = {b / 1e9:.3f} GB read per launch = {b / alg_reads:.3f} x the algorithmic reads ({alg_reads / 1e9:.3f} GB), {len(v)} launches")
PY
