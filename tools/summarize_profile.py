#!/usr/bin/env python3
"""Condenses a tools/profile.sh output directory into a markdown summary:
per-kernel average durations (kernel-trace --stats) and per-launch HBM
traffic from the FETCH_SIZE / WRITE_SIZE passes."""
import csv
import glob
import os
import statistics
import sys

out = sys.argv[1]


def find(sub, suffix):
    hits = glob.glob(os.path.join(out, sub, "**", "*" + suffix), recursive=True)
    return hits[0] if hits else None


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void gkomi::", "").replace("gkomi::", "")
    return name[:name.index("(")] if "(" in name else name


print(f"# rocprofv3 summary ({os.path.basename(out)})\n")
st = find("trace", "kernel_stats.csv")
if st:
    print("## kernel-trace --stats (bench.py --steps 200 --warmup 20)\n")
    print("| kernel | calls | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|")
    for r in csv.DictReader(open(st)):
        print(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | "
              f"{float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |")
tr = find("trace", "kernel_trace.csv")
if tr:
    # bench.py's cold leg runs the nontemporal instantiation of the nonzero-split kernel
    # (template arguments ..., Dot = false, NT = true, ...), its warm leg the plain one
    by = {}
    for r in csv.DictReader(open(tr)):
        if "csr_split_kernel" in r["Kernel_Name"]:
            by.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, d in sorted(by.items()):
        leg = "cold leg (rotating copies, nontemporal streams)" if "false, true, false, true" in k else \
            "warm leg (same matrix every step)" if "false, true, false, false" in k else "other"
        print(f"\n`{k}`: {len(d)} launches, mean {statistics.mean(d)/1e3:.2f} us, median {statistics.median(d)/1e3:.2f} us, "
              f"min {min(d)/1e3:.2f} us -- {leg}")


def pmc(sub, counter):
    f = find(sub, "counter_collection.csv")
    if not f:
        return {}
    acc = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        acc.setdefault(short(r["Kernel_Name"]), []).append(float(r["Counter_Value"]))
    return acc


fetch, write = pmc("pmc_fetch", "FETCH_SIZE"), pmc("pmc_write", "WRITE_SIZE")
cal = pmc("pmc_fetch_membench", "FETCH_SIZE")
if fetch or write:
    print("\n## PMC passes (per launch, KiB as reported; x1024 = bytes)\n")
    print("| kernel | launches | FETCH_SIZE mean | WRITE_SIZE mean |")
    print("|---|---|---|---|")
    for k in sorted(set(fetch) | set(write)):
        f = statistics.mean(fetch[k]) if k in fetch else float("nan")
        w = statistics.mean(write[k]) if k in write else float("nan")
        print(f"| `{k}` | {len(fetch.get(k, write.get(k, [])))} | {f:.0f} | {w:.0f} |")
if cal:
    print("\n## FETCH_SIZE calibration (tools/membench.py, known read bytes per launch = 71,952,004 + 16 B)\n")
    for k, v in cal.items():
        print(f"- `{k}`: mean FETCH_SIZE {statistics.mean(v):.0f} KiB over {len(v)} launches "
              f"-> {statistics.mean(v)*1024/71952004:.3f} of the bytes actually read")

# per-launch HBM traffic of the cold leg's kernel for bench.py's roofline.traffic
cold = [k for k in set(fetch) & set(write) if "csr_split_kernel" in k and "false, true, false, true" in k]
if cold and len(sys.argv) > 2:
    import json
    k = cold[0]
    t = int((2 * statistics.mean(fetch[k]) + statistics.mean(write[k])) * 1024)
    json.dump({"kernel": k, "fetch_size_kib_mean": statistics.mean(fetch[k]), "write_size_kib_mean": statistics.mean(write[k]),
               "traffic_bytes_per_launch": t,
               "formula": "(2 * FETCH_SIZE + WRITE_SIZE) * 1024: the gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md, "
                          "calibrated in the same run on tools/membench.hip",
               "algorithmic_bytes_per_launch": 79952004}, open(sys.argv[2], "w"), indent=1)
    print(f"\nHBM traffic per launch of the cold leg: {t} B vs 79952004 algorithmic ({t/79952004:.3f}x) -> {sys.argv[2]}")
