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
    # the plain (non-advanced, no dot epilogue) instantiation, swizzled or not
    rows = [r for r in csv.DictReader(open(tr)) if "csr_stream_kernel<256, 1, 1536, false, true, false" in r["Kernel_Name"]
            or "csr_stream_kernel<256, 1, 1536, false, false, false" in r["Kernel_Name"]]
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows]
    if len(d) >= 440:
        print(f"\ncsr_stream_kernel, bench order = 20 warm-up + 200 cold (rotating copies) + 20 + 200 warm: "
              f"cold mean {statistics.mean(d[20:220])/1e3:.2f} us, warm mean {statistics.mean(d[240:440])/1e3:.2f} us")


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
