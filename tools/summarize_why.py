#!/usr/bin/env python3
"""Condenses a tools/profile_why.sh directory: per strategy and counter the mean
per launch over the cold launches (first 64 dispatches of the SpMV kernel) and
the warm ones (next 64), plus the derived ratios the question needs."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
print(f"# PMC passes, 64 cold + 64 warm launches per pass ({os.path.basename(root)})\n")
for sdir in sorted(glob.glob(os.path.join(root, "*"))):
    if not os.path.isdir(sdir):
        continue
    vals = {}
    for f in glob.glob(os.path.join(sdir, "**", "*counter_collection.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "csr_" in r["Kernel_Name"] and "srow" not in r["Kernel_Name"]]
        byd = defaultdict(dict)
        for r in rows:
            d = byd[int(r["Dispatch_Id"])]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        ids = sorted(byd)
        for c in {c for d in byd.values() for c in d}:
            seq = [byd[i].get(c, 0.0) for i in ids]
            if len(seq) >= 128:
                vals[c] = (sum(seq[:64]) / 64, sum(seq[64:128]) / 64)
    if not vals:
        continue
    print(f"## strategy {os.path.basename(sdir)}\n")
    print("| counter | cold / launch | warm / launch |")
    print("|---|---|---|")
    for c in sorted(vals):
        print(f"| {c} | {vals[c][0]:.4g} | {vals[c][1]:.4g} |")
    nan = float("nan")
    g = lambda c, i: vals.get(c, (nan, nan))[i]
    print()
    for i, name in ((0, "cold"), (1, "warm")):
        wc = g("SQ_WAVE_CYCLES", i)
        line = [f"{name}:"]
        if wc == wc:
            line.append(f"wait_any {g('SQ_WAIT_ANY', i) / wc:.2f}, wait_inst {g('SQ_WAIT_INST_ANY', i) / wc:.3f}, "
                        f"active_inst {g('SQ_ACTIVE_INST_ANY', i) / wc:.3f} of wave-cycles;")
        if g("TCC_EA0_RDREQ_sum", i) == g("TCC_EA0_RDREQ_sum", i):
            line.append(f"L2 hit rate {g('TCC_HIT_sum', i) / (g('TCC_HIT_sum', i) + g('TCC_MISS_sum', i)):.3f}, "
                        f"mean EA read latency {g('TCC_EA0_RDREQ_LEVEL_sum', i) / g('TCC_EA0_RDREQ_sum', i):.0f} L2 cycles;")
        if g("SQ_LEVEL_WAVES", i) == g("SQ_LEVEL_WAVES", i):
            line.append(f"SQ_LEVEL_WAVES {g('SQ_LEVEL_WAVES', i):.4g}, SQ_INST_LEVEL_VMEM {g('SQ_INST_LEVEL_VMEM', i):.4g}")
        print(" ".join(line))
    print()
