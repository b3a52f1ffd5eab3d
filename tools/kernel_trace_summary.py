#!/usr/bin/env python3
"""Summarises a rocprofv3 --kernel-trace CSV: per (kernel, grid size) call count and mean duration (us)."""
import csv
import glob
import sys
from collections import defaultdict

paths = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
acc = defaultdict(lambda: [0, 0.0])
for p in paths:
    for r in csv.DictReader(open(p)):
        name = r["Kernel_Name"].split("(")[0][:48]
        grid = r.get("Grid_Size") or r.get("Grid_Size_X")
        wg = r.get("Workgroup_Size") or r.get("Workgroup_Size_X")
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = acc[(name, grid, wg)]
        a[0] += 1
        a[1] += d
filt = sys.argv[2] if len(sys.argv) > 2 else ""
for (name, grid, wg), (n, tot) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if filt in name:
        print(f"{name:50s} grid={grid:>9s} wg={wg:>5s} calls={n:5d} mean_us={tot / n:10.1f} total_ms={tot / 1e3:9.2f}")
