#!/usr/bin/env python3
"""Idle gaps of the GPU inside one proof period (quotient kernel to quotient kernel) of a rocprofv3 kernel trace, all streams merged:
usage: gap_timeline.py <dir with *_kernel_trace.csv> [min_gap_us]  -> gaps with the kernels on either side"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
min_gap = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"])[:40]
# one proof period in steady state = from one quotient evaluation to the next (the bench proves back to back)
qs = [i for i, r in enumerate(rows) if "quotient_kernel" in r["Kernel_Name"]]
start, end = qs[-2], qs[-1]
rows = rows[:end]
t0 = int(rows[start]["Start_Timestamp"]); busy_end = t0; prev = rows[start]; total_gap = 0.0
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s > busy_end:
        g = (s - busy_end) / 1e3
        total_gap += g
        if g >= min_gap: print("%9.1f us  gap %7.1f us  after %-40s before %s" % ((busy_end - t0) / 1e3, g, name(prev), name(r)))
    if e > busy_end: busy_end = e; prev = r
print("proof span %.1f us, idle %.1f us" % ((busy_end - t0) / 1e3, total_gap))
