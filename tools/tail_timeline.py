#!/usr/bin/env python3
"""Timeline of the last proof's tail (openings -> queries) from a rocprofv3 kernel trace directory:
usage: tail_timeline.py <dir with *_kernel_trace.csv>   (prints start offset, duration, gap before, kernel)"""
import csv, glob, sys, re
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
last_open = max(i for i, r in enumerate(rows) if "ext_pow_table_kernel" in r["Kernel_Name"])
first = max(i for i, r in enumerate(rows[:last_open]) if "quotient_kernel" in r["Kernel_Name"])
t0 = int(rows[first]["Start_Timestamp"]); prev_end = t0
for r in rows[first:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = re.sub(r"\(.*", "", r["Kernel_Name"])[:44]
    print("%9.1f us  dur %8.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, name))
    prev_end = max(prev_end, e)
