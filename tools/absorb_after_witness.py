#!/usr/bin/env python3
"""Are the sponge launches of a proof slower right after device witness generation, and which ones?  Reads a rocprofv3 --kernel-trace
CSV of `bench.py --steps K --no-batch-mode` (K resident proofs, then 1 + 5 generate_trace + prove pairs), groups the
leaf_absorb_kernel launches by proof (39 per G1 proof) and prints the mean duration of launch 0..38 for proofs that follow a
tg::range_check_kernel launch against proofs that do not.
usage: absorb_after_witness.py <dir with *_kernel_trace.csv>"""
import csv, glob, os, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
proofs, cur, after_witness, flag = [], [], [], False
for s, e, name in rows:
    if name.startswith("void tg::range_check_kernel") or name.startswith("tg::range_check_kernel"):
        flag = True
    if name.startswith("leaf_absorb_kernel"):
        cur.append((e - s) / 1e3)
        if len(cur) == 39:
            proofs.append(cur); after_witness.append(flag); cur = []; flag = False
res = [p for p, a in zip(proofs, after_witness) if not a][1:]      # (the first proof of the process is a warm-up)
aft = [p for p, a in zip(proofs, after_witness) if a][1:]
print(f"# {len(res)} resident proofs, {len(aft)} proofs right after generate_trace; microseconds per leaf_absorb_kernel launch (27 trace chunks, then 12 Z chunks)")
for k in range(39):
    a = sum(p[k] for p in res) / max(len(res), 1); b = sum(p[k] for p in aft) / max(len(aft), 1)
    print(f"launch {k:2d}  resident {a:8.1f}  after witness {b:8.1f}  ratio {b / a:6.3f}")
print("sum      resident %8.1f  after witness %8.1f" % (sum(sum(p) for p in res) / max(len(res), 1), sum(sum(p) for p in aft) / max(len(aft), 1)))
