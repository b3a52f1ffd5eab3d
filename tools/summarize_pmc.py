#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection.csv files into
profiles/<tag>_pmc_summary.json: per kernel, launches, average counter value (KB) and bytes.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE tallies 128-byte
requests at 64 bytes (MI355X_MICROARCH.md, section HBM), so the read side is doubled; WRITE_SIZE is exact.
The factor is the guide's figure for wide coalesced streams; our kernels read 8 bytes per lane, where it
is uncalibrated -- the NTT passes (known write bytes, measured exactly) give read factors of 1.6-2.
An optional third file (a --pmc SQ_INSTS_VALU pass) adds the dynamic VALU wave-instruction count per launch, which is
what bounds the Poseidon kernels (one wave64 VALU instruction occupies its SIMD for 4 cycles).
usage: summarize_pmc.py <fetch_csv> <write_csv> <out_json> [<valu_csv>]
"""
import collections, csv, json, sys


def load(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    fa = sum(f.get(k, [0])) / max(len(f.get(k, [])), 1)
    wa = sum(w.get(k, [0])) / max(len(w.get(k, [])), 1)
    out[k] = {"launches_fetch_pass": len(f.get(k, [])), "launches_write_pass": len(w.get(k, [])),
              "FETCH_SIZE_avg_KB": fa, "WRITE_SIZE_avg_KB": wa,
              "hbm_bytes_per_launch_corrected": (2 * fa + wa) * 1024, "hbm_bytes_per_launch_raw": (fa + wa) * 1024}
if len(sys.argv) > 4:
    v = load(sys.argv[4], "SQ_INSTS_VALU")
    for k, vals in v.items():
        out.setdefault(k, {})["valu_wave_instructions_per_launch"] = sum(vals) / len(vals)
        out[k]["launches_valu_pass"] = len(vals)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print("wrote", sys.argv[3])
