#!/usr/bin/env python3
"""Trace-commit timeline of the last proof in a rocprofv3 kernel trace (csv): the sponge launches and the gaps between them on the
hash stream, the transform passes on the main stream (duration of each of the four passes of a chunk: alone for the first chunk,
beside a sponge launch for the others).  usage: commit_timeline.py <kernel_trace.csv>"""
import csv, re, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")[:28]
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
absorbs = [i for i, r in enumerate(rows) if "leaf_absorb" in r["Kernel_Name"]]
per_proof = 39 if len(absorbs) % 39 == 0 else None
if per_proof is None: sys.exit("expected 39 sponge launches per proof (G1ExpStark(128): 27 trace + 12 Z)")
last = absorbs[-per_proof:][:27]
t_first = int(rows[last[0]]["Start_Timestamp"])
i = last[0]
while i > 0 and int(rows[i]["Start_Timestamp"]) > t_first - 600000: i -= 1
t0 = int(rows[i + 1]["Start_Timestamp"])
gaps = [(int(rows[b]["Start_Timestamp"]) - int(rows[a]["End_Timestamp"])) / 1e3 for a, b in zip(last, last[1:])]
tend = int(rows[last[-1]]["End_Timestamp"])
ntt = [r for r in rows[i + 1:] if "ntt_fast" in r["Kernel_Name"] and t0 <= int(r["Start_Timestamp"]) < tend]
print("trace commit of the last proof: %d sponge launches, %.1f us each on average, first one starts %.0f us after the stage" %
      (len(last), sum(dur(rows[k]) for k in last) / len(last), (t_first - t0) / 1e3))
print("gaps between sponge launches (hash stream waits for the next chunk's transforms): mean %.1f us, sum %.0f us" % (sum(gaps) / len(gaps), sum(gaps)))
print("transform passes in the stage: %d, sum of durations %.2f ms, idle between them %.0f us; last pass ends at %.2f ms, last sponge launch at %.2f ms" %
      (len(ntt), sum(dur(r) for r in ntt) / 1e3, sum((int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(ntt, ntt[1:])),
       (int(ntt[-1]["End_Timestamp"]) - t0) / 1e6, (tend - t0) / 1e6))
print("passes of chunk 0 (nothing else on the GPU): " + ", ".join("%s %.0f us" % (name(r)[-18:], dur(r)) for r in ntt[:4]))
for c in (1, 2, 13):
    print("passes of chunk %d (beside sponge launch %d): " % (c, c - 1) + ", ".join("%s %.0f us" % (name(r)[-18:], dur(r)) for r in ntt[4 * c:4 * c + 4]))
