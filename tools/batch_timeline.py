#!/usr/bin/env python3
"""Where the GPU's time goes with several proofs in flight, from a rocprofv3 kernel trace of tools/batch_run.py: over the steady part
of the run (first to last leaf_absorb_kernel, warm-up proofs cut off) the share of wall time with no kernel at all, with no sponge
launch running, with one / two / three sponge launches running at once, and the kernel time by family.
usage: batch_timeline.py <dir with *_kernel_trace.csv> [skip_first_fraction=0.25]"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.25
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
absorb = [r for r in rows if "leaf_absorb" in r["Kernel_Name"]]
t_lo = int(absorb[int(len(absorb) * skip)]["Start_Timestamp"]); t_hi = int(absorb[-1]["End_Timestamp"])
ev = []
fam = collections.Counter()
def family(n):
    if "leaf_absorb" in n: return "sponge"
    if "ntt_" in n: return "transforms"
    if "tg::" in n: return "witness"
    if "quotient" in n: return "quotient"
    if "merkle" in n or "fri_leaf" in n: return "merkle / fri hashing"
    if "permz" in n or "permutation_z" in n: return "permutation z"
    if "openings" in n: return "openings"
    return "other"
for r in rows:
    s, e = max(int(r["Start_Timestamp"]), t_lo), min(int(r["End_Timestamp"]), t_hi)
    if e <= s: continue
    k = family(r["Kernel_Name"])
    fam[k] += e - s
    ev.append((s, 1, k == "sponge")); ev.append((e, -1, k == "sponge"))
ev.sort()
any_n = sp_n = 0; last = t_lo
idle = nosponge = 0; by_sp = collections.Counter()
for t, d, is_sp in ev:
    dt = t - last
    if any_n == 0: idle += dt
    if sp_n == 0: nosponge += dt
    by_sp[min(sp_n, 3)] += dt
    last = t
    any_n += d
    if is_sp: sp_n += d
wall = t_hi - t_lo
print("steady window %.1f ms: no kernel at all %.1f %%, no sponge launch running %.1f %%; sponge launches running at once: 0: %.1f %%  1: %.1f %%  2: %.1f %%  3+: %.1f %%" %
      (wall / 1e6, 100 * idle / wall, 100 * nosponge / wall, *(100 * by_sp[k] / wall for k in range(4))))
print("kernel time by family, as a share of the window (kernels overlap, so the shares add up to more than 100 %):")
for k, v in fam.most_common(): print("   %-22s %6.1f %%" % (k, 100 * v / wall))
