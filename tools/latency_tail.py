#!/usr/bin/env python3
"""Per-proof wall-clock distribution of G1ExpStark(128) prove() on a GPU box: median, percentiles, the slowest proofs and the
stage that grew in them (host noise shows in the host-bound stages).  usage: python tools/latency_tail.py [proofs=300]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bn254_amd as S
import oracle_lib as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
S.lib().sbn_set_device(0)
ios, _ = O.g1exp_inputs(128, 1)
stark = S.G1ExpStark(128); cfg = stark.config()
p = S.Prover(stark, cfg, 16)
p.generate_trace(ios)
for _ in range(5):
    p.prove()
ts, stages = [], []
for _ in range(n):
    t0 = time.perf_counter(); p.prove(); ts.append((time.perf_counter() - t0) * 1e3); stages.append(p.stage_times())
ts = np.array(ts)
med = {k: float(np.median([s[k] for s in stages])) for k in stages[0]}
print("proofs %d: min %.2f  median %.2f  mean %.2f  p90 %.2f  p99 %.2f  max %.2f ms" % (n, ts.min(), np.median(ts), ts.mean(), np.percentile(ts, 90), np.percentile(ts, 99), ts.max()))
for i in np.argsort(ts)[-5:][::-1]:
    grew = sorted(((stages[i][k] - med[k], k) for k in med if not k.endswith("launches")), reverse=True)[:2]
    print("  proof %3d: %.2f ms; grew: %s" % (i, ts[i], ", ".join("%s +%.2f" % (k, d) for d, k in grew)))
p.close()
