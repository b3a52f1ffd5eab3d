#!/usr/bin/env python3
"""Does a short idle gap before prove() cost time?  G1ExpStark(128), resident trace: prove() back to back, then with the host sleeping
G milliseconds (busy wait) between proofs.  The instance-list path (generate_trace + prove) shows prove() 0.6-0.8 ms slower than the
resident loop even when no witness kernel runs (profiles/r4_skip_15.json); this isolates the gap itself.
usage (GPU box): idle_gap_prove_time.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import starky_bn254_amd as S
import bench
stark = S.G1ExpStark(128); cfg = stark.config()
ios = bench.synthetic_ios(128, 1000, "g1")
p = S.Prover(stark, cfg, 16)
p.generate_trace(ios)
for _ in range(5): p.prove()
def spin(ms):
    t = time.perf_counter() + ms * 1e-3
    while time.perf_counter() < t: pass
for gap in (0, 0.2, 0.5, 1, 2, 5, 20, 0):
    acc = 0.0; st = {}
    for _ in range(12):
        spin(gap)
        t0 = time.perf_counter(); p.prove(); acc += time.perf_counter() - t0
        for k, v in p.stage_times().items(): st[k] = st.get(k, 0.0) + v / 12
    print("idle gap %5.1f ms: prove %.3f ms  trace_commit %.3f  sponge launches %.3f + %.3f" % (gap, acc / 12 * 1e3, st["trace_commit"], st["trace_absorb_kernels_ms"], st["z_absorb_kernels_ms"]), flush=True)
p.close()
