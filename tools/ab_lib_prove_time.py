#!/usr/bin/env python3
"""A/B of two builds of libsbn254.so on the time of prove() alone (no verification: a MEASUREMENT build may compute wrong proofs).
usage: SBN_LIB=<other .so> ab_lib_prove_time.py [steps]   -> one line: ms per G1ExpStark(128) proof and the stage split"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import starky_bn254_amd as S
import bench
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
stark = S.G1ExpStark(128); cfg = stark.config()
ios = bench.synthetic_ios(128, 1000, "g1")
p = S.Prover(stark, cfg, 16)
p.generate_trace(ios)
for _ in range(5):
    p.prove()
acc = {}
t0 = time.perf_counter()
for _ in range(steps):
    p.prove()
    for k, v in p.stage_times().items():
        acc[k] = acc.get(k, 0.0) + v / steps
ms = (time.perf_counter() - t0) / steps * 1e3
print(os.environ.get("SBN_LIB", "in-tree library"), "ms_per_proof %.3f" % ms, {k: round(acc[k], 3) for k in ("trace_commit", "z_commit", "trace_absorb_kernels_ms", "z_absorb_kernels_ms", "quotient_commit", "fri_layers")})
