#!/usr/bin/env python3
"""Diagnostic driver for counter passes over the quotient kernel: proves G1ExpStark(128) (device witness) a few times
WITHOUT verifying, so that SBN_DIAG_QUOTIENT_SEGMASK (run only some of the four constraint segments; the proof is then
invalid; compiled in only with -DSBN_DIAG, see tools/pmc_quotient_segments.sh) can be used under rocprofv3 --pmc.  usage: quotient_diag.py [table g1|g2] [proofs]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import starky_bn254_amd as S

table = sys.argv[1] if len(sys.argv) > 1 else "g1"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
stark = S.G1ExpStark(128) if table == "g1" else S.G2ExpStark(128)
prover = S.Prover(stark, stark.config(), 16)
prover.generate_trace(bench.synthetic_ios(128, 1000, table))
for _ in range(reps):
    prover.prove()
print({k: round(v, 3) for k, v in prover.stage_times().items() if k.startswith("quotient")})
