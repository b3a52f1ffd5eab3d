#!/usr/bin/env python3
"""Determinism soak over the pipeline variants of the commit stage on a GPU box: the same trace proved many times must give the
same words every time -- at 2^16 rows (fused middle pass, one transform stream), 2^17 rows (48-column chunks) and 2^18 rows (the
512-point fused pass, two transform streams with alternating buffers).  A race between the streams would show as a rare differing proof.
usage: soak_sizes.py [scale=1]   (scale multiplies the proof counts)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bn254_amd as S
import oracle_lib as O          # seeded input generators only
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
CASES = [("g1", 128, 200), ("g2", 128, 60), ("g1", 256, 40), ("fq", 512, 30), ("g1", 512, 12), ("fq12", 512, 6)]
mk = {"g1": (S.G1ExpStark, O.g1exp_inputs), "g2": (S.G2ExpStark, O.g2exp_inputs), "fq": (S.FqExpStark, O.fqexp_inputs), "fq12": (S.Fq12ExpStark, O.fq12exp_inputs)}
bad_total = 0
for table, num_io, count in CASES:
    count = max(2, int(count * scale))
    stark = mk[table][0](num_io); cfg = stark.config()
    p = S.Prover(stark, cfg, (512 * num_io).bit_length() - 1)
    p.generate_trace(mk[table][1](num_io, 7)[0])
    first = p.prove()
    S.verify_stark_proof(stark, first, cfg)
    want = hashlib.sha256(first.to_bytes()).hexdigest()
    t0 = time.time(); bad = 0
    for _ in range(count):
        if hashlib.sha256(p.prove().to_bytes()).hexdigest() != want: bad += 1
    d = p.describe()
    print("%s(%d): %d proofs in %.1f s (chunk %s, streams %s, fused %s), %d differ from the first (%s)" % (type(stark).__name__, num_io, count, time.time() - t0, d["ntt_chunk"], d["ntt_streams"], d["ntt_fused"], bad, want[:16]), flush=True)
    bad_total += bad
    p.close()
sys.exit(1 if bad_total else 0)
