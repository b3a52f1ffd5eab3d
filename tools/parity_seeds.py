#!/usr/bin/env python3
"""One-off parity sweep on a GPU box: G1ExpStark(128) (or, with a leading `g2`, G2ExpStark(128)) proofs for extra seeds, GPU (device
witness) vs the CPU oracle, word for word.
usage: parity_seeds.py [g2] 11 12 ...   (test infrastructure: imports the oracle through tests/oracle_lib.py)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bn254_amd as S
import oracle_lib as O

g2 = len(sys.argv) > 1 and sys.argv[1] == "g2"
stark = S.G2ExpStark(128) if g2 else S.G1ExpStark(128); cfg = stark.config()
prover = S.Prover(stark, cfg, 16)
print("switches:", prover.describe(), flush=True)
for seed in [int(x) for x in sys.argv[2 if g2 else 1:]]:
    ios, native = (O.g2exp_inputs if g2 else O.g1exp_inputs)(128, seed)
    trace, pi = (O.g2exp_trace if g2 else O.g1exp_trace)(ios)
    t = time.time(); want, _ = O.prove(O.AIR_G2_EXP if g2 else O.AIR_G1_EXP, 128, trace, pi); t_cpu = time.time() - t
    pi_dev = prover.generate_trace(ios)
    t = time.time(); got = prover.prove(); t_gpu = time.time() - t
    same = np.array_equal(np.asarray(got.words), np.asarray(want)) and np.array_equal(np.asarray(pi_dev), np.asarray(pi))
    S.verify_stark_proof(stark, got, cfg)
    print("seed %d: identical=%s  words=%d  cpu %.1f s  gpu %.1f ms" % (seed, same, len(got.words), t_cpu, t_gpu * 1e3), flush=True)
    if not same: sys.exit(1)
print("parity sweep ok")
