#!/usr/bin/env python3
"""Differential soak of the device witness generation on a GPU box: for many seeded instance lists the trace the device kernels
write (flags, curve chains in the placement the prover picked, gadget witness, u16 range check in its prefix-count form) must equal,
word for word, the trace of the product's host generator (one lane-free C++ pass, lookup.rs's merge done literally), and the public
inputs must agree.  Every fourth list replaces the exponents by extreme patterns (zero, all ones, one bit, alternating) so that
the range-checked columns are dominated by a few values.  Both sides are product code written independently of each other; the
oracle is only used for its seeded input generators.
usage: soak_witness.py [table=g1|g2|fq12|fq] [num_io=128] [first_seed=1000] [count=40]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bn254_amd as S
import oracle_lib as O

table = sys.argv[1] if len(sys.argv) > 1 else "g1"
num_io = int(sys.argv[2]) if len(sys.argv) > 2 else 128
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
count = int(sys.argv[4]) if len(sys.argv) > 4 else 40
mk, gen, e0, rows_per = {"g1": (S.G1ExpStark, O.g1exp_inputs, 32, 512), "g2": (S.G2ExpStark, O.g2exp_inputs, 64, 512),
                         "fq12": (S.Fq12ExpStark, O.fq12exp_inputs, 192, 512), "fq": (S.FqExpStark, O.fqexp_inputs, 16, 512)}[table]
stark = mk(num_io); cfg = stark.config()
bits = (rows_per * num_io).bit_length() - 1
prover = S.Prover(stark, cfg, bits)
print("switches:", prover.describe(), flush=True)
PATTERNS = [np.zeros(8, np.uint32), np.full(8, 0xFFFFFFFF, np.uint32), np.array([1, 0, 0, 0, 0, 0, 0, 0], np.uint32),
            np.array([0, 0, 0, 0, 0, 0, 0, 0x80000000], np.uint32), np.full(8, 0xAAAAAAAA, np.uint32), np.full(8, 0x55555555, np.uint32)]
t0 = time.time(); bad = 0
for i in range(count):
    seed = seed0 + i
    ios, _ = gen(num_io, seed)
    if i % 4 == 3:
        ios = ios.copy()
        rng = np.random.default_rng(seed)
        for k in range(num_io):
            if rng.integers(0, 4) != 0:
                pat = PATTERNS[int(rng.integers(0, len(PATTERNS)))].copy()
                if table == "fq12": pat[7] &= 0x0FFFFFFF          # exponent < r
                ios[k, e0:e0 + 8] = pat
    want_trace, want_pi = stark.generate_trace_and_public_inputs(ios)
    pi = prover.generate_trace(ios)
    got = prover.read_trace()
    rows = np.nonzero((got != np.asarray(want_trace)).any(axis=1))[0]
    same = rows.size == 0 and np.array_equal(np.asarray(pi), np.asarray(want_pi))
    if not same:
        bad += 1
        print("seed %d%s: DIFFERS, first differing columns %s" % (seed, " (pattern exponents)" if i % 4 == 3 else "", rows[:8].tolist()), flush=True)
    if i % 4 == 3 and same:                                      # one proof per pattern list, checked by the product's verifier
        S.verify_stark_proof(stark, prover.prove(), cfg)
    if i % 10 == 9: print("  %d lists, %.0f s" % (i + 1, time.time() - t0), flush=True)
prover.close()
print("%s num_io %d: %d instance lists (seeds %d..%d), %d differ from the host generator; %.0f s" % (table, num_io, count, seed0, seed0 + count - 1, bad, time.time() - t0))
sys.exit(1 if bad else 0)
