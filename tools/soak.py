#!/usr/bin/env python3
"""Determinism soak on a GPU box: the same trace proved many times must give the same words every time (a race between the
streams of the commit pipeline, or between the ranks of a split proof, would show as a rare differing proof).
usage: soak.py [single_proofs=300] [split_world=8] [split_proofs=20]"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import starky_bn254_amd as S
import oracle_lib as O
from starky_bn254_amd import split

n_single = int(sys.argv[1]) if len(sys.argv) > 1 else 300
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n_split = int(sys.argv[3]) if len(sys.argv) > 3 else 20
S.lib().sbn_set_device(0)
ios, _ = O.g1exp_inputs(128, 1)
stark = S.G1ExpStark(128); cfg = stark.config()
p = S.Prover(stark, cfg, 16)
p.generate_trace(ios)
first = p.prove().words.copy()
t0 = time.time(); bad = 0
for i in range(n_single):
    if not np.array_equal(p.prove().words, first): bad += 1
print(f"single GPU: {n_single} proofs in {time.time() - t0:.1f} s, {bad} differ from the first", flush=True)
p.close()
golden = hashlib.sha256(first.astype('<u8').tobytes()).hexdigest()
proofs, _ = split.prove_local(stark, cfg, 16, world, ios=ios, proofs=n_split)
ok = all(hashlib.sha256(pr.words.astype('<u8').tobytes()).hexdigest() == golden for pr in proofs)
print(f"split over {world} local ranks: {n_split} proofs per rank, last proofs equal the single-GPU proof: {ok}", flush=True)
sys.exit(1 if (bad or not ok) else 0)
