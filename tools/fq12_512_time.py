#!/usr/bin/env python3
"""Fq12ExpStark(512) (2^18 rows x 11,786 columns, BASELINE config[4] on ONE GPU): device witness, three timed prove() calls, stage times,
product verifier.  usage (on a GPU box): python tools/fq12_512_time.py"""
import sys, os, time, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import starky_bn254_amd as S, oracle_lib as O
ios, _ = O.fq12exp_inputs(512, 3)
stark = S.Fq12ExpStark(512); cfg = stark.config()
p = S.Prover(stark, cfg, 18)
p.generate_trace(ios)
pr = p.prove()
ts = []
for _ in range(3):
    t0 = time.perf_counter(); pr2 = p.prove(); ts.append(time.perf_counter() - t0)
st = p.stage_times()
import hashlib
print("prove ms", [round(t * 1e3, 1) for t in ts], {k: round(v, 1) for k, v in st.items() if v > 1}, hashlib.sha256(pr2.to_bytes()).hexdigest()[:16])
S.verify_stark_proof(stark, pr2, cfg)
p.close()
