#!/usr/bin/env python3
"""prove() time of any Exp table at any supported size on a GPU box (device witness, one warm-up, `steps` timed proofs, product
verifier): the A/B tool for sizes bench.py does not quote.  usage: prove_time.py <g1|g2|fq12|fq> <num_io> [steps=5]"""
import os, sys, time, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import starky_bn254_amd as S
import oracle_lib as O          # seeded input generators only
table, num_io = sys.argv[1], int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
mk, gen = {"g1": (S.G1ExpStark, O.g1exp_inputs), "g2": (S.G2ExpStark, O.g2exp_inputs), "fq12": (S.Fq12ExpStark, O.fq12exp_inputs), "fq": (S.FqExpStark, O.fqexp_inputs)}[table]
stark = mk(num_io); cfg = stark.config()
p = S.Prover(stark, cfg, (512 * num_io).bit_length() - 1)
p.generate_trace(gen(num_io, 3)[0])
pr = p.prove()
ts = []; acc = {}
for _ in range(steps):
    t0 = time.perf_counter(); pr = p.prove(); ts.append(time.perf_counter() - t0)
    for k, v in p.stage_times().items(): acc[k] = acc.get(k, 0.0) + v / steps
S.verify_stark_proof(stark, pr, cfg)
d = p.describe()
print("%s(%d) chunk %s streams %s fused %s: prove ms min %.2f mean %.2f  trace_commit %.2f (sponge launches %.2f in %d) z_commit %.2f (%.2f in %d)  sha256 %s" % (
    type(stark).__name__, num_io, d["ntt_chunk"], d["ntt_streams"], d["ntt_fused"], min(ts) * 1e3, sum(ts) / len(ts) * 1e3, acc["trace_commit"], acc["trace_absorb_kernels_ms"], round(acc["trace_absorb_launches"]), acc["z_commit"], acc["z_absorb_kernels_ms"], round(acc["z_absorb_launches"]),
    hashlib.sha256(pr.to_bytes()).hexdigest()[:16]), flush=True)
p.close()
