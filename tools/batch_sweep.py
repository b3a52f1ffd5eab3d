#!/usr/bin/env python3
"""Throughput of the batch prover (sbn_batch_prover_*, instance lists -> proofs) against the number of proofs in flight.
Run from the repo root on a GPU box: python tools/batch_sweep.py
Round 1, G1ExpStark(128): 2 -> 28.5, 3 -> 30.7, 4 -> 30.8, 5 -> 31.1, 6 -> 31.0 proofs/s (resident traces, 3 in flight: 34.6)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import starky_bn254_amd as S, bench
stark = S.G1ExpStark(128); cfg = stark.config()
ios = bench.synthetic_ios(128, 0, "g1")
for inflight in (2, 3, 4, 5, 6):
    bp = S.BatchProver(stark, cfg, 16, inflight)
    units = np.broadcast_to(ios, (6 * inflight,) + ios.shape)
    bp.prove_ios(units[:inflight]); torch.cuda.synchronize()
    t0 = time.perf_counter(); bp.prove_ios(units); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"inflight={inflight}: {len(units)/dt:.2f} proofs/s from instance lists", flush=True)
    bp.close()
