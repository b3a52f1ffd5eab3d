#!/usr/bin/env python3
"""Runs one large table end to end on the GPU: host witness -> H2D -> prove -> product verify.
usage: run_large.py fq12 <num_io> | fq <num_io> | g2 <num_io> | g1 <num_io>     (prints one JSON line)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import starky_bn254_amd as S

P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
table, num_io = sys.argv[1], int(sys.argv[2])
rng = np.random.default_rng(7)
t0 = time.time()
if table == "fq12":
    stark = S.Fq12ExpStark(num_io)
    ios = np.zeros((num_io, 200), dtype=np.uint32)
    for k in range(num_io):
        for c in range(24):
            v = int.from_bytes(rng.bytes(32), "little") % P
            ios[k, 8 * c:8 * c + 8] = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        e = int.from_bytes(rng.bytes(32), "little") % R
        ios[k, 192:200] = [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
elif table == "fq":
    stark = S.FqExpStark(num_io)
    ios = np.zeros((num_io, 24), dtype=np.uint32)
    for k in range(num_io):
        for c in range(2):
            v = int.from_bytes(rng.bytes(32), "little") % P
            ios[k, 8 * c:8 * c + 8] = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        ios[k, 16:24] = np.frombuffer(rng.bytes(32), dtype=np.uint32)
else:
    sys.path.insert(0, ROOT)
    import bench
    stark = S.G1ExpStark(num_io) if table == "g1" else S.G2ExpStark(num_io)
    ios = bench.synthetic_ios(num_io, 7, table)
cfg = stark.config()
degree_bits = (512 * num_io).bit_length() - 1
t_in = time.time() - t0
t0 = time.time(); trace, pi = stark.generate_trace_and_public_inputs(ios); t_gen = time.time() - t0
print(f"# trace {trace.shape} {trace.nbytes/1e9:.2f} GB generated in {t_gen:.1f}s", file=sys.stderr, flush=True)
t0 = time.time(); prover = S.Prover(stark, cfg, degree_bits); t_create = time.time() - t0
t0 = time.time(); prover.load_trace(trace, pi); t_load = time.time() - t0
print(f"# prover created {t_create:.1f}s, trace loaded {t_load:.1f}s", file=sys.stderr, flush=True)
times = []
for it in range(3):
    t0 = time.time(); proof = prover.prove(); times.append(time.time() - t0)
    print(f"# prove {it}: {times[-1]*1e3:.1f} ms", file=sys.stderr, flush=True)
stages = prover.stage_times()
t0 = time.time(); S.verify_stark_proof(stark, proof, cfg); t_ver = time.time() - t0
# the same instances with the witness generated on the device: no host trace, no PCIe transfer of it
dev = {}
try:
    prover.generate_trace(ios)                                   # warm-up (pinned staging buffer, worker pool)
    t0 = time.time(); pi_dev = prover.generate_trace(ios); dev["device_tracegen_wall_ms"] = (time.time() - t0) * 1e3
    dev["device_tracegen_gpu_ms"] = prover.stage_times()["device_tracegen_ms"]
    t0 = time.time(); proof_dev = prover.prove(); dev["prove_ms"] = (time.time() - t0) * 1e3
    dev["same_public_inputs"] = bool((pi_dev == pi).all())
    dev["same_proof"] = bool((proof_dev.words == proof.words).all())
except S.SbnError as e:
    dev["unsupported"] = str(e)
print(json.dumps({"table": type(stark).__name__, "num_io": num_io, "rows": 512 * num_io, "columns": stark.num_columns,
                  "public_inputs": stark.num_public_inputs, "trace_GB": trace.nbytes / 1e9, "tracegen_s": t_gen, "create_s": t_create,
                  "h2d_s": t_load, "prove_ms": [t * 1e3 for t in times], "verify_s": t_ver, "proof_words": int(len(proof.words)),
                  "stage_ms": stages, "device_witness": dev}))
