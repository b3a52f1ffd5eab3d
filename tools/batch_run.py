#!/usr/bin/env python3
"""Three provers in flight on one GPU, each proving `count` proofs either from a resident trace or from its instance list
(device witness inside the loop): the workload behind bench.py's batch_mode, alone in a process so that a kernel trace of it
can be cut cleanly (tools/batch_timeline.py).  usage: batch_run.py resident|lists [count=12] [provers=3]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import starky_bn254_amd as S
import bench
mode = sys.argv[1]
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
nprov = int(sys.argv[3]) if len(sys.argv) > 3 else 3
stark = S.G1ExpStark(128); cfg = stark.config()
ios = [bench.synthetic_ios(128, 1000 + i, "g1") for i in range(nprov)]
ps = [S.Prover(stark, cfg, 16) for _ in range(nprov)]
for p, io in zip(ps, ios): p.generate_trace(io); p.prove()
def run(p, io):
    for _ in range(count):
        if mode == "lists": p.generate_trace(io)
        p.prove()
ths = [threading.Thread(target=run, args=(p, io)) for p, io in zip(ps, ios)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
dt = time.perf_counter() - t0
print("%s: %d provers x %d proofs in %.3f s = %.2f proofs/s (%.2f ms per proof)" % (mode, nprov, count, dt, nprov * count / dt, dt / (nprov * count) * 1e3), flush=True)
for p in ps: p.close()
