#!/usr/bin/env python3
"""Thread scaling of the CPU oracle's prove() on the GPU box's host cores: where the restated CPU baseline spends its time and
which phases stop scaling (the hash work alone is ~0.1-0.3 s of 256-core time, the commitment took
8.4 s).  Oracle only -- nothing here touches the product.

usage: python tools/oracle_scaling.py [threads ...]      (default: 8 32 64 128 all; G1ExpStark(128), seed 1)
Per thread count: two consecutive proofs (cold allocator, warm allocator), every stage and commit phase in seconds, as JSON
lines on stdout."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    ncpu = os.cpu_count() or 8
    want = [int(a) for a in sys.argv[1:]] or [8, 32, 64, 128, ncpu]
    threads = sorted({t for t in want if t <= ncpu} | ({ncpu} if not sys.argv[1:] else set()))
    ios, _ = O.g1exp_inputs(128, 1)
    t0 = time.time()
    trace, pi = O.g1exp_trace(ios)
    print(json.dumps({"host_cores": ncpu, "trace_generation_s": time.time() - t0, "table": "G1ExpStark(128), 2^16 rows x 1676 columns"}), flush=True)
    digest = None
    for t in threads:
        O.lib().orc_set_threads(t)
        for run in ("cold", "warm"):
            w, secs = O.prove(O.AIR_G1_EXP, 128, trace, pi)
            d = int(np.bitwise_xor.reduce(w))
            assert digest in (None, d), "proof words depend on the thread count"
            digest = d
            print(json.dumps({"threads": t, "run": run, "prove_s": round(secs, 3), "stages_s": {k: round(v, 3) for k, v in O.last_stage_seconds().items()}}), flush=True)


if __name__ == "__main__":
    main()
