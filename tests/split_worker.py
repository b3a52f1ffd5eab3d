#!/usr/bin/env python3
"""One rank of a split proof (test infrastructure; started by tests/test_split_gpu.py).

usage: split_worker.py <g1|fq12> <num_io> <seed> <outdir> <staged 0|1>      env: RANK WORLD_SIZE MASTER_ADDR MASTER_PORT
Every rank generates the witness on its device, proves its share and writes the proof words (rank<r>.npy) and its stage
times (rank<r>.json).  staged=1: gloo + host-staged exchange, all ranks on device 0 (a one-GPU box); staged=0: RCCL, rank r
on device r."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import oracle_lib as O  # noqa: E402  (seeded inputs only)
import starky_bn254_amd as S  # noqa: E402
from starky_bn254_amd.split import SplitProver  # noqa: E402


def main():
    table, num_io, seed, outdir, staged = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5] == "1"
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dev = 0 if staged else int(os.environ.get("LOCAL_RANK", rank))
    torch.cuda.set_device(dev)
    S.lib().sbn_set_device(dev)
    import datetime
    dist.init_process_group("gloo" if staged else "nccl", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
    if table == "g1":
        ios, _ = O.g1exp_inputs(num_io, seed)
        stark = S.G1ExpStark(num_io)
    else:
        ios, _ = O.fq12exp_inputs(num_io, seed)
        stark = S.Fq12ExpStark(num_io)
    cfg = stark.config()
    bits = (512 * num_io).bit_length() - 1
    sp = SplitProver(stark, cfg, bits, staged=staged)
    pi = sp.generate_trace(ios)
    dist.barrier()
    t0 = time.time()
    proof = sp.prove()
    t1 = time.time()
    proof2 = sp.prove()
    t2 = time.time()
    assert np.array_equal(proof.words, proof2.words)
    np.save(os.path.join(outdir, f"rank{rank}.npy"), proof.words)
    with open(os.path.join(outdir, f"rank{rank}.json"), "w") as f:
        json.dump({"rank": rank, "world": world, "first_prove_s": t1 - t0, "second_prove_s": t2 - t1, "stage_ms": sp.stage_times(),
                   "exchange_bytes_sent_per_proof": sp.comm.bytes_sent // 2, "public_inputs": int(len(pi))}, f)
    sp.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
