"""The oversized-trace split (BASELINE config[4] as written: one trace over several GPUs, sbn_split_prover_*): every rank's
proof must be word for word the single-GPU proof, i.e. have the committed digest of the CPU oracle's proof.  On the
one-GPU test box the ranks share device 0 and exchange through the host-staged backend (gloo); the RCCL backend runs the
same library code with a different transport (bench.py --split on a multi-GPU node)."""
import hashlib
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run_split(table, num_io, seed, world, outdir, staged=True, timeout=600):
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SBN_HOST_THREADS=str(max(1, (os.cpu_count() or 8) // world)))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "split_worker.py"), table, str(num_io), str(seed), str(outdir),
                                       "1" if staged else "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out.decode(errors="replace"))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{outs[r][-3000:]}"
    words = [np.load(os.path.join(outdir, f"rank{r}.npy")) for r in range(world)]
    info = [json.load(open(os.path.join(outdir, f"rank{r}.json"))) for r in range(world)]
    return words, info


@pytest.mark.parametrize("table,num_io,seed,key,world", [
    ("fq12", 16, 3, "fq12exp_io16_seed3", 1),      # the sharded code path with trivial collectives
    ("fq12", 16, 3, "fq12exp_io16_seed3", 2),      # the reference's test_fq12_exp_raw size, two ranks: next rows are local
    ("g1", 128, 1, "g1exp_io128_seed1", 2),        # BASELINE config[1] as one split proof
    ("g1", 128, 1, "g1exp_io128_seed1", 4),        # four ranks: the rows i + 2 arrive as a second plane
    ("fq12", 16, 3, "fq12exp_io16_seed3", 4),
])
def test_split_proof_equals_single_gpu_proof(S, golden, tmp_path, table, num_io, seed, key, world):
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    words, info = run_split(table, num_io, seed, world, tmp_path, staged=True)
    g = golden["proof_digests"][key]
    for r in range(world):
        assert len(words[r]) == g["proof_words"]
        assert hashlib.sha256(words[r].astype("<u8").tobytes()).hexdigest() == g["proof_sha256"], f"rank {r}"
        assert info[r]["stage_ms"]["split_exchange_ms"] > 0
    stark = S.G1ExpStark(num_io) if table == "g1" else S.Fq12ExpStark(num_io)
    S.verify_stark_proof(stark, S.Proof(words[0], (512 * num_io).bit_length() - 1), stark.config())
    print(f"{table} world {world}: second prove {[round(i['second_prove_s'], 3) for i in info]} s, "
          f"exchange {[round(i['stage_ms']['split_exchange_ms'], 1) for i in info]} ms, "
          f"sent per rank {[i['exchange_bytes_sent_per_proof'] >> 20 for i in info]} MiB")
