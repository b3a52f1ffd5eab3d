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
        assert info[r]["stage_ms"]["split_exchange_ms"] > 0        # world 1: the quotient / FRI gathers still pass through the transport
    stark = S.G1ExpStark(num_io) if table == "g1" else S.Fq12ExpStark(num_io)
    S.verify_stark_proof(stark, S.Proof(words[0], (512 * num_io).bit_length() - 1), stark.config())
    print(f"{table} world {world}: second prove {[round(i['second_prove_s'], 3) for i in info]} s, "
          f"exchange {[round(i['stage_ms']['split_exchange_ms'], 1) for i in info]} ms, "
          f"sent per rank {[i['exchange_bytes_sent_per_proof'] >> 20 for i in info]} MiB")


def _digest(words):
    return hashlib.sha256(np.asarray(words).astype("<u8").tobytes()).hexdigest()


@pytest.mark.parametrize("table,num_io,seed,key,world", [
    ("fq12", 16, 3, "fq12exp_io16_seed3", 8),      # BASELINE config[4]'s world size: two cap subtrees per rank, two planes
    ("g1", 128, 1, "g1exp_io128_seed1", 8),
    ("fq12", 16, 3, "fq12exp_io16_seed3", 16),     # the largest world sbn.h admits: one cap subtree per rank
    ("g1", 128, 1, "g1exp_io128_seed1", 16),
    ("g1", 128, 1, "g1exp_io128_seed1", 2),        # (the local transport at the sizes the multi-process test covers too)
])
def test_split_proof_local_ranks(S, O, golden, table, num_io, seed, key, world):
    """8 and 16 ranks -- config[4]'s world size and the maximum -- as threads of ONE process on device 0 (sbn_local_comm_create;
    the box allows at most 6 GPU processes, and RCCL one rank per device): every rank's proof is word for word the
    single-GPU proof = the committed digest of the oracle's proof; the exchanges are real device-to-device copies ordered by
    events, with the pipeline (transform k+1 | exchange k | sponge k-1) running on three streams per rank."""
    from starky_bn254_amd import split
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    S.lib().sbn_set_device(0)
    ios, _ = (O.g1exp_inputs if table == "g1" else O.fq12exp_inputs)(num_io, seed)
    stark = S.G1ExpStark(num_io) if table == "g1" else S.Fq12ExpStark(num_io)
    bits = (512 * num_io).bit_length() - 1
    proofs, times = split.prove_local(stark, stark.config(), bits, world, ios=ios, proofs=2)
    g = golden["proof_digests"][key]
    for r in range(world):
        assert len(proofs[r].words) == g["proof_words"]
        assert _digest(proofs[r].words) == g["proof_sha256"], f"rank {r} of {world}"
        assert times[r]["split_exchange_ms"] > 0
    S.verify_stark_proof(stark, proofs[0], stark.config())
    print(f"{table} world {world} (local ranks): exchange {[round(t['split_exchange_ms'], 1) for t in times]} ms")


def test_transport_selftests(S):
    """sbn_comm_selftest through both native transports: uneven blocks, self blocks, the all-gather form, the host all-gather.
    RCCL at world 1 (one rank per device is all a one-GPU box allows): ncclSend / ncclRecv to itself inside a group, and
    ncclAllGather; local at world 8 (the node's size; bench.py --split runs the same self-test before its first proof at
    world > 1).  A transport that delivers a wrong block fails FAST, naming the receiving rank and the block: rank 2's exchange
    is wrapped so that it posts the blocks for ranks 0 and 1 swapped."""
    import ctypes as C
    import threading
    from starky_bn254_amd import split
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    S.lib().sbn_set_device(0)
    rc = split.RcclComm(1 << 22, 1 << 22, 0, 1)
    rc.selftest()
    rc.close()
    world = 8
    for corrupt in (False, True):
        grp = split.LocalGroup(world, 1 << 23, 1 << 23)
        comms = [grp.comms[r] for r in range(world)]
        keep = []
        if corrupt:
            orig = grp.comms[2].all_to_all

            def swapped(ctx, stream, so, sl, ro, rl):
                arr = (C.c_uint64 * world)(*[so[i] for i in range(world)])
                arr[0], arr[1] = arr[1], arr[0]
                return orig(ctx, stream, arr, sl, ro, rl)
            cb = split._A2A(swapped)
            keep.append(cb)
            bad = split._Comm.from_buffer_copy(grp.comms[2])
            bad.all_to_all = cb
            comms[2] = bad
        res, msg = [None] * world, [None] * world

        def go(r):
            res[r] = S.lib().sbn_comm_selftest(C.byref(comms[r]))
            if res[r] != 0:
                msg[r] = S.lib().sbn_last_error().decode()
                grp.abort()
        th = [threading.Thread(target=go, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
        grp.close()
        if not corrupt:
            assert res == [0] * world, msg
        else:
            assert res[0] != 0 and res[1] != 0, (res, msg)
            assert "on rank 0 the block from rank 2" in msg[0] and "on rank 1 the block from rank 2" in msg[1], msg


def test_split_proof_over_the_native_rccl_transport_world_1(S, O, golden):
    """sbn_rccl_comm_create + sbn_split_prover_* with no torch.distributed anywhere: the gathers of the quotient values and
    of the FRI partial sums run as grouped ncclSend / ncclRecv on the prover's stream, the caps / openings / query rows
    through ncclAllGather.  World 1 is what a one-GPU box can run; the proof must be the committed one."""
    from starky_bn254_amd import split
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    S.lib().sbn_set_device(0)
    ios, _ = O.fq12exp_inputs(16, 3)
    stark = S.Fq12ExpStark(16)
    cfg = stark.config()
    sb, rb = split.exchange_bytes(stark, cfg, 13, 1)
    comm = split.RcclComm(sb, rb, 0, 1)
    sp = split.SplitProver(stark, cfg, 13, transport=comm)
    sp.generate_trace(ios)
    proof = sp.prove()
    t = sp.stage_times()
    sp.close()
    comm.close()
    g = golden["proof_digests"]["fq12exp_io16_seed3"]
    assert _digest(proof.words) == g["proof_sha256"]
    assert t["split_exchange_ms"] > 0


def test_split_proof_at_config4_size_two_local_ranks(S, O, golden):
    """BASELINE config[4]'s table at full size -- Fq12ExpStark(512): 2^18 rows x 11,786 columns, 185 + 84 column blocks, send
    slots of 0.27 GB, a 36 GB row matrix per rank -- split over two ranks (threads, one GPU; four and more ranks of this size
    do not fit one card's HBM beside each other): the offsets and counts that only get large here (block * 64 * M words, the
    FRI group weights of 93 own blocks, the 2^19-point split first LDE pass) must give, on both ranks, word for word the
    single-GPU proof of the same instances, which the product verifier accepts."""
    from starky_bn254_amd import split
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    S.lib().sbn_set_device(0)
    ios, _ = O.fq12exp_inputs(512, 3)
    stark = S.Fq12ExpStark(512)
    cfg = stark.config()
    proofs, times = split.prove_local(stark, cfg, 18, 2, ios=ios, proofs=1)
    assert np.array_equal(proofs[0].words, proofs[1].words)
    split_words = proofs[0].words.copy()
    del proofs
    single = S.Prover(stark, cfg, 18)
    try:
        single.generate_trace(ios)
        ref = single.prove()
    finally:
        single.close()
    assert np.array_equal(split_words, ref.words)
    # ... and the CPU oracle PROVER's proof of the same instances (committed digest, tools/oracle_digest.py)
    g = golden["proof_digests"]["fq12exp_io512_seed3"]
    assert len(split_words) == g["proof_words"]
    assert hashlib.sha256(np.ascontiguousarray(split_words, dtype="<u8").tobytes()).hexdigest() == g["proof_sha256"]
    S.verify_stark_proof(stark, ref, cfg)
    print(f"config[4] size, 2 local ranks: exchange {[round(t['split_exchange_ms'], 1) for t in times]} ms")


def test_a_rank_without_a_witness_stops_all_ranks(S, O):
    """a rank that failed before prove() (here: it never loaded a trace) must not leave the others waiting in
    the first exchange.  sbn_split_prover_prove agrees on a status word over the host all-gather first: EVERY rank returns an
    error, nobody hangs; after the missing witness is supplied the same provers produce the proof."""
    import threading
    import time
    from starky_bn254_amd import split
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    S.lib().sbn_set_device(0)
    ios, _ = O.fq12exp_inputs(16, 3)
    stark = S.Fq12ExpStark(16)
    cfg = stark.config()
    sb, rb = split.exchange_bytes(stark, cfg, 13, 2)
    grp = split.LocalGroup(2, sb, rb)
    provers = [split.SplitProver(stark, cfg, 13, transport=grp.comms[r]) for r in range(2)]
    provers[0].generate_trace(ios)
    res = [None, None]

    def go(r):
        try:
            res[r] = provers[r].prove()
        except Exception as e:  # noqa: BLE001
            res[r] = e
    t0 = time.time()
    th = [threading.Thread(target=go, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=120) for t in th]
    assert not any(t.is_alive() for t in th), "a rank hangs"
    assert all(isinstance(x, S.SbnError) for x in res), res
    assert "rank 1 has no trace" in str(res[0]) and time.time() - t0 < 60
    provers[1].generate_trace(ios)
    th = [threading.Thread(target=go, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert all(isinstance(x, S.Proof) for x in res), res
    assert np.array_equal(res[0].words, res[1].words)
    for p in provers:
        p.close()
    grp.close()
