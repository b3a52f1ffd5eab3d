"""world_size-2 (and 4, 8) `gloo` tests on CPU of the collectives behind the oversized-trace split (starky_bn254_amd/split.py):
TorchComm.all_to_all with uneven blocks and with one block sent to every rank (the library's device all-gather), and
TorchComm.all_gather_host, in the host-staged mode the one-GPU parity test uses.  The sharded proving itself needs a GPU
(tests/test_split_gpu.py)."""
import os
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pattern(src, dst, n):
    return ((np.arange(n, dtype=np.uint64) * 7 + 1000 * src + 31 * dst) % 251).astype(np.uint8)


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from starky_bn254_amd.split import TorchComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # columns -> rows exchange with uneven column shares: rank r sends (r + 1) * 40 bytes to every rank
    mine = (rank + 1) * 40
    send = torch.zeros(world * mine + 64, dtype=torch.uint8)
    for d in range(world):
        send[d * mine:(d + 1) * mine] = torch.from_numpy(_pattern(rank, d, mine))
    offs = [sum((s + 1) * 40 for s in range(r)) for r in range(world)]
    recv = torch.zeros(offs[-1] + world * 40 + 512, dtype=torch.uint8)
    comm = TorchComm(send, recv, staged=True)
    assert comm.all_to_all([(d * mine, mine) for d in range(world)], [(offs[s], (s + 1) * 40) for s in range(world)]) == 0
    ok = all(np.array_equal(recv[offs[s]:offs[s] + (s + 1) * 40].numpy(), _pattern(s, rank, (s + 1) * 40)) for s in range(world))
    # all-gather expressed as an all-to-all: the same block goes to every rank, received behind the matrices
    send[:16] = torch.from_numpy(_pattern(rank, 99, 16))
    base = offs[-1] + world * 40
    assert comm.all_to_all([(0, 16)] * world, [(base + 16 * s, 16) for s in range(world)]) == 0
    ok = ok and all(np.array_equal(recv[base + 16 * s:base + 16 * (s + 1)].numpy(), _pattern(s, 99, 16)) for s in range(world))
    # the last step of a commitment: the highest rank has no column block left (zero-length blocks are skipped on both sides)
    # and nobody sends to itself (the pack kernel writes a rank's own rows straight into its receive matrix)
    recv.zero_()
    has = [r != world - 1 for r in range(world)]
    for d in range(world):
        send[d * 32:(d + 1) * 32] = torch.from_numpy(_pattern(rank, d, 32) + 1)
    sends = [(d * 32, 32 if (has[rank] and d != rank) else 0) for d in range(world)]
    recvs = [(s * 32, 32 if (has[s] and s != rank) else 0) for s in range(world)]
    assert comm.all_to_all(sends, recvs) == 0
    for s_ in range(world):
        got = recv[s_ * 32:(s_ + 1) * 32].numpy()
        want = (_pattern(s_, rank, 32) + 1) if (has[s_] and s_ != rank) else np.zeros(32, dtype=np.uint8)
        ok = ok and np.array_equal(got, want)
    sent_last = 32 * (world - 1) if has[rank] else 0
    g = comm.all_gather_host(_pattern(rank, 7, 24))
    ok = ok and g.shape == (world, 24) and all(np.array_equal(g[s], _pattern(s, 7, 24)) for s in range(world))
    ret[rank] = bool(ok) and comm.bytes_sent == (world - 1) * (mine + 16) + sent_last
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_split_collectives_on_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert [ret[r] for r in range(world)] == [True] * world


def test_split_row_ownership_model():
    """The index algebra the split kernels implement (kernels.cuh split_pack_kernel, prover.hip commit_split), restated with
    numpy: rank s owns the LDE rows j * R + rho(s) (rho = bit reversal on log R bits), which are exactly the leaves
    [s * m/R, (s+1) * m/R) of the bit-reversed Merkle tree, local leaf = bitrev(j); the next-row plane of rank s holds the
    rows i + 2."""
    def brev(x, bits):
        return int(format(x, "0%db" % bits)[::-1], 2) if bits else 0
    log_m = 7
    m = 1 << log_m
    for log_r in (0, 1, 2, 3, 4):
        R = 1 << log_r
        ml = m >> log_r
        owner_of_leaf = {}
        for s in range(R):
            rho = brev(s, log_r)
            for j in range(ml):
                i = j * R + rho
                leaf = brev(i, log_m)
                assert leaf >> (log_m - log_r) == s                      # complete cap subtrees
                assert leaf & (ml - 1) == brev(j, log_m - log_r)          # local leaf index = bitrev(local row)
                owner_of_leaf[leaf] = s
                # next row (i + 2): local row j + 1 with two ranks, local row j of the other plane from four ranks up
                nxt = (i + 2) % m
                if R == 1:
                    assert nxt == (j + 2) % ml
                elif R == 2:
                    assert nxt == ((j + 1) % ml) * R + rho
                else:
                    p = (nxt - 2) % m                                      # split_pack_kernel: row nxt is the next row of point p
                    assert brev(p % R, log_r) == s and p >> log_r == j
        assert sorted(owner_of_leaf) == list(range(m))


def test_split_column_ownership_model():
    """The column side of the split (prover.hip ColShare): columns are dealt in blocks of 64, round-robin; a rank keeps its
    blocks compactly; in step k the ranks transform the blocks k*R .. k*R+R-1, which are exactly the columns the sequential
    leaf sponge absorbs next; FRI combines an own block as one group whose weight is alpha^(first global column)."""
    ob = 64
    for total in (1676, 762, 2822, 11786, 5328, 9802, 64, 65, 4):
        for R in (1, 2, 4, 8, 16):
            nblocks = -(-total // ob)
            steps = -(-nblocks // R)
            cols = lambda b: min(ob, total - b * ob) if b < nblocks else 0   # noqa: E731
            seen = []
            for r in range(R):
                own = [b for b in range(r, nblocks, R)]
                local = 0
                for k, b in enumerate(own):
                    assert b == k * R + r
                    assert local == k * ob                       # compact storage: only the globally last block is short,
                    for j in range(cols(b)):                     # and it is the last own block of its owner
                        g = ((local + j) // ob * R + r) * ob + (local + j) % ob   # ColShare::global_col
                        assert g == b * ob + j
                        seen.append(g)
                    local += cols(b)
                assert len(own) <= steps
            assert sorted(seen) == list(range(total))
            # the sponge of step k absorbs [k*R*ob, min(total, (k+1)*R*ob)): consecutive, complete, in order
            edges = [(k * R * ob, min(total, (k + 1) * R * ob)) for k in range(steps)]
            assert edges[0][0] == 0 and edges[-1][1] == total and all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert all((e[1] - e[0]) % 8 == 0 for e in edges[:-1])   # the sponge permutes after every 8 columns


def test_python_side_share_and_cpu_helpers():
    """split.own_columns / exchange_bytes_sent (what bench.py reports as bytes on the wire) restate prover.hip ColShare;
    sharding.effective_cpus() never exceeds the visible CPUs and honours the affinity mask."""
    sys.path.insert(0, ROOT)
    from starky_bn254_amd import sharding, split
    for total in (1676, 762, 11786, 5328, 63, 64, 65):
        for world in (1, 2, 4, 8, 16):
            own = [split.own_columns(total, world, r) for r in range(world)]
            assert sum(own) == total and max(own) - min(own) <= 64
            assert own == sorted(own, reverse=True)             # rank 0 never owns fewer columns than a later rank
    n = sharding.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    if hasattr(os, "sched_getaffinity"):
        assert n <= len(os.sched_getaffinity(0))
