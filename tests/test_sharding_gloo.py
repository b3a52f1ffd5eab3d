"""world_size-2 `gloo` test of the N>1 path: independent units are dealt to ranks with no data-path
collective; rank results (here: witness digests made by the product's host trace generator) are
gathered and must equal a single-process run; the MAX-over-ranks timing reduce works."""
import os
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    import starky_bn254_amd as S
    from starky_bn254_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    units = sharding.shard_units(5, rank, world)
    local = {}
    for u in units:
        pts, _ = O.g1op_inputs(256, sharding.unit_seed(100, u))
        local[u] = sharding.digest(S.G1Stark().generate_trace(pts))
    dist.barrier()
    merged = sharding.gather_digests(local, dist)
    tmax = sharding.max_over_ranks(1.0 + rank, dist)
    if rank == 0:
        ret["merged"] = merged
        ret["tmax"] = tmax
    dist.destroy_process_group()


def test_two_rank_sharding_matches_single_process():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import starky_bn254_amd as S
    from starky_bn254_amd import sharding
    assert sharding.shard_units(5, 0, 2) == [0, 2, 4] and sharding.shard_units(5, 1, 2) == [1, 3]
    assert sorted(sum((sharding.shard_units(256, r, 8) for r in range(8)), [])) == list(range(256))
    assert all(len(sharding.shard_units(256, r, 8)) == 32 for r in range(8))      # BASELINE config[2]
    expect = {}
    for u in range(5):
        pts, _ = O.g1op_inputs(256, sharding.unit_seed(100, u))
        expect[u] = sharding.digest(S.G1Stark().generate_trace(pts))
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, ret), nprocs=2, join=True)
    assert dict(ret["merged"]) == expect
    assert ret["tmax"] == 2.0
