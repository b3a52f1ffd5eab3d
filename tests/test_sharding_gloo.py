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


def _dry_run(*flags):
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if not k.startswith("SBN_") and k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout)


def test_bench_dry_run_plans_for_eight_ranks():
    """First contact with an 8-GPU node: `bench.py --gpus 8 ... --dry-run` prints every rank's plan without a
    GPU or a process group.  config[4]: 185 trace + 84 Z column blocks dealt round-robin, two planes per block from four ranks
    up, staging sizes = sbn_split_exchange_bytes, ~2.2 GB per xGMI link and proof; config[2]: 256 units, 32 per rank, seeds
    1000..1255; default mode: one seed per rank."""
    import starky_bn254_amd as S
    from starky_bn254_amd import split
    p = _dry_run("--gpus", "8", "--split", "--table", "fq12")
    assert p["mode"] == "split" and p["n_gpus"] == 8 and p["table"] == "Fq12ExpStark(512)" and p["degree_bits"] == 18
    assert (p["num_columns"], p["permutation_zs"], p["planes_per_block"], p["merkle_cap_subtrees_per_rank"]) == (11786, 5328, 2, 2)
    stark = S.Fq12ExpStark(512)
    sb, rb = split.exchange_bytes(stark, stark.config(), 18, 8)
    assert (p["staging_send_bytes_per_rank"], p["staging_recv_bytes_per_rank"]) == (sb, rb)
    assert len(p["ranks"]) == 8 and [r["rank"] for r in p["ranks"]] == list(range(8))
    assert sum(r["trace_column_blocks"] for r in p["ranks"]) == 185 and sum(r["trace_columns"] for r in p["ranks"]) == 11786
    assert sum(r["z_column_blocks"] for r in p["ranks"]) == 84 and sum(r["z_columns"] for r in p["ranks"]) == 5328
    assert max(r["trace_column_blocks"] for r in p["ranks"]) - min(r["trace_column_blocks"] for r in p["ranks"]) <= 1
    assert all(1.8e9 < r["bytes_per_link_per_proof"] < 2.6e9 for r in p["ranks"])
    assert p["message_bytes_per_peer_block_plane"] == 64 * 65536 * 8 and p["xgmi_links_per_gpu"] == 7
    assert "switches" in p and p["switches"]["experimental"] == "0"
    b = _dry_run("--gpus", "8", "--batch", "256", "--seed", "1000")
    assert b["mode"] == "batch" and b["seeds"] == "1000..1255" and [r["units"] for r in b["ranks"]] == [32] * 8
    assert b["ranks"][3]["first_units"] == [3, 11, 19, 27]
    d = _dry_run("--gpus", "4")
    assert d["mode"] == "default" and [r["seed"] for r in d["ranks"]] == [1000, 1001, 1002, 1003] and d["launch"].startswith("python -m torch.distributed.run")
