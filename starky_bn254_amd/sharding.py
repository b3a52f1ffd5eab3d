"""Multi-GPU layout for independent proofs (BASELINE config[2]: a batch of scalar-mult tables).

Proofs are independent units: they are dealt round-robin to ranks, every rank
proves its own units on its own GPU, and there is NO data-path collective.  The only communication is
control plane: a barrier and a MAX-reduce of the timed region, plus an optional all-gather of proof
digests so rank 0 can check the batch.  One process per GPU, `torch.distributed` (backend "nccl" =
RCCL on ROCm; "gloo" in CPU tests).
"""
import hashlib


def shard_units(num_units, rank, world_size):
    """Unit ids owned by `rank` (round-robin, as SURVEY config 2 prescribes)."""
    return list(range(rank, num_units, world_size))


def unit_seed(base_seed, unit_id):
    return base_seed + unit_id


def digest(words):
    return hashlib.sha256(words.astype("<u8").tobytes()).hexdigest()


def max_over_ranks(seconds, dist, device=None):
    """MAX of a python float over all ranks (the bench's timed region)."""
    import torch
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(local, dist):
    """All-gather {unit_id: digest} dictionaries (control plane only)."""
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, local)
    merged = {}
    for d in out:
        merged.update(d)
    return merged


def effective_cpus():
    """CPUs this process may really use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box hands a
    container all 256 hardware threads in /proc but only a share of them -- 16 per GPU on the pool this was built on; 256
    worker threads on such a share run slower than 32).  Falls back to os.cpu_count()."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:                                                      # cgroup v2
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = int(q) / int(period)
    except (OSError, ValueError):
        try:                                                  # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                quota = q / period
        except (OSError, ValueError):
            pass
    if quota:
        n = max(1, min(n, int(math.ceil(quota))))
    return n
