#!/usr/bin/env python3
"""Headline benchmark: G1 scalar-mult proofs/sec at trace height 2^16 (BASELINE.json), MI355X.

One "step" = one prove() of G1ExpStark(num_io=128): 65,536 rows x 1,676 columns, 7,168 public inputs
(= 128 scalar multiplications), FRI blow-up 2 / 84 queries / 16-bit PoW, with the trace already
resident in HBM (witness generation and the host->device copy are outside the timed region; the
PCIe-inclusive figure is in DESIGN.md).  N > 1: one process per GPU, every rank proves its own
independent instance (weak scaling, no data-path collective); the timed region is bracketed by a
barrier + torch.cuda.synchronize() and the MAX over ranks is reported.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is 6.29 TB/s
NUM_IO = 128
DEGREE_BITS = 16


_RESULT_OUT = None


def emit(line):
    out = _RESULT_OUT if _RESULT_OUT is not None else sys.stdout
    out.write(json.dumps(line) + "\n")
    out.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--skip-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo (rehearsal)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal only: every rank uses GPU 0 (needs --backend gloo)")
    ap.add_argument("--no-batch-mode", action="store_true", help="skip the extra 3-proofs-in-flight throughput measurement")
    ap.add_argument("--concurrency", type=int, default=1,
                    help="independent proofs in flight per GPU (batch mode, BASELINE config[2]); 1 = single-proof latency (default)")
    ap.add_argument("--table", choices=["g1", "g2", "fq12"], default="g1",
                    help="g1 = G1ExpStark(128), the BASELINE metric (default); g2 = G2ExpStark(128), BASELINE config[3]; fq12 (with --split) = Fq12ExpStark")
    ap.add_argument("--batch", type=int, default=0,
                    help="BASELINE config[2] literally: this many independent G1 proofs (seeds seed+unit), units dealt round-robin to the ranks, "
                         "witness generated on the device inside the timed region, per-unit digests gathered on rank 0 (strong scaling; --steps is ignored)")
    ap.add_argument("--inflight", type=int, default=3, help="--batch: prover contexts (proofs in flight) per GPU (default 3)")
    ap.add_argument("--split", action="store_true",
                    help="BASELINE config[4]: ONE proof split over all ranks (sbn_split_prover_*, RCCL all-to-all + all-gathers); --table fq12 --num-io 512 is the config as written")
    ap.add_argument("--transport", choices=["rccl", "torch"], default="rccl",
                    help="--split with --backend nccl: rccl = the library's own RCCL transport (grouped ncclSend / ncclRecv, no Python in the data "
                         "path; default), torch = the same exchanges through torch.distributed")
    ap.add_argument("--num-io", type=int, default=None, help="instances of the table (default 128; --split --table fq12: 512)")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the per-rank plan of this mode for --gpus ranks (units per rank, or column blocks / bytes per xGMI link / staging sizes of a "
                         "split proof) as one JSON object and exit: touches no GPU and needs no process group (first contact with an 8-GPU node)")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="PMC-measured HBM bytes per dominant-kernel launch; default: profiles/*_pmc_summary.json (separate rocprofv3 --pmc passes)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.dry_run:
        print(json.dumps(dry_run_plan(args), indent=1))
        return
    # host worker pool of the library (Jacobian chains of the device witness generation, canonical checks): the ranks of a
    # node share its cores
    from starky_bn254_amd.sharding import effective_cpus
    os.environ.setdefault("SBN_HOST_THREADS", str(max(1, min(64, effective_cpus() // max(world, 1)))))
    # HIP maps a process's streams onto 4 hardware queues by default; three provers in flight hold nine streams, and the small
    # copies of witness generation then wait behind another prover's sponge launch on the same queue.  Eight queues: batches from
    # instance lists +1.6 .. 4.7 %, one proof in flight unchanged -- but ONE SPLIT proof at world 1 is 12 - 16 % slower with them (its
    # transform / exchange / sponge streams then land on queues of their own: 24.0 -> 27.9 ms for G1, 0.664 -> 0.743 s for the 2^18-row
    # table), so only --batch asks for them (profiles/r4_hwq_and_inflight_sweeps.txt).  Read by the HIP runtime when it starts, so it
    # is set before torch is imported.
    if args.batch:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

    # The contract is ONE JSON line on stdout.  RCCL prints a version banner on file descriptor 1 when its first communicator
    # comes up, so the descriptor is kept aside for the result and everything else that writes to "stdout" goes to stderr.
    global _RESULT_OUT
    if _RESULT_OUT is None:
        sys.stdout.flush()
        _RESULT_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    import numpy as np
    import torch
    import starky_bn254_amd as S

    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run for --gpus > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the prover path has no CPU fallback")
    if args.same_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.split:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        import datetime
        pg_timeout = datetime.timedelta(seconds=300)     # a rank that dies must fail the others, not hang them
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev, timeout=pg_timeout)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world, timeout=pg_timeout)

    rc = S.lib().sbn_set_device(local_rank)
    if rc != 0:
        raise SystemExit("sbn_set_device failed")
    from starky_bn254_amd import sharding

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if dist is None:
            return x
        return sharding.max_over_ranks(x, dist, device=dev if args.backend == "nccl" else None)

    if args.split:
        return bench_split(args, S, np, torch, dist, rank, world, barrier, max_over_ranks)
    if args.batch:
        return bench_batch(args, S, np, torch, dist, rank, world, barrier, max_over_ranks)
    if args.table == "fq12":
        raise SystemExit("--table fq12 is the oversized-trace workload: use it with --split")

    # synthetic inputs: random G1 points and 256-bit scalars (src/curves/g1/exp.rs:794-809), one instance set per rank
    stark = S.G1ExpStark(NUM_IO) if args.table == "g1" else S.G2ExpStark(NUM_IO)
    cfg = stark.config()
    ios = synthetic_ios(NUM_IO, sharding.unit_seed(args.seed, rank), args.table)
    t0 = time.time()
    trace, pi = stark.generate_trace_and_public_inputs(ios)
    t_tracegen = time.time() - t0
    prover = S.Prover(stark, cfg, DEGREE_BITS)
    t0 = time.time()
    prover.load_trace(trace, pi)
    t_h2d = time.time() - t0
    extra = []                      # batch mode: more provers on the same GPU, each on its own pair of streams
    for _ in range(max(args.concurrency, 1) - 1):
        p2 = S.Prover(stark, cfg, DEGREE_BITS)
        p2.load_trace(trace, pi)
        extra.append(p2)

    proof = None
    for _ in range(args.warmup):
        proof = prover.prove()
    stage_acc = {}
    barrier()
    t0 = time.perf_counter()
    if extra:
        import threading

        def worker(p):
            for _ in range(args.steps):
                p.prove()
        threads = [threading.Thread(target=worker, args=(p,)) for p in extra]
        for th in threads:
            th.start()
    for _ in range(args.steps):
        proof = prover.prove()
        for k, v in prover.stage_times().items():      # HIP-event times on the prover's stream
            stage_acc[k] = stage_acc.get(k, 0.0) + v
    if extra:
        for th in threads:
            th.join()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(elapsed)

    # every rank checks its own proof outside the timed region
    S.verify_stark_proof(stark, proof, cfg)

    # batch mode, reported beside the single-proof line: 3 independent proofs in flight on this GPU (one prover context
    # and one host thread each, include/sbn.h sbn_batch_prover_*), which fills the latency-bound tails and the
    # host-transcript gaps of one proof with the kernels of the others.  Units are instance lists: witness generation
    # (on the device) is part of the measured work.  Outside the timed region; rank 0 of a single-GPU run only.
    batch = None
    if rank == 0 and world == 1 and max(args.concurrency, 1) == 1 and not args.no_batch_mode:
        inflight, units = 3, 3 * max(4, min(args.steps, 10))
        bp = S.BatchProver(stark, cfg, DEGREE_BITS, inflight)          # C ABI: sbn_batch_prover_*
        ios_units = np.broadcast_to(ios, (units,) + ios.shape)
        bp.prove_ios(ios_units[:inflight])                             # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        proofs = bp.prove_ios(ios_units)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        batch = {"proofs_in_flight": inflight, "proofs": units, "proofs_per_s": units / dt, "ms_per_proof_throughput": dt / units * 1e3,
                 "from": "instance lists (witness generated on the device inside the timed region)",
                 "same_proofs_as_single": bool(all((p.words == proof.words).all() for p in proofs))}
        bp.close()
        # the same with the traces already resident in HBM (one prover context and host thread per proof in flight)
        import threading
        provers = [prover] + [S.Prover(stark, cfg, DEGREE_BITS) for _ in range(inflight - 1)]
        for p in provers[1:]:
            p.load_trace(trace, pi)
            p.prove()
        reps = units // inflight

        def run(p):
            for _ in range(reps):
                p.prove()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=run, args=(p,)) for p in provers]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        torch.cuda.synchronize()
        batch["resident_traces_proofs_per_s"] = inflight * reps / (time.perf_counter() - t0)
        for p in provers[1:]:
            p.close()

    # instance list -> proof with the witness generated on the device (outside the timed region, reported
    # beside the host-generator + PCIe path): wall clock of generate_trace + prove, 5 repetitions after one warm-up
    e2e = None
    if rank == 0:
        prover.generate_trace(ios)
        t0 = time.perf_counter()
        wall_gen = wall_prove = 0.0
        st_after = {}
        for _ in range(5):
            ta = time.perf_counter()
            pi_dev = prover.generate_trace(ios)
            tb = time.perf_counter()
            tg_ms = prover.stage_times()["device_tracegen_ms"]
            proof_dev = prover.prove()
            wall_gen += tb - ta
            wall_prove += time.perf_counter() - tb
            for k, v in prover.stage_times().items():
                st_after[k] = st_after.get(k, 0.0) + v / 5
        torch.cuda.synchronize()
        e2e = {"device_tracegen_ms": tg_ms, "ios_to_proof_ms_device_witness": (time.perf_counter() - t0) / 5 * 1e3,
               "generate_trace_wall_ms": wall_gen / 5 * 1e3, "prove_after_generate_wall_ms": wall_prove / 5 * 1e3,
               "stage_ms_of_prove_after_generate": {k: round(v, 3) for k, v in st_after.items() if k in ("trace_commit", "trace_absorb_kernels_ms", "perm_z", "z_commit", "z_absorb_kernels_ms", "quotient_eval", "openings", "fri_combine", "fri_layers")},
               "same_proof_as_host_witness": bool((proof_dev.words == proof.words).all() and (pi_dev == pi).all())}

    if rank == 0:
        steps = max(args.steps, 1)
        stage_ms = {k: v / steps for k, v in stage_acc.items()}
        ms_per_step = elapsed / (steps * max(args.concurrency, 1)) * 1e3
        line = {
            "metric": "G1 scalar-mult proofs/sec at trace height 2^16" if args.table == "g1" else "G2 scalar-mult proofs/sec at trace height 2^16",
            "value": world * steps * max(args.concurrency, 1) / elapsed,
            "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{type(stark).__name__}(num_io=128): prove() of a 2^16-row x {stark.num_columns}-column trace (128 scalar mults), trace resident in HBM",
                       "degree_bits": DEGREE_BITS, "num_columns": stark.num_columns, "num_public_inputs": stark.num_public_inputs,
                       "permutation_zs": stark.num_permutation_zs(cfg), "fri": "rate_bits=1 cap=4 arity=16 queries=84 pow_bits=16",
                       "proofs_per_rank": args.steps * max(args.concurrency, 1), "proofs_in_flight_per_gpu": max(args.concurrency, 1),
                       "parallelism": f"independent proofs x{world}, no collective", "host_threads_per_rank": int(os.environ["SBN_HOST_THREADS"]),
                       "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4")), "switches": prover.describe()},
            "stage_ms": stage_ms,
            "host": {"tracegen_s": t_tracegen, "h2d_s": t_h2d, "trace_bytes": int(trace.nbytes)},
        }
        line.update(rooflines(args, stark, cfg, stage_ms, ms_per_step))
        if batch:
            line["batch_mode"] = batch
            # BASELINE's unit is throughput: the two 3-proofs-in-flight figures at the top level beside `value` (= ONE proof in flight,
            # the figure every round has reported)
            line["throughput_resident_traces_3_in_flight_proofs_per_s"] = batch["resident_traces_proofs_per_s"]
            line["throughput_instance_lists_3_in_flight_proofs_per_s"] = batch["proofs_per_s"]
        if e2e:
            e2e["ios_to_proof_ms_host_witness"] = (t_tracegen + t_h2d) * 1e3 + ms_per_step / max(args.concurrency, 1) * max(args.concurrency, 1)
            line["end_to_end"] = e2e
        if world == 1 and not args.skip_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(trace, pi, args.table)
        emit(line)
    prover.close()
    if dist is not None:
        dist.barrier()                  # rank 0 did the extra end-to-end leg: leave together
        dist.destroy_process_group()


def rooflines(args, stark, cfg, stage_ms, ms_per_proof, degree_bits=DEGREE_BITS):
    """`roofline` (dominant kernel) and `proof_roofline` (whole proof) objects of the bench line."""
    traffic, source = args.traffic_bytes, "--traffic-bytes argument"
    if traffic is None:
        traffic, source = committed_pmc("leaf_absorb_kernel"), committed_pmc_file() or "no committed PMC summary"
        source = f"committed rocprofv3 --pmc passes ({source}), not measured in this run"
    n, m, C = 1 << degree_bits, 1 << (degree_bits + 1), stark.num_columns
    Zc = stark.num_permutation_zs(cfg)
    # dominant kernel: leaf_absorb_kernel over the trace LDE (Poseidon sponge, one launch per 64-column
    # chunk on the hash stream; timed launch by launch with HIP events on that stream)
    launches = max(stage_ms.get("trace_absorb_launches", 1.0), 1.0)
    # a launch absorbs cols_in_launch columns of the M-row LDE (8 M bytes each) and the last one writes
    # the M digests (32 M bytes); averaged over the launches of one proof.  The sponge state carried between the 64-column
    # launches (its capacity, [4][M] words out and in again) is an artefact of the chunking, not algorithmic traffic: reported apart.
    alg_bytes = (8.0 * m * C + 32.0 * m) / launches
    chunk_state_bytes = 2 * 32.0 * m * (launches - 1) / launches     # the sponge capacity [4][M] out and in again (the rate part is overwritten)
    dom_ms = stage_ms.get("trace_absorb_kernels_ms", float("nan")) / launches
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else None
    p_cols = 1 + 3 * (Zc // 2)                         # distinct trace columns read by the permutation argument
    proof_alg_bytes = 8.0 * n * (6 * C + 7 * Zc + p_cols)   # 8.67 GB for G1
    perms_per_launch = m * -(-C // 8) / launches        # one sponge permutation per row and 8 absorbed columns
    proof_traffic, traffic_file = committed_proof_traffic()
    return {
        "roofline": {"bound": "hbm", "kernel": "leaf_absorb_kernel (trace LDE, Poseidon sponge per row, 64-column chunks)", "launches_per_proof": launches,
                     "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                     "traffic": traffic, "traffic_source": source, "algorithmic_bytes_per_launch": alg_bytes, "chunk_state_bytes": chunk_state_bytes,
                     "avg_launch_ms": dom_ms,
                     "note": "VALU-issue-bound kernel (ceil(C/8) Poseidon permutations per row): the HBM fraction is reported as the contract asks, the bound it runs against is valu_issue",
                     "valu_issue": valu_issue(dom_ms, perms_per_launch)},
        "proof_roofline": {"algorithmic_bytes_per_proof": proof_alg_bytes,
                           "achieved_GBps": proof_alg_bytes / (ms_per_proof * 1e-3) / 1e9,
                           "frac_of_hbm_peak": proof_alg_bytes / (ms_per_proof * 1e-3) / 1e9 / HBM_PEAK_GBS,
                           "proof_traffic_bytes": proof_traffic if args.table == "g1" else None,
                           "traffic_over_algorithmic": (proof_traffic / proof_alg_bytes) if (proof_traffic and args.table == "g1") else None,
                           "proof_traffic_source": f"sum over the prover's kernels of PMC bytes per launch x launches per proof, {traffic_file} (G1ExpStark(128); not measured in this run)"},
    }


def bench_batch(args, S, np, torch, dist, rank, world, barrier, max_over_ranks):
    """BASELINE config[2] as written: a batch of `--batch` independent G1 scalar-mult proofs (seeds seed + unit), dealt
    round-robin to the ranks (sharding.shard_units), every rank proving its units with 3 proofs in flight from instance
    lists (witness generated on the device INSIDE the timed region).  No data-path collective; digests gathered for rank 0."""
    from starky_bn254_amd import sharding
    if args.table != "g1":
        raise SystemExit("--batch is the G1 batch of BASELINE config[2]")
    stark = S.G1ExpStark(NUM_IO)
    cfg = stark.config()
    units = sharding.shard_units(args.batch, rank, world)
    t0 = time.time()
    ios_units = np.stack([synthetic_ios(NUM_IO, sharding.unit_seed(args.seed, u), "g1") for u in units]) if units else np.zeros((0, NUM_IO, 40), np.uint32)
    t_inputs = time.time() - t0
    inflight = max(1, min(args.inflight, 16))
    bp = S.BatchProver(stark, cfg, DEGREE_BITS, inflight)
    if len(units):
        bp.prove_ios(ios_units[:min(inflight, len(units))])            # warm-up (untimed)
    barrier()
    t0 = time.perf_counter()
    proofs = bp.prove_ios(ios_units) if len(units) else []
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(elapsed)
    local = {int(u): sharding.digest(p.words) for u, p in zip(units, proofs)}
    for p in proofs[:2]:
        S.verify_stark_proof(stark, p, cfg)                              # a sample per rank, outside the timed region
    merged = sharding.gather_digests(local, dist) if dist is not None else local
    if rank == 0:
        ok = sorted(merged) == list(range(args.batch)) and len(set(merged.values())) == args.batch
        line = {"metric": "G1 scalar-mult proofs/sec at trace height 2^16", "value": args.batch / elapsed, "unit": "proofs/s", "n_gpus": world,
                "steps": len(units), "warmup": min(inflight, len(units)), "ms_per_step": elapsed / max(len(units), 1) * 1e3, "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"batch of {args.batch} independent G1ExpStark(128) proofs (2^16 rows x 1676 columns each), instance list -> device witness -> proof, "
                                       f"{inflight} proofs in flight per GPU (BASELINE config[2])",
                           "units_per_rank": len(units), "seeds": f"{args.seed}..{args.seed + args.batch - 1}", "parallelism": f"units round-robin over {world} ranks, no collective",
                           "host_threads_per_rank": int(os.environ["SBN_HOST_THREADS"]), "hip_hw_queues": int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))},
                "batch_check": {"digests_gathered": len(merged), "all_units_present_and_distinct": bool(ok)},
                "host": {"synthetic_inputs_s": t_inputs}}
        emit(line)
    bp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def bench_split(args, S, np, torch, dist, rank, world, barrier, max_over_ranks):
    """BASELINE config[4]: ONE proof split over the ranks (sbn_split_prover_*): columns for NTT / LDE / Z / openings / FRI
    combine, Merkle-subtree rows for hashing / constraints / queries, RCCL all-to-all in between.  A step = one whole proof."""
    from starky_bn254_amd.split import SplitProver, exchange_bytes_sent
    num_io = args.num_io or (512 if args.table == "fq12" else NUM_IO)
    stark = {"g1": S.G1ExpStark, "g2": S.G2ExpStark, "fq12": S.Fq12ExpStark}[args.table](num_io)
    cfg = stark.config()
    bits = (512 * num_io).bit_length() - 1
    ios = synthetic_ios_fq12(num_io, args.seed) if args.table == "fq12" else synthetic_ios(num_io, args.seed, args.table)
    native = args.backend == "nccl" and args.transport == "rccl"
    sp = SplitProver(stark, cfg, bits, staged=(args.backend != "nccl"), transport="rccl" if native else "torch")
    selftest = None
    if world > 1:
        # first contact with a multi-GPU node: a pattern exchange over the very transport the proof will use (uneven blocks, the
        # all-gather form, the host all-gather), checked on the device, BEFORE the first proof -- a wrong byte names the receiving
        # rank and the block (= sending rank) instead of surfacing as a proof that does not verify
        t0 = time.perf_counter()
        sp.selftest()
        selftest = {"passed": True, "seconds": time.perf_counter() - t0}
    sp.generate_trace(ios)
    proof = None
    for _ in range(args.warmup):
        proof = sp.prove()
    acc = {}
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        proof = sp.prove()
        for k, v in sp.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    elapsed = max_over_ranks(elapsed)
    S.verify_stark_proof(stark, proof, cfg)
    digests = sharding_gather({rank: digest_of(proof)}, dist)
    if rank == 0:
        steps = max(args.steps, 1)
        n, C, Zc = 1 << bits, stark.num_columns, stark.num_permutation_zs(cfg)
        p_cols = {"g1": 1 + 3 * (Zc // 2), "g2": 1 + 3 * (Zc // 2), "fq12": 1 + 6 * (Zc // 4)}[args.table]
        alg = 8.0 * n * (6 * C + 7 * Zc + p_cols)
        ms = elapsed / steps * 1e3
        line = {"metric": f"{type(stark).__name__} proofs/sec at trace height 2^{bits}, one proof split over {world} GPU(s)", "value": steps / elapsed, "unit": "proofs/s",
                "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "strong",
                "vs_baseline": None, "dtype": "u64", "data": "synthetic",
                "config": {"workload": f"{type(stark).__name__}(num_io={num_io}): ONE prove() of a 2^{bits}-row x {C}-column trace split over {world} rank(s) "
                                       f"(BASELINE config[4]{' as written' if args.table == 'fq12' and num_io == 512 else ''}), witness resident on every rank",
                           "degree_bits": bits, "num_columns": C, "permutation_zs": Zc, "backend": ("RCCL, native transport (sbn_rccl_comm_create)" if native else "RCCL (torch.distributed nccl)") if args.backend == "nccl" else "host-staged gloo",
                           "parallelism": f"64-column blocks dealt round-robin -> pipelined all-to-all per block -> row shard (Merkle cap subtrees), {world} ranks"},
                "proof_roofline": {"algorithmic_bytes_per_proof": alg, "achieved_GBps": alg / (ms * 1e-3) / 1e9,
                                   "frac_of_aggregate_hbm_peak": alg / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world)},
                "stage_ms_rank0": {k: v / steps for k, v in acc.items()},
                "exchange": {"bytes_sent_per_proof_rank0": exchange_bytes_sent(stark, cfg, bits, world, 0), "exchange_ms_rank0": acc.get("split_exchange_ms", 0.0) / steps,
                             "transport_selftest": selftest},
                "all_ranks_same_proof": len(set(digests.values())) == 1}
        emit(line)
    sp.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def dry_run_plan(args):
    """The per-rank plan of `--gpus N [--batch B | --split]` without touching a GPU: what every rank will own, send and allocate.
    Library calls used: table shapes and sbn_split_exchange_bytes (host arithmetic) and sbn_settings_check."""
    import starky_bn254_amd as S
    from starky_bn254_amd import sharding, split
    world = max(args.gpus, 1)
    plan = {"mode": "split" if args.split else ("batch" if args.batch else "default"), "n_gpus": world,
            "launch": f"python -m torch.distributed.run --nnodes=1 --nproc-per-node {world} --master-addr 127.0.0.1 --master-port P bench.py --gpus {world} ..." if world > 1 else "python bench.py ...",
            "host_cpus_effective": sharding.effective_cpus(), "host_threads_per_rank": max(1, min(64, sharding.effective_cpus() // world))}
    try:
        plan["switches"] = S.api.settings_check()
    except S.SbnError as e:
        plan["switches_error"] = str(e)
    if args.split:
        num_io = args.num_io or (512 if args.table == "fq12" else NUM_IO)
        stark = {"g1": S.G1ExpStark, "g2": S.G2ExpStark, "fq12": S.Fq12ExpStark}[args.table](num_io)
        cfg = stark.config()
        bits = (512 * num_io).bit_length() - 1
        C, Z, m = stark.num_columns, stark.num_permutation_zs(cfg), 2 << bits
        sb, rb = split.exchange_bytes(stark, cfg, bits, world)
        planes = 2 if world >= 4 else 1
        plan.update({"table": f"{type(stark).__name__}({num_io})", "degree_bits": bits, "num_columns": C, "permutation_zs": Z, "column_block": 64,
                     "planes_per_block": planes, "staging_send_bytes_per_rank": sb, "staging_recv_bytes_per_rank": rb,
                     "merkle_cap_subtrees_per_rank": 16 // world if world <= 16 else 0, "lde_rows_per_rank": m // world,
                     "message_bytes_per_peer_block_plane": 64 * (m // world) * 8, "xgmi_links_per_gpu": max(world - 1, 0)})
        ranks = []
        for r in range(world):
            tc, zc = split.own_columns(C, world, r), split.own_columns(Z, world, r)
            sent = split.exchange_bytes_sent(stark, cfg, bits, world, r)
            ranks.append({"rank": r, "trace_column_blocks": len(range(r, -(-C // 64), world)), "trace_columns": tc, "z_column_blocks": len(range(r, -(-Z // 64), world)),
                          "z_columns": zc, "bytes_sent_per_proof": sent, "bytes_per_link_per_proof": sent // max(world - 1, 1)})
        plan["ranks"] = ranks
    elif args.batch:
        plan.update({"table": f"G1ExpStark({NUM_IO})", "units": args.batch, "seeds": f"{args.seed}..{args.seed + args.batch - 1}", "proofs_in_flight_per_gpu": max(1, min(args.inflight, 16)),
                     "collectives": "none on the data path (barrier, MAX of the timed region, all-gather of digests)",
                     "ranks": [{"rank": r, "units": len(sharding.shard_units(args.batch, r, world)), "first_units": sharding.shard_units(args.batch, r, world)[:4]} for r in range(world)]})
    else:
        plan.update({"table": f"{'G2' if args.table == 'g2' else 'G1'}ExpStark({NUM_IO})", "proofs_per_rank": args.steps, "scaling": "weak",
                     "collectives": "none on the data path (barrier, MAX of the timed region)",
                     "ranks": [{"rank": r, "seed": sharding.unit_seed(args.seed, r), "proofs": args.steps} for r in range(world)]})
    return plan


def digest_of(proof):
    import hashlib
    return hashlib.sha256(proof.to_bytes()).hexdigest()


def sharding_gather(local, dist):
    if dist is None or dist.get_world_size() == 1:
        return local
    from starky_bn254_amd import sharding
    return sharding.gather_digests(local, dist)


def synthetic_ios_fq12(num_io, seed):
    """num_io x (x, offset: 12 flat-basis Fq coefficients each; exponent: uniform mod r), u32 LE limbs (src/fields/fq12/exp.rs:647-660)."""
    import numpy as np
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 200), dtype=np.uint32)
    for k in range(num_io):
        for c in range(24):
            v = int.from_bytes(rng.bytes(32), "little") % P
            ios[k, 8 * c:8 * c + 8] = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        e = int.from_bytes(rng.bytes(32), "little") % R
        ios[k, 192:200] = [(e >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
    return ios


def committed_pmc(kernel, key="hbm_bytes_per_launch_corrected"):
    """A per-launch figure of `kernel` from the newest committed PMC summary (tools/summarize_pmc.py)."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])   # r1_v10 after r1_v9
    if not files:
        return None
    d = json.load(open(files[-1]))
    return d.get(kernel, {}).get(key)


def committed_proof_traffic():
    """HBM bytes one G1 proof moves, all kernels together, from the newest committed PMC summary: per kernel, corrected bytes
    per launch x launches, divided by the proofs of the profiled run (quotient_combine_kernel runs once per proof).  Witness
    generation (tg::*) and the one-time table kernels of prover creation are not part of prove()."""
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    proofs = d.get("quotient_combine_kernel", {}).get("launches_fetch_pass", 0)
    if not proofs:
        return None, os.path.relpath(files[-1], ROOT)
    skip = ("tg::", "pow_table_kernel", "domain_tables_kernel")
    tot = sum(v.get("hbm_bytes_per_launch_corrected", 0.0) * v.get("launches_fetch_pass", 0) for k, v in d.items() if isinstance(v, dict) and not k.startswith(skip))
    return tot / proofs, os.path.relpath(files[-1], ROOT)


def committed_pmc_file():
    import glob
    import re
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_summary.json")),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    return os.path.relpath(files[-1], ROOT) if files else None


def valu_issue(dom_ms, perms_per_launch):
    """The bound this kernel actually runs against: VALU issue.  The instruction mix of one permutation (tools/poseidon_mix.py,
    from the hand-scheduled streams) is priced with the MEASURED issue cost of every instruction class at the kernel's 2
    waves per SIMD (tools/microbench/valu_rates.hip -> profiles/r2_valu_rates.txt: v_mad_u64_u32, carry adds and VOP3 ops
    hold a SIMD ~2 ns whatever the occupancy; only plain VOP2 adds / moves reach ~1 ns with a partner wave) -- the time the
    1,024 SIMDs need just to ISSUE a launch's instructions, against the measured launch time."""
    import glob
    import re
    n = committed_pmc("leaf_absorb_kernel", "valu_wave_instructions_per_launch")
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sponge_issue_model.json")), key=lambda f: [int(x) for x in re.findall(r"\d+", os.path.basename(f))])
    if not dom_ms or dom_ms <= 0:
        return None
    out = {"wave_instructions_per_launch": n, "achieved_per_s": (n / (dom_ms * 1e-3)) if n else None}
    if files:
        m = json.load(open(files[-1]))
        wave_perms_per_simd = perms_per_launch / 64 / 1024      # wave64 permutations of one launch per SIMD (1,024 SIMDs)
        issue_ms = m["issue_us_per_wave_permutation_per_simd"] * wave_perms_per_simd * 1e-3
        # two fractions: against the MODEL (the stream priced with measured per-class issue costs -- a model, not a bound: a launch can
        # beat it) and against the CLOCK peak (one wave-instruction per SIMD every 4 cycles at 2.4 GHz: a bound)
        clock_peak = 1024 * 2.4e9 / 4
        out.update({"issue_model_ms_per_launch": issue_ms, "frac_of_model": issue_ms / dom_ms, "model_peak_per_s": 1024 / (m["weighted_ns_per_wave_instruction"] * 1e-9),
                    "clock_peak_per_s": clock_peak, "frac_of_clock_peak": (out["achieved_per_s"] / clock_peak) if out["achieved_per_s"] else None,
                    "model": os.path.relpath(files[-1], ROOT), "rates": m["source"]})
    else:
        peak = 1024 * 2.4e9 / 4
        out.update({"clock_peak_per_s": peak, "frac_of_clock_peak": out["achieved_per_s"] / peak if out["achieved_per_s"] else None, "model": "4 cycles at 2.4 GHz (no committed model)"})
    return out


def synthetic_ios(num_io, seed, table="g1"):
    """num_io x (x, offset: random affine points; exp_val: 8 uniform u32 limbs), u32 LE limbs.
    g1: points on y^2 = x^3 + 3 over Fq; g2: points on the twist y^2 = x^3 + 3/(9+i) over Fq2."""
    import numpy as np
    P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    rng = np.random.default_rng(seed)
    if table == "g2":
        return synthetic_ios_g2(num_io, rng, P)

    def point():
        while True:
            x = int.from_bytes(rng.bytes(32), "little") % P
            rhs = (x * x * x + 3) % P
            y = pow(rhs, (P + 1) // 4, P)
            if y * y % P == rhs:
                return x, (P - y if rng.integers(0, 2) else y)

    ios = np.zeros((num_io, 40), dtype=np.uint32)
    for k in range(num_io):
        for j, v in enumerate(point() + point()):
            ios[k, 8 * j:8 * j + 8] = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        ios[k, 32:40] = rng.integers(0, 1 << 32, size=8, dtype=np.uint64)
    return ios


def synthetic_ios_g2(num_io, rng, P):
    import numpy as np

    def mul(a, b):
        return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)

    def add(a, b):
        return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)

    def inv(a):
        n = pow(a[0] * a[0] + a[1] * a[1], -1, P)
        return (a[0] * n % P, (-a[1]) * n % P)

    def fpow(a, e):
        r = (1, 0)
        while e:
            if e & 1:
                r = mul(r, a)
            a = mul(a, a)
            e >>= 1
        return r

    def sqrt(a):                                   # p = 3 mod 4
        a1 = fpow(a, (P - 3) // 4)
        alpha = mul(mul(a1, a1), a)
        if mul(fpow(alpha, P), alpha) == (P - 1, 0):
            return None
        x0 = mul(a1, a)
        if alpha == (P - 1, 0):
            return mul((0, 1), x0)
        return mul(fpow(add((1, 0), alpha), (P - 1) // 2), x0)

    b2 = mul((3, 0), inv((9, 1)))

    def point():
        while True:
            x = (int.from_bytes(rng.bytes(32), "little") % P, int.from_bytes(rng.bytes(32), "little") % P)
            rhs = add(mul(mul(x, x), x), b2)
            y = sqrt(rhs)
            if y is not None and mul(y, y) == rhs:
                return [x[0], x[1], y[0], y[1]]

    ios = np.zeros((num_io, 72), dtype=np.uint32)
    for k in range(num_io):
        for j, v in enumerate(point() + point()):
            ios[k, 8 * j:8 * j + 8] = [(v >> (32 * i)) & 0xFFFFFFFF for i in range(8)]
        ios[k, 64:72] = rng.integers(0, 1 << 32, size=8, dtype=np.uint64)
    return ios


def cpu_baseline(trace, pi, table="g1"):
    """The CPU oracle's prove() (a restatement "port", OpenMP) on the SAME trace, on the host cores this process really has.
    Sample: one full proof -- the smallest unit of this workload (the table cannot be smaller than 2^16 rows).
    Thread count: the box shows every hardware thread of the host (256) but gives a container a share of them (cgroup
    cpu.max = 16 CPUs on the pool this was built on), and the oracle is SLOWER with 256 threads than with 32 on such a share
    (profiles/r3_oracle_scaling.jsonl).  So the count is the fastest of a short calibration (a 2^11-row G1Stark proof) over
    the cgroup / affinity limit and twice that limit, or, when no limit is visible, over 16, 32, 64, ... threads; every
    calibration point is reported, `cores` is the winner."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib as O
    from starky_bn254_amd.sharding import effective_cpus
    visible, eff = os.cpu_count() or 1, effective_cpus()
    calib = {}
    pts, _ = O.g1op_inputs(2048, 3)
    tr = O.g1op_trace(pts)
    cands = sorted({t for t in ((eff, 2 * eff) if eff < visible else (16, 32, 64, 128, visible)) if 1 <= t <= visible})
    for t in cands:
        O.lib().orc_set_threads(t)
        O.prove(O.AIR_G1_OP, 0, tr, np.zeros(0, dtype=np.uint64))                      # warm the allocator at this width
        calib[t] = min(O.prove(O.AIR_G1_OP, 0, tr, np.zeros(0, dtype=np.uint64))[1] for _ in range(2))
    threads = min(calib, key=calib.get)
    O.lib().orc_set_threads(threads)
    reps, runs = 3, []                  # ~15 s of CPU work on the box: three whole proofs (one proof is the smallest unit of this workload)
    for _ in range(reps):
        words, secs = O.prove(O.AIR_G1_EXP if table == "g1" else O.AIR_G2_EXP, NUM_IO, trace, pi)
        runs.append(secs)
    secs = sum(runs) / reps
    return {"value": 1.0 / secs, "unit": "proofs/s", "cores": threads, "kind": "port",
            "kind_note": "restated C++ port (oracle/), NOT the Rust reference: the reference cannot be built in this image",
            "sample": f"{reps} full 2^16-row prove() calls on the same trace (oracle/, OpenMP, {threads} threads), mean; one proof is the smallest unit of this workload",
            "host_cpus_visible": visible, "host_cpus_effective": eff, "thread_calibration_s": {str(k): round(v, 3) for k, v in calib.items()},
            "seconds": secs, "seconds_per_run": [round(x, 3) for x in runs], "stage_seconds": O.last_stage_seconds()}


if __name__ == "__main__":
    main()
