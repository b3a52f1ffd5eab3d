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
    ap.add_argument("--table", choices=["g1", "g2"], default="g1",
                    help="g1 = G1ExpStark(128), the BASELINE metric (default); g2 = G2ExpStark(128), BASELINE config[3]")
    ap.add_argument("--traffic-bytes", type=float, default=None,
                    help="PMC-measured HBM bytes per dominant-kernel launch; default: profiles/*_pmc_summary.json (separate rocprofv3 --pmc passes)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import starky_bn254_amd as S

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    rc = S.lib().sbn_set_device(local_rank)
    if rc != 0:
        raise SystemExit("sbn_set_device failed")

    # synthetic inputs: random G1 points and 256-bit scalars (src/curves/g1/exp.rs:794-809), one instance set per rank
    from starky_bn254_amd import sharding
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

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

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
    if dist is not None:
        elapsed = sharding.max_over_ranks(elapsed, dist, device=dev if args.backend == "nccl" else None)

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
        for _ in range(5):
            pi_dev = prover.generate_trace(ios)
            tg_ms = prover.stage_times()["device_tracegen_ms"]
            proof_dev = prover.prove()
        torch.cuda.synchronize()
        e2e = {"device_tracegen_ms": tg_ms, "ios_to_proof_ms_device_witness": (time.perf_counter() - t0) / 5 * 1e3,
               "same_proof_as_host_witness": bool((proof_dev.words == proof.words).all() and (pi_dev == pi).all())}

    if rank == 0:
        steps = max(args.steps, 1)
        if args.traffic_bytes is None:
            args.traffic_bytes = committed_traffic("leaf_absorb_kernel")
        n, m, C = 1 << DEGREE_BITS, 1 << (DEGREE_BITS + 1), stark.num_columns
        Zc = stark.num_permutation_zs(cfg)
        stage_ms = {k: v / steps for k, v in stage_acc.items()}
        # dominant kernel: leaf_absorb_kernel over the trace LDE (Poseidon sponge, one launch per 64-column
        # chunk on the hash stream; timed launch by launch with HIP events on that stream)
        launches = max(stage_ms.get("trace_absorb_launches", 1.0), 1.0)
        tot_bytes = 8.0 * m * C + 32.0 * m + 2 * 96.0 * m * (launches - 1)   # LDE once, digests, carried sponge state
        alg_bytes = tot_bytes / launches
        dom_ms = stage_ms.get("trace_absorb_kernels_ms", float("nan")) / launches
        achieved = alg_bytes / (dom_ms * 1e-3) / 1e9 if dom_ms and dom_ms > 0 else None
        p_cols = 1 + 3 * (Zc // 2)                         # distinct trace columns read by the permutation argument
        proof_alg_bytes = 8.0 * n * (6 * C + 7 * Zc + p_cols)   # 8.67 GB
        ms_per_step = elapsed / (steps * max(args.concurrency, 1)) * 1e3
        line = {
            "metric": "G1 scalar-mult proofs/sec at trace height 2^16" if args.table == "g1" else "G2 scalar-mult proofs/sec at trace height 2^16",
            "value": world * steps * max(args.concurrency, 1) / elapsed,
            "unit": "proofs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "config": {"workload": f"{type(stark).__name__}(num_io=128): prove() of a 2^16-row x {C}-column trace (128 scalar mults), trace resident in HBM",
                       "degree_bits": DEGREE_BITS, "num_columns": C, "num_public_inputs": stark.num_public_inputs,
                       "permutation_zs": Zc, "fri": "rate_bits=1 cap=4 arity=16 queries=84 pow_bits=16",
                       "proofs_per_rank": args.steps * max(args.concurrency, 1), "proofs_in_flight_per_gpu": max(args.concurrency, 1),
                       "parallelism": f"independent proofs x{world}, no collective"},
            "roofline": {"bound": "hbm", "kernel": "leaf_absorb_kernel (trace LDE, Poseidon sponge per row, 64-column chunks)", "launches_per_proof": launches,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None,
                         "traffic": args.traffic_bytes, "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": dom_ms,
                         "note": "ALU-bound kernel (210 Poseidon permutations per row); see DESIGN.md",
                         "valu_issue": valu_issue(dom_ms)},
            "proof_roofline": {"algorithmic_bytes_per_proof": proof_alg_bytes,
                               "achieved_GBps": proof_alg_bytes / (ms_per_step * 1e-3) / 1e9,
                               "frac_of_hbm_peak": proof_alg_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "stage_ms": stage_ms,
            "host": {"tracegen_s": t_tracegen, "h2d_s": t_h2d, "trace_bytes": int(trace.nbytes)},
        }
        if batch:
            line["batch_mode"] = batch
        if e2e:
            e2e["ios_to_proof_ms_host_witness"] = (t_tracegen + t_h2d) * 1e3 + ms_per_step / max(args.concurrency, 1) * max(args.concurrency, 1)
            line["end_to_end"] = e2e
        if world == 1 and not args.skip_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(trace, pi, args.table)
        print(json.dumps(line), flush=True)
    prover.close()
    if dist is not None:
        dist.barrier()                  # rank 0 did the extra end-to-end leg: leave together
        dist.destroy_process_group()


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


def committed_traffic(kernel):
    return committed_pmc(kernel)


def valu_issue(dom_ms):
    """The bound this kernel actually runs against: VALU issue.  Wave-instructions per launch from the committed
    SQ_INSTS_VALU pass; a wave64 VALU instruction occupies its SIMD for 4 cycles, 1,024 SIMDs at the 2.4 GHz peak clock."""
    n = committed_pmc("leaf_absorb_kernel", "valu_wave_instructions_per_launch")
    if not n or not dom_ms or dom_ms <= 0:
        return None
    peak = 1024 * 2.4e9 / 4
    ach = n / (dom_ms * 1e-3)
    return {"wave_instructions_per_launch": n, "achieved_per_s": ach, "peak_per_s": peak, "frac": ach / peak}


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
    """The CPU oracle's prove() (a restatement "port", OpenMP over all host cores) on the SAME trace.
    Sample: one full proof -- the smallest unit of this workload (the table cannot be smaller than 2^16 rows)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    words, secs = O.prove(O.AIR_G1_EXP if table == "g1" else O.AIR_G2_EXP, NUM_IO, trace, pi)
    return {"value": 1.0 / secs, "unit": "proofs/s", "cores": os.cpu_count(), "kind": "port",
            "sample": "1 full 2^16-row prove() on the same trace (oracle/, OpenMP on all host cores): the smallest unit of this workload",
            "seconds": secs}


if __name__ == "__main__":
    main()
