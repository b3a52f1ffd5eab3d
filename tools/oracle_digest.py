#!/usr/bin/env python3
"""One-off ORACLE prover runs for the tables whose CPU proof takes minutes and tens of GB (Fq12ExpStark(128): 2^16 x 10,250;
Fq12ExpStark(512) = BASELINE config[4]: 2^18 x 11,786) -- test infrastructure, like tests/golden/make_golden.py.

    python3 tools/oracle_digest.py fq12exp <num_io> <seed> [--no-gpu] [--out gpurun_out/digest_<...>.json]

Builds the seeded instance list (tests/oracle_lib.py fq12exp_inputs), lets the ORACLE generate the witness and prove it,
and writes the entry that goes into tests/golden/proof_digests.json (proof sha256, trace cap, proof-of-work witness, word
count, sha256 of witness and public inputs) together with the parity kit's per-stage digests (tests/parity_kit.py).  On a GPU
box it FIRST proves the same instance list on the device (witness generated on the device) and reports whether the two proofs
are the same words, and if not, the first stage that differs.  Prints progress lines (a run is minutes long)."""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import parity_kit as K  # noqa: E402


def log(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def mem_gb():
    try:
        for line in open("/proc/self/status"):
            if line.startswith("VmHWM"):
                return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return -1.0


def main():
    table, num_io, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    assert table == "fq12exp"
    out_path = sys.argv[sys.argv.index("--out") + 1] if "--out" in sys.argv else os.path.join(ROOT, "gpurun_out", f"digest_{table}_io{num_io}_seed{seed}.json")
    os.makedirs(os.path.dirname(out_path), exist_ok=True)
    ios, _ = O.fq12exp_inputs(num_io, seed)
    res = {"table": "Fq12ExpStark", "num_io": num_io, "seed": seed, "rows": 512 * num_io, "oracle_threads": O._effective_cpus()}
    gpu_words = None
    if "--no-gpu" not in sys.argv:
        import starky_bn254_amd as S
        stark = S.Fq12ExpStark(num_io)
        cfg = stark.config()
        prover = S.Prover(stark, cfg, (512 * num_io).bit_length() - 1)
        try:
            pi_dev = prover.generate_trace(ios)
            t0 = time.time()
            proof = prover.prove()
            res["gpu_prove_s"] = time.time() - t0
        finally:
            prover.close()
        gpu_words = proof.words.copy()
        res["gpu_proof_sha256"] = hashlib.sha256(proof.to_bytes()).hexdigest()
        res["gpu_public_inputs_sha256"] = hashlib.sha256(np.asarray(pi_dev, dtype=np.uint64).tobytes()).hexdigest()
        log(f"device proof: {len(gpu_words)} words, sha256 {res['gpu_proof_sha256'][:16]}, {res['gpu_prove_s']:.3f} s")
        del proof, prover
    t0 = time.time()
    trace, pi = O.fq12exp_trace(ios)
    res["oracle_tracegen_s"] = time.time() - t0
    log(f"oracle witness {trace.shape} = {trace.nbytes / 1e9:.1f} GB in {res['oracle_tracegen_s']:.1f} s")
    th = hashlib.sha256()
    for c0 in range(0, trace.shape[0], 256):          # tobytes() of the whole matrix would copy it
        th.update(trace[c0:c0 + 256].tobytes())
    entry = {"trace_sha256": th.hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest()}
    log("witness sha256", entry["trace_sha256"][:16], "-- proving on", res["oracle_threads"], "threads")
    words, secs = O.prove(O.AIR_FQ12_EXP, num_io, trace, pi)
    del trace
    res["oracle_prove_s"] = secs
    res["oracle_stage_s"] = O.last_stage_seconds()
    res["peak_rss_gb"] = mem_gb()
    log(f"oracle prove {secs:.1f} s, peak RSS {res['peak_rss_gb']:.1f} GB")
    assert O.verify(O.AIR_FQ12_EXP, num_io, words)[0] == 0
    entry.update({"proof_words": int(len(words)), "proof_sha256": hashlib.sha256(words.astype("<u8").tobytes()).hexdigest(),
                  "trace_cap0": [int(x) for x in words[12:16]], "pow_witness": int(words[-1 - len(pi)]),
                  "oracle_prove_seconds": secs, "oracle_threads": res["oracle_threads"]})
    res["digest_key"] = f"fq12exp_io{num_io}_seed{seed}"
    res["digest"] = entry
    res["stages"] = K.stage_digests(words, O.poseidon_permute)
    assert res["stages"]["pow_ok"]
    if gpu_words is not None:
        res["gpu_equals_oracle"] = bool(len(gpu_words) == len(words) and np.array_equal(gpu_words, words))
        if not res["gpu_equals_oracle"]:
            res["first_difference"] = K.first_difference(K.stage_digests(gpu_words, O.poseidon_permute), res["stages"])
        log("GPU proof == oracle proof:", res["gpu_equals_oracle"], res.get("first_difference", ""))
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    log("wrote", out_path)
    return 0 if res.get("gpu_equals_oracle", True) else 1


if __name__ == "__main__":
    sys.exit(main())
