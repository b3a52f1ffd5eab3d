"""The reference-parity hand-over kit (tests/parity_kit.py, tests/golden/parity_kit/*.json).

CPU: the committed kit is what the oracle produces today -- inputs, witness digests and every stage of prove() in both
FRI variants -- and agrees with tests/golden/proof_digests.json.  GPU: the device prover's proof of the kit's inputs
matches the kit STAGE BY STAGE (the first differing stage is named), which is what the Rust side
(integration/rust/starky-bn254-amd/tests/parity.rs) prints for the reference's own prover.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import parity_kit as K

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KIT_DIR = os.path.join(ROOT, "tests", "golden", "parity_kit")
ORACLE_ONLY = ["mystark_lookup_fixed"]      # the reference's pinned 8-row lookup table: below the device prover's minimum size
CASES = ["mystark_rows512_seed9", "flagstark_io16_seed8", "flagu64stark_io16_seed10", "modular_rows512_seed6", "g1op_rows512_seed0", "fq12expu64_io16_seed5", "fq12exp_io16_seed3", "g1exp_io128_seed1", "fqexp_io128_seed4",
         "g2exp_io128_seed2"]
# (inputs, trace) generators of the oracle per table
GEN = {"ModularStark": ("modular_inputs", "modular_trace"), "G1Stark": ("g1op_inputs", "g1op_trace"), "G1ExpStark": ("g1exp_inputs", "g1exp_trace"),
       "G2ExpStark": ("g2exp_inputs", "g2exp_trace"), "Fq12ExpStark": ("fq12exp_inputs", "fq12exp_trace"),
       "Fq12ExpU64Stark": ("fq12expu64_inputs", "fq12expu64_trace"), "FqExpStark": ("fqexp_inputs", "fqexp_trace"),
       "FlagStark": ("flags_inputs", "flags_trace"), "FlagU64Stark": ("flags_u64_inputs", "flags_u64_trace")}


def load(case):
    return json.load(open(os.path.join(KIT_DIR, case + ".json")))


@pytest.mark.parametrize("case", CASES + ORACLE_ONLY)
def test_kit_file_is_complete_and_matches_the_proof_digests(case, golden):
    kit = load(case)
    assert set(kit["stages"]) == {"times_x", "plain"}
    for v, d in kit["stages"].items():
        assert [k for k in K.STAGES if k not in d] == [], v
        assert d["pow_ok"] and len(d["query_indices"]) == 84 and len(d["trace_cap"]) == 64 and len(d["final_poly"]) in (16, 32, 64, 128)
        assert len(d["fri_commit_caps"]) == len(d["fri_betas"])
    # the two FRI variants share every stage up to the FRI batch challenge and part ways at the first FRI commitment
    a, b = kit["stages"]["times_x"], kit["stages"]["plain"]
    assert K.first_difference(a, b) == ("fri_commit_caps" if a["fri_commit_caps"] else "final_poly")   # (8 rows: no FRI layer at all)
    assert len(kit["inputs_u32"]) == int(np.prod(kit["inputs_shape"]))
    if case in ORACLE_ONLY:
        return
    if case not in golden["proof_digests"]:          # (the two unit-test tables joined in round 3: the kit is their only digest)
        return
    g = golden["proof_digests"][case]
    assert kit["stages"]["times_x"]["proof"] == {"sha256": g["proof_sha256"], "words": g["proof_words"]}
    assert kit["trace_sha256"] == g["trace_sha256"]
    assert len(kit["inputs_u32"]) == int(np.prod(kit["inputs_shape"]))


# The two tables whose oracle proof takes minutes and tens to hundreds of GB (tools/oracle_digest.py, run once on a GPU box):
# default FRI variant only, inputs by seed (their sha256 is committed instead of 25,600 / 102,400 numbers).
BIG_CASES = ["fq12exp_io128_seed11", "fq12exp_io512_seed3"]


@pytest.mark.parametrize("case", BIG_CASES)
def test_big_kit_files_match_the_proof_digests_and_the_seeded_inputs(case, golden, O):
    kit = load(case)
    d = kit["stages"]["times_x"]
    assert [k for k in K.STAGES if k not in d] == []
    assert d["pow_ok"] and len(d["query_indices"]) == 84 and len(d["trace_cap"]) == 64 and len(d["fri_commit_caps"]) == len(d["fri_betas"]) == 3
    g = golden["proof_digests"][case]
    assert d["proof"] == {"sha256": g["proof_sha256"], "words": g["proof_words"]}
    assert kit["trace_sha256"] == g["trace_sha256"] and kit["public_inputs_sha256"] == g["pi_sha256"]
    assert g["gpu_proof_equal_on_the_box"] is True
    ios, _ = O.fq12exp_inputs(kit["num_io"], kit["seed"])
    assert list(ios.shape) == kit["inputs_shape"]
    assert hashlib.sha256(np.ascontiguousarray(ios, dtype="<u4").tobytes()).hexdigest() == kit["inputs_u32_sha256"]
    assert kit["rows"] == 512 * kit["num_io"] and kit["num_columns"] == O.lib().orc_air_num_columns(O.AIR_FQ12_EXP, kit["num_io"])


def _oracle_case(O, kit):
    if kit["table"] == "MyStark":
        ins, tab = O.lookup_inputs(kit["rows"], kit["seed"])
        assert [int(x) for x in np.stack([ins, tab]).reshape(-1)] == kit["inputs_u32"], "the committed inputs are not the seeded ones"
        return (ins, tab), O.lookup_trace(ins, tab), np.zeros(0, dtype=np.uint64)
    fin, ftr = GEN[kit["table"]]
    size = kit["num_io"] or kit["rows"]
    inputs, _ = getattr(O, fin)(size, kit["seed"])
    flat = np.asarray(inputs, dtype=np.uint64).view(np.uint32) if kit["table"] == "FlagU64Stark" else np.asarray(inputs, dtype=np.uint32)
    assert [int(x) for x in flat.reshape(-1)] == kit["inputs_u32"], "the committed inputs are not the seeded ones"
    res = getattr(O, ftr)(inputs)
    trace, pi = res if isinstance(res, tuple) else (res, np.zeros(0, dtype=np.uint64))
    return inputs, trace, pi


# G1ExpStark(128) is the headline workload: its default-variant stages are regenerated here too (about a minute on 8 cores)
@pytest.mark.parametrize("case,variants", [("mystark_rows512_seed9", ("times_x", "plain")), ("flagstark_io16_seed8", ("times_x", "plain")),
                                           ("flagu64stark_io16_seed10", ("times_x", "plain")),
                                           ("modular_rows512_seed6", ("times_x", "plain")), ("g1op_rows512_seed0", ("times_x", "plain")),
                                           ("fq12expu64_io16_seed5", ("times_x",)), ("g1exp_io128_seed1", ("times_x",))])
def test_oracle_regenerates_the_kit(O, case, variants):
    kit = load(case)
    inputs, trace, pi = _oracle_case(O, kit)
    assert hashlib.sha256(trace.tobytes()).hexdigest() == kit["trace_sha256"]
    assert hashlib.sha256(np.asarray(pi, dtype=np.uint64).tobytes()).hexdigest() == kit["public_inputs_sha256"]
    for v in variants:
        try:
            O.set_final_poly_times_x(v == "times_x")
            w, _ = O.prove(kit["air_kind"], kit["num_io"], trace, pi)
        finally:
            O.set_final_poly_times_x(True)
        got = json.loads(json.dumps(K.stage_digests(w, O.poseidon_permute)))
        assert K.first_difference(got, kit["stages"][v]) is None, (v, K.first_difference(got, kit["stages"][v]))


def test_mystark_on_the_references_fixed_input(O, golden, S):
    """MyStark (src/utils/lookup.rs:136-213) on inputs [6, 3, 1, 1, 0, 0, 0, 0] / table 0..7 (lookup.rs:154-161), the only
    workload whose inputs the reference pins: the oracle's trace equals the committed lookup fixture, its proof verifies,
    its stage digests are the kit's in both FRI variants -- and the product refuses the 8-row table instead of faking it."""
    kit = load("mystark_lookup_fixed")
    ins, tab = O.lookup_fixed_inputs()
    assert [int(x) for x in np.stack([ins, tab]).reshape(-1)] == kit["inputs_u32"]
    fx = golden["lookup_fixed"]
    assert [int(x) for x in ins] == fx["inputs"] and [int(x) for x in tab] == fx["table"]
    trace = O.lookup_trace(ins, tab)
    assert [int(x) for x in trace[2]] == fx["sorted_inputs"] and [int(x) for x in trace[3]] == fx["permuted_table"]
    assert hashlib.sha256(trace.tobytes()).hexdigest() == kit["trace_sha256"]
    for v in ("times_x", "plain"):
        try:
            O.set_final_poly_times_x(v == "times_x")
            w, _ = O.prove(O.AIR_LOOKUP, 0, trace, np.zeros(0, dtype=np.uint64))
            assert O.verify(O.AIR_LOOKUP, 0, w) == (0, "")
            bad = w.copy()
            bad[12] ^= 1
            assert O.verify(O.AIR_LOOKUP, 0, bad)[0] != 0
        finally:
            O.set_final_poly_times_x(True)
        got = json.loads(json.dumps(K.stage_digests(w, O.poseidon_permute)))
        assert K.first_difference(got, kit["stages"][v]) is None, (v, K.first_difference(got, kit["stages"][v]))
    # a wrong lookup (7 is looked up but the table column lacks it) is not provable: the constraints do not vanish
    bad_tr = trace.copy()
    bad_tr[2, 7] = 5
    lv, nv = [int(x) for x in bad_tr[:, 6]], [int(x) for x in bad_tr[:, 7]]
    assert any(O.eval_constraints(O.AIR_LOOKUP, 0, lv, nv, [], [3, 5], 1, 0, 0))


def test_transcript_replay_uses_the_plain_poseidon_definition(O, golden):
    """The kit's challenger hashes with the oracle's plain-definition permutation, which is pinned by plonky2's published
    test vectors (tests/golden/poseidon_kat.json): the replay cannot drift with the fast forms the provers use."""
    for v in golden["poseidon_kat"]["vectors"]:
        assert O.poseidon_permute([int(x, 16) for x in v["input"]]) == [int(x, 16) for x in v["output"]]
    ch = K.Challenger(O.poseidon_permute)
    ch.observe_words(range(1, 20))
    a = [ch.challenge() for _ in range(10)]
    probe = O.challenger_probe(list(range(1, 20)), 10)
    assert a == [int(x) for x in probe]


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_gpu_proof_matches_the_kit_stage_by_stage(S, O, case):
    """Device witness (Exp tables) or host witness (single-operation tables) from the kit's committed inputs, proved on the
    GPU in both FRI variants: every stage equals the kit's; a mismatch names the first stage that differs."""
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    kit = load(case)
    table = kit["table"]
    inputs = np.array(kit["inputs_u32"], dtype=np.uint32).reshape(kit["inputs_shape"])
    stark = getattr(S, table)(kit["num_io"]) if kit["num_io"] else getattr(S, table)()
    host_witness = not kit["num_io"] or table in ("FlagStark", "FlagU64Stark")      # the Exp tables build their witness on the device
    if table == "FlagU64Stark":
        inputs = np.ascontiguousarray(inputs).view(np.uint64).reshape(-1)            # (low, high) u32 pairs -> the u64 exponents
    bits = kit["rows"].bit_length() - 1
    for v in ("times_x", "plain"):
        cfg = stark.config()
        cfg.fri_variant = S.api.FRI_TIMES_X if v == "times_x" else S.api.FRI_PLAIN
        if not host_witness:
            pr = S.Prover(stark, cfg, bits)
            pi = pr.generate_trace(inputs)
            assert hashlib.sha256(np.asarray(pi, dtype=np.uint64).tobytes()).hexdigest() == kit["public_inputs_sha256"]
            proof = pr.prove()
            pr.close()
        else:
            trace = stark.generate_trace(inputs[0], inputs[1]) if table == "MyStark" else stark.generate_trace(inputs)
            assert hashlib.sha256(trace.tobytes()).hexdigest() == kit["trace_sha256"]
            proof = S.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
        got = json.loads(json.dumps(K.stage_digests(proof.words, O.poseidon_permute)))
        assert K.first_difference(got, kit["stages"][v]) is None, (case, v, K.first_difference(got, kit["stages"][v]))
