"""GPU parity tests (run on the MI355X box with `-m gpu`): every call goes through the C ABI of
libsbn254.so and is compared bit-for-bit with the CPU oracle, the committed golden digests, and -- at
the full BASELINE size -- through size-independent properties (independent verifier accepts, tamper
rejection, determinism)."""
import hashlib
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
P = 0xFFFFFFFF00000001
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gpu(S):
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device: GPU tests must run on the MI355X box (there is no CPU fallback)")
    return S


def test_poseidon_permutation_batch(gpu, O, golden):
    kat = golden["poseidon_kat"]["vectors"]
    states = np.array([[int(x, 16) for x in v["input"]] for v in kat], dtype=np.uint64)
    out = gpu.poseidon_permute_batch(states)
    assert [[int(x) for x in r] for r in out] == [[int(x, 16) for x in v["output"]] for v in kat]
    rng = np.random.default_rng(5)
    st = rng.integers(0, P, size=(4096, 12), dtype=np.uint64)
    st[0] = P - 1; st[1] = 0; st[2, :] = [P - 1, 0, 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, 1 << 63, 2, 3, 4, 5, 6]
    out = gpu.poseidon_permute_batch(st)
    for r in list(range(8)) + [100, 4095]:
        assert [int(x) for x in out[r]] == O.poseidon_permute([int(x) for x in st[r]])
    with pytest.raises(gpu.SbnError) as e:               # non-canonical input is refused
        gpu.poseidon_permute_batch(np.full((1, 12), P, dtype=np.uint64))
    assert e.value.code == -2


def test_device_field_multiply_edge_cases(gpu):
    """The 13-instruction multiply from single-instruction primitives (csrc/gl.cuh gl_mul_weak: operator* of every kernel and the
    transforms' twiddle multiply) on arbitrary 64-bit representatives, against Python integers: operands that make the cross-term
    carry (2^96 = -1 as a borrow-in), the reduction's borrow and the borrow of lo(P3) - b (k 2^48 times j 2^48) fire, and a random sweep."""
    sp = [0, 1, 2, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, 1 << 63, 2**64 - 1, P, P - 1, P + 1, 2**64 - 2**32, 0xFFFFFFFE00000001, 0x00000001FFFFFFFF]
    sp += [k << 48 for k in (1, 3, 255, 256, 4095, 0x3B05, 0x1345, 65535)]
    a = [x for x in sp for _ in sp]
    b = [y for _ in sp for y in sp]
    rng = np.random.default_rng(31)
    ra = rng.integers(0, 2**64, size=1 << 16, dtype=np.uint64)
    rb = rng.integers(0, 2**64, size=1 << 16, dtype=np.uint64)
    rb[::3] &= np.uint64(0xFFFFFFFF00000000); ra[::5] &= np.uint64(0xFFFF000000000000)
    a = np.concatenate([np.array(a, dtype=np.uint64), ra]); b = np.concatenate([np.array(b, dtype=np.uint64), rb])
    want = [(int(x) * int(y)) % P for x, y in zip(a, b)]
    for mode in (0, 1):
        got = gpu.field_mul_batch(a, b, mode)
        assert [int(v) for v in got] == want, mode


def test_poseidon_reduction_edge_cases(gpu, O):
    """The 13-instruction multiply of the S-box streams (tools/gen_poseidon_sbox_asm.py) has three rare paths: the
    cross-term carry (weight 2^96 = -1, taken as a borrow-in), the borrow of the 2^96 term, and the borrow of
    lo(P3) - b (only when lo(P3) = 0).  States are built so that the FIRST round's S-box inputs x = state + rc hit them:
    x = k 2^48 squares to k^2 2^96 (lo(P0) = lo(P2) = lo(P3) = 0 with hi(P3) > 0: all three borrows), x near 2^32, 2^64
    and p exercise the carries.  Every row is compared with the oracle's plain-definition permutation."""
    rc = O.round_constants()[:12]
    xs = [k << 48 for k in (1, 2, 3, 255, 256, 257, 4095, 32768, 65535)]
    xs += [(k << 48) + d for k in (1, 65535) for d in (1, 0xFFFFFFFF, 1 << 32)]
    xs += [0, 1, 0xFFFFFFFF, 1 << 32, (1 << 32) + 1, P - 1, P - 2, P - (1 << 32), 0xFFFFFFFF00000000 - 1, 0xFFFFFFFE00000001,
           0x00000001FFFFFFFF, 0x8000000000000000, 0x7FFFFFFFFFFFFFFF, 0xFFFF0000FFFF0000 % P, 0x0000FFFF0000FFFF]
    rng = np.random.default_rng(77)
    rows = []
    for x in xs:                                      # the special value in every position, random elsewhere
        for pos in range(12):
            r = [int(v) for v in rng.integers(0, P, size=12, dtype=np.uint64)]
            r[pos] = (x - rc[pos]) % P
            rows.append(r)
    for x in xs:                                      # and in all twelve positions at once (three interleaved streams)
        rows.append([(x - rc[i]) % P for i in range(12)])
    st = np.array(rows, dtype=np.uint64)
    out = gpu.poseidon_permute_batch(st)
    for r in range(len(rows)):
        assert [int(v) for v in out[r]] == O.poseidon_permute(rows[r]), f"row {r}"
    big = rng.integers(0, P, size=(1 << 16, 12), dtype=np.uint64)      # a broad random sweep against the host definition
    dev = gpu.poseidon_permute_batch(big)
    assert np.array_equal(dev, gpu.poseidon_permute_host(big, use_definition=True))


@pytest.mark.parametrize("ncols,n", [(1, 512), (3, 512), (4, 1024), (5, 512), (8, 2048), (9, 4096), (20, 1024), (17, 65536), (130, 16384), (9, 131072), (3, 262144), (66, 262144)])
def test_commit_matches_oracle(gpu, O, ncols, n):
    """PolynomialBatch::from_values: iNTT, coset LDE (shift 7, blow-up 2), Poseidon leaves, Merkle cap.
    Covers <=4 columns (hash_or_noop copies the row), a ragged last sponge block, and NTT sizes that
    use one, and both, LDS tile widths."""
    rng = np.random.default_rng(ncols * 1000 + n)
    cols = rng.integers(0, P, size=(ncols, n), dtype=np.uint64)
    cols[0, :4] = [0, P - 1, 1, 0xFFFFFFFF00000000]
    cap, co, lde = gpu.commit_values(cols, want_coeffs=True, want_lde=True)
    rcap, rco, rlde = O.commit_values(cols, want_coeffs=True, want_lde=True)
    assert np.array_equal(co, rco)
    assert np.array_equal(lde, rlde)
    assert np.array_equal(cap, rcap)


@pytest.mark.parametrize("n", [65536, 131072, 262144])
def test_commit_edge_values_in_the_register_passes(gpu, O, n):
    """The register transform passes work on WEAK values (any u64 congruent mod p) whose carries are repaid twice; a uniformly random
    column reaches the second repayment once in ~2^32 operations.  Columns built from the values around 0, 2^32 and p (constant,
    alternating, in runs of 16 / 256 / 4096 rows, and mixed at random) push sums and differences to the wrap-around points all the time:
    coefficients, LDE and cap must still equal the oracle's (canonical arithmetic).  One size per fused middle pass (2^16, 2^17, 2^18 rows)."""
    edge = np.array([0, 1, 2, 0xFFFFFFFF, 0x100000000, 0x100000001, P - 1, P - 2, P - 0xFFFFFFFF, P - 0x100000000, P - 0x100000001,
                     0xFFFFFFFE00000001, 0x7FFFFFFF80000000, 0xFFFFFFFF00000000 - 1, 0x8000000000000000, 0xFFFFFFFE00000000], dtype=np.uint64)
    assert (edge < P).all()
    rng = np.random.default_rng(n)
    cols = np.zeros((16, n), dtype=np.uint64)
    cols[0, :] = P - 1
    cols[1, :] = np.where(np.arange(n) % 2 == 0, 0, P - 1)
    cols[2, :] = 0xFFFFFFFF
    cols[3, :] = np.where(np.arange(n) % 2 == 0, P - 1, 1)
    for c, run in ((4, 16), (5, 256), (6, 4096)):
        cols[c, :] = edge[(np.arange(n) // run) % len(edge)]
    cols[7, :] = edge[np.arange(n) % len(edge)]
    for c in range(8, 16):
        cols[c, :] = edge[rng.integers(0, len(edge), size=n)]
    cols[15, ::3] = rng.integers(0, P, size=len(cols[15, ::3]), dtype=np.uint64)
    cap, co, lde = gpu.commit_values(cols, want_coeffs=True, want_lde=True)
    rcap, rco, rlde = O.commit_values(cols, want_coeffs=True, want_lde=True)
    assert np.array_equal(co, rco)
    assert np.array_equal(lde, rlde)
    assert np.array_equal(cap, rcap)


def test_commit_linearity_large(gpu):
    """Size-independent property at the G1 trace height: LDE is linear, LDE(a) + LDE(b) == LDE(a+b) mod p."""
    rng = np.random.default_rng(9)
    n = 65536
    a = rng.integers(0, P, size=(6, n), dtype=np.uint64)
    b = rng.integers(0, P, size=(6, n), dtype=np.uint64)
    s = ((a.astype(object) + b.astype(object)) % P).astype(np.uint64)
    _, _, la = gpu.commit_values(a, want_lde=True)
    _, _, lb = gpu.commit_values(b, want_lde=True)
    _, _, ls = gpu.commit_values(s, want_lde=True)
    assert np.array_equal(((la.astype(object) + lb.astype(object)) % P).astype(np.uint64), ls)


@pytest.mark.parametrize("rows,seed", [(512, 0), (1024, 3)])
def test_g1stark_proof_bit_exact(gpu, O, golden, rows, seed):
    """BASELINE config[0] (plumbing): G1Stark proof bytes == oracle proof bytes == committed digest."""
    pts, _ = O.g1op_inputs(rows, seed)
    trace = gpu.G1Stark().generate_trace(pts)
    stark = gpu.G1Stark()
    cfg = stark.config()
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    ref, _ = O.prove(O.AIR_G1_OP, 0, trace, np.zeros(0, dtype=np.uint64))
    assert np.array_equal(proof.words, ref)
    g = golden["proof_digests"][f"g1op_rows{rows}_seed{seed}"]
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(O.AIR_G1_OP, 0, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    assert proof.recover_degree_bits(cfg) == rows.bit_length() - 1


@pytest.mark.parametrize("table,rows,seed", [("ModularStark", 512, 6), ("ModularStark", 2048, 26), ("Fq12Stark", 512, 7)])
def test_single_operation_tables_proof_bit_exact(gpu, O, golden, table, rows, seed):
    """The reference's test tables for the modular gadget and the Fq12 product (test_modular_stark, modular.rs:540-560;
    test_fq12_mul, fq12/mul.rs): GPU proof bytes == oracle proof bytes == committed digest, both verifiers accept."""
    stark = getattr(gpu, table)()
    kind, inputs = {"ModularStark": (O.AIR_MODULAR, O.modular_inputs), "Fq12Stark": (O.AIR_FQ12_MUL, O.fq12mul_inputs)}[table]
    ops, _ = inputs(rows, seed)
    trace = stark.generate_trace(ops)
    cfg = stark.config()
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    ref, _ = O.prove(kind, 0, trace, np.zeros(0, dtype=np.uint64))
    assert np.array_equal(proof.words, ref)
    if rows == 512:
        g = golden["proof_digests"][f"{'modular' if table == 'ModularStark' else 'fq12mul'}_rows512_seed{seed}"]
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(kind, 0, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    assert proof.recover_degree_bits(cfg) == rows.bit_length() - 1
    # an invalid witness (one output limb off by one) still yields a proof, which both verifiers reject
    bad = trace.copy(); bad[40 if table == "ModularStark" else 400, 3] ^= 1
    p2 = gpu.prove(stark, cfg, bad, np.zeros(0, dtype=np.uint64))
    assert O.verify(kind, 0, p2.words)[0] != 0
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, p2, cfg)


@pytest.mark.parametrize("rows,seed", [(512, 9), (4096, 19)])
def test_mystark_lookup_table_proof_bit_exact(gpu, O, rows, seed):
    """Reference `MyStark` (src/utils/lookup.rs:136-229), the unit-test table of the lookup argument, at 512 and 4096 rows:
    4 trace columns and 2 Z columns, so every Merkle leaf of the first two commitments is the row itself (hash_or_noop) --
    the narrow-matrix path of the prover.  GPU proof words == the oracle's; both verifiers accept; a row whose looked-up
    value is missing from the permuted table yields a proof both verifiers reject."""
    stark = gpu.LookupStark()
    cfg = stark.config()
    ins, tab = O.lookup_inputs(rows, seed)
    trace = stark.generate_trace(ins, tab)
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    ref, _ = O.prove(O.AIR_LOOKUP, 0, trace, np.zeros(0, dtype=np.uint64))
    assert np.array_equal(proof.words, ref)
    assert [int(x) for x in proof.words[1:6]] == [rows.bit_length() - 1, 4, 2, 4, 0]
    assert O.verify(O.AIR_LOOKUP, 0, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    r = next(i for i in range(1, rows) if trace[2, i] != trace[2, i - 1])
    bad = trace.copy()
    bad[3, r], bad[3, r + 1] = bad[3, r + 1], bad[3, r]          # still a permutation of the table, but the lookup breaks
    if bad[3, r] != trace[3, r]:
        p2 = gpu.prove(stark, cfg, bad, np.zeros(0, dtype=np.uint64))
        assert O.verify(O.AIR_LOOKUP, 0, p2.words)[0] != 0
        with pytest.raises(gpu.SbnError):
            gpu.verify_stark_proof(stark, p2, cfg)


@pytest.mark.parametrize("num_io,seed", [(16, 8), (1, 18)])
def test_flagstark_proof_bit_exact(gpu, O, num_io, seed):
    """Reference `FlagStark` (src/utils/flags.rs:379-565; 16 inputs in its test): the one table WITHOUT permutation pairs -- no
    permutation challenges are drawn, no Z commitment is made, the proof header says n_perm_zs = 0.  GPU proof words == the
    oracle's; both verifiers accept; a flipped bit column is rejected by both."""
    stark = gpu.FlagStark(num_io)
    cfg = stark.config()
    limbs, _ = O.flags_inputs(num_io, seed)
    trace = stark.generate_trace(limbs)
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    ref, _ = O.prove(O.AIR_FLAGS, num_io, trace, np.zeros(0, dtype=np.uint64))
    assert np.array_equal(proof.words, ref)
    assert [int(x) for x in proof.words[1:6]] == [(512 * num_io).bit_length() - 1, 17 + 4 * num_io, 0, 4, 0]
    assert O.verify(O.AIR_FLAGS, num_io, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    assert proof.recover_degree_bits(cfg) == (512 * num_io).bit_length() - 1
    bad = trace.copy()
    bad[5, 7] ^= 1                                               # the bit column of row 7
    p2 = gpu.prove(stark, cfg, bad, np.zeros(0, dtype=np.uint64))
    assert O.verify(O.AIR_FLAGS, num_io, p2.words)[0] != 0
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, p2, cfg)


def test_flag_u64_stark_proof_bit_exact(gpu, O):
    """The `FlagStark` of src/fields/fq12_u64/flags_u64.rs:289-447 (16 u64 exponents as in its test, :425-447): GPU proof
    words == the oracle's, no permutation-Z commitment, both verifiers accept."""
    stark = gpu.FlagU64Stark(16)
    cfg = stark.config()
    exps, _ = O.flags_u64_inputs(16, 10)
    trace = stark.generate_trace(exps)
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    ref, _ = O.prove(O.AIR_FLAGS_U64, 16, trace, np.zeros(0, dtype=np.uint64))
    assert np.array_equal(proof.words, ref)
    assert [int(x) for x in proof.words[1:6]] == [11, 71, 0, 4, 0]
    assert O.verify(O.AIR_FLAGS_U64, 16, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)


def test_g1stark_proof_without_the_times_x_step(gpu, O):
    """sbn_config.fri_variant = SBN_FRI_PLAIN (later upstream plonky2: no multiply-by-X of the FRI final polynomial):
    GPU proof bytes == the oracle's in the same mode, and differ from the default mode's."""
    pts, _ = O.g1op_inputs(1024, 3)
    stark = gpu.G1Stark()
    trace = stark.generate_trace(pts)
    cfg = stark.config()
    cfg.fri_variant = gpu.api.FRI_PLAIN
    proof = gpu.prove(stark, cfg, trace, np.zeros(0, dtype=np.uint64))
    try:
        O.set_final_poly_times_x(False)
        ref, _ = O.prove(O.AIR_G1_OP, 0, trace, np.zeros(0, dtype=np.uint64))
    finally:
        O.set_final_poly_times_x(True)
    assert np.array_equal(proof.words, ref)
    gpu.verify_stark_proof(stark, proof, cfg)
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, proof, stark.config())


@pytest.fixture(scope="module")
def g1exp_gpu_proof(gpu, g1exp_case):
    stark = gpu.G1ExpStark(128)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 16)
    prover.load_trace(g1exp_case["trace"], g1exp_case["pi"])
    p1 = prover.prove()
    p2 = prover.prove()
    times = prover.stage_times()
    prover.close()
    return stark, cfg, p1, p2, times


def test_g1exp_proof_matches_oracle_digest(g1exp_gpu_proof, golden):
    """BASELINE config[1]: G1ExpStark(128), 2^16 rows: GPU proof bytes == the CPU oracle's (committed sha256)."""
    _, _, p1, p2, times = g1exp_gpu_proof
    g = golden["proof_digests"]["g1exp_io128_seed1"]
    assert len(p1.words) == g["proof_words"]
    assert [int(x) for x in p1.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(p1.to_bytes()).hexdigest() == g["proof_sha256"]
    assert np.array_equal(p1.words, p2.words)                      # deterministic (smallest PoW witness)
    assert set(times) >= {"trace_commit", "trace_absorb_kernels_ms", "quotient_eval", "fri_layers"} and all(v >= 0 for v in times.values())


def test_g1exp_proof_verifies_and_tamper_is_rejected(gpu, O, g1exp_gpu_proof):
    stark, cfg, p1, _, _ = g1exp_gpu_proof
    assert O.verify(O.AIR_G1_EXP, 128, p1.words) == (0, "")        # independent verifier (oracle)
    gpu.verify_stark_proof(stark, p1, cfg)                          # product verifier
    w = p1.words
    ncol, nz = 1676, 762
    openings = 12 + 3 * 64
    for idx in (13, openings + 3, openings + 2 * ncol + 1, len(w) - 7168 - 1, len(w) - 1):
        t = w.copy()
        t[idx] = (int(t[idx]) + 1) % P
        assert O.verify(O.AIR_G1_EXP, 128, t)[0] != 0
        with pytest.raises(gpu.SbnError):
            gpu.verify_stark_proof(stark, gpu.Proof(t, 16), cfg)


def test_g1exp_full_oracle_proof_equality(gpu, O, g1exp_case, g1exp_gpu_proof):
    """Runs the CPU oracle's prove() on the same trace on the box (about a minute) and compares every word."""
    _, _, p1, _, _ = g1exp_gpu_proof
    ref, _ = O.prove(O.AIR_G1_EXP, 128, g1exp_case["trace"], g1exp_case["pi"])
    assert np.array_equal(p1.words, ref)


@pytest.mark.parametrize("seed", [11, 12])
def test_g1exp_more_seeds_full_oracle_proof_equality(gpu, O, seed):
    """More seeds of BASELINE config[1] (with seed 1 above: three in all), the witness generated on the device: trace,
    public inputs and every proof word equal the CPU oracle's (its prove() runs on the box, about a minute per seed)."""
    ios, _ = O.g1exp_inputs(128, seed)
    trace, pi = O.g1exp_trace(ios)
    stark = gpu.G1ExpStark(128)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 16)
    try:
        assert np.array_equal(prover.generate_trace(ios), pi)
        got = prover.prove()
    finally:
        prover.close()
    want, _ = O.prove(O.AIR_G1_EXP, 128, trace, pi)
    assert np.array_equal(got.words, want)
    gpu.verify_stark_proof(stark, got, cfg)


@pytest.mark.parametrize("table,num_io,seed", [("g2", 128, 22), ("fq12", 16, 23), ("fq", 128, 24)])
def test_other_exp_tables_second_seed_full_oracle_proof_equality(gpu, O, table, num_io, seed):
    """A second seed (beside the committed digests) for G2ExpStark(128), Fq12ExpStark(16) and FqExpStark(128), the witness
    generated on the device: public inputs and every proof word equal the CPU oracle's own prove() of its own trace."""
    kind, inputs, gen, cls, bits = {
        "g2": (O.AIR_G2_EXP, O.g2exp_inputs, O.g2exp_trace, gpu.G2ExpStark, 16),
        "fq12": (O.AIR_FQ12_EXP, O.fq12exp_inputs, O.fq12exp_trace, gpu.Fq12ExpStark, 13),
        "fq": (O.AIR_FQ_EXP, O.fqexp_inputs, O.fqexp_trace, gpu.FqExpStark, 16)}[table]
    ios, _ = inputs(num_io, seed)
    trace, pi = gen(ios)
    stark = cls(num_io)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, bits)
    try:
        assert np.array_equal(prover.generate_trace(ios), pi)
        got = prover.prove()
    finally:
        prover.close()
    want, _ = O.prove(kind, num_io, trace, pi)
    assert np.array_equal(got.words, want)
    gpu.verify_stark_proof(stark, got, cfg)


def _check_every_chain_placement(gpu, stark, cfg, bits, ios, pi_want, trace_want):
    """The curve chains of the device witness (src/curves/g1/exp.rs:255-318) in every placement the library can pick
    (SBN_TRACEGEN_DEVICE_CHAIN, read when a prover is created): 0 = host pool (eight instances per AVX-512 IFMA register, or the
    scalar form with SBN_NO_AVX512-less CPUs), 1 = one lane per instance (chain_kernel), 2 = one wave per instance
    (chain_coop_kernel, what a rank with a small host share gets), and the old one-pass range-check kernel: same trace words."""
    for env in ({"SBN_TRACEGEN_DEVICE_CHAIN": "0"}, {"SBN_TRACEGEN_DEVICE_CHAIN": "1"}, {"SBN_TRACEGEN_DEVICE_CHAIN": "2"},
                {"SBN_EXPERIMENTAL": "1", "SBN_RANGE_CHECK": "1"}):
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            pr = gpu.Prover(stark, cfg, bits)
        finally:
            for k, v in old.items():
                if v is None:
                    del os.environ[k]
                else:
                    os.environ[k] = v
        try:
            d = pr.describe()
            if "SBN_TRACEGEN_DEVICE_CHAIN" in env:
                assert d["curve_chains"].startswith({"0": "host_pool", "1": "device_lane", "2": "device_wave"}[env["SBN_TRACEGEN_DEVICE_CHAIN"]]), d
            else:
                assert d["range_check"] == "1" and d["experimental"] == "1", d
            assert np.array_equal(pr.generate_trace(ios), pi_want), env
            got = pr.read_trace()
            bad = np.nonzero((got != trace_want).any(axis=1))[0]
            assert bad.size == 0, (env, bad[:8].tolist())
        finally:
            pr.close()


def test_g1exp_device_witness_generation_matches_oracle(gpu, O, g1exp_case, g1exp_gpu_proof, golden):
    """G1ExpStark::generate_trace on the device (src/curves/g1/exp.rs:255-327): trace and public inputs equal the CPU
    oracle's word for word, and proving straight from the device-resident trace gives the same proof bytes."""
    stark, cfg, p1, _, _ = g1exp_gpu_proof
    prover = gpu.Prover(stark, cfg, 16)
    try:
        pi = prover.generate_trace(g1exp_case["ios"])
        assert np.array_equal(pi, g1exp_case["pi"])
        dev = prover.read_trace()
        ref = g1exp_case["trace"]
        bad = np.nonzero((dev != ref).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        proof = prover.prove()
        assert np.array_equal(proof.words, p1.words)
        assert prover.stage_times()["device_tracegen_ms"] > 0
        # structured columns: a constant limb column, one repeating 65535, and a degenerate instance
        ios = g1exp_case["ios"].copy()
        ios[:, 32:40] = 0                                          # exp_val = 0: no additions at all
        ios[3, 32:40] = 0xFFFFFFFF
        pi2 = prover.generate_trace(ios)
        t_host, pi_host = stark.generate_trace_and_public_inputs(ios)
        assert np.array_equal(pi2, pi_host)
        assert np.array_equal(prover.read_trace(), t_host)
        _check_every_chain_placement(gpu, stark, cfg, 16, ios, pi_host, t_host)
        ios[5, 16:32] = ios[5, 0:16]                               # offset == x with bit 0 set: x1 == x2 in the first add
        ios[5, 32] = 1
        with pytest.raises(gpu.SbnError) as e:
            prover.generate_trace(ios)
        assert e.value.code == -8
        with pytest.raises(gpu.SbnError) as e:
            stark.generate_trace_and_public_inputs(ios)
        assert e.value.code == -8
    finally:
        prover.close()


@pytest.mark.parametrize("table,num_io", [("g1", 256), ("fq", 512), ("g2", 256)])
def test_device_witness_above_2pow16_rows(gpu, O, table, num_io):
    """The reference pads the instance list to any power of two >= 128 (src/curves/g1/circuit.rs:273-277), i.e. tables of
    2^17, 2^18 ... rows, where the multiplicities of the u16 range check no longer fit 16-bit counters (the table column then
    ends in n - 65535 copies of 65535).  Device witness == the product's host generator, word for word, on random instances
    and on instances whose range-checked columns are mostly one value (multiplicity > 2^16); the proof verifies."""
    if table == "g1":
        ios, _ = O.g1exp_inputs(num_io, 21)
        stark = gpu.G1ExpStark(num_io)
        e0 = 32
    elif table == "g2":                                      # include/sbn.h: G2_EXP device witness covers 2^16 .. 2^18 rows
        ios, _ = O.g2exp_inputs(num_io, 23)
        stark = gpu.G2ExpStark(num_io)
        e0 = 64
    else:
        ios, _ = O.fqexp_inputs(num_io, 22)
        stark = gpu.FqExpStark(num_io)
        e0 = 16
    cfg = stark.config()
    bits = (512 * num_io).bit_length() - 1
    prover = gpu.Prover(stark, cfg, bits)
    try:
        for variant in range(2):
            if variant == 1:
                ios = ios.copy()
                ios[:, e0:e0 + 8] = 0                          # exponent 0: no additions / multiplications at all
                ios[3, e0:e0 + 8] = 0xFFFFFFFF
            pi = prover.generate_trace(ios)
            t_host, pi_host = stark.generate_trace_and_public_inputs(ios)
            assert np.array_equal(pi, pi_host)
            dev = prover.read_trace()
            bad = np.nonzero((dev != t_host).any(axis=1))[0]
            assert bad.size == 0, f"variant {variant}: first differing columns: {bad[:8].tolist()}"
            del dev, t_host
        proof = prover.prove()
        gpu.verify_stark_proof(stark, proof, cfg)
        assert proof.recover_degree_bits(cfg) == bits
    finally:
        prover.close()


def test_load_trace_device_and_trace_device_ptr(gpu, O, g1exp_case):
    """sbn_prover_trace_device_ptr / sbn_prover_load_trace_device (include/sbn.h): a trace that is already resident in HBM --
    here the witness one prover generated on the device -- is handed to a second prover by device pointer (device-to-device
    copy, nothing crosses PCIe); a caller may also fill the prover's own trace buffer in place and declare it loaded.  Both
    give the proof of the host-loaded trace."""
    stark = gpu.G1ExpStark(128)
    cfg = stark.config()
    a = gpu.Prover(stark, cfg, 16)
    b = gpu.Prover(stark, cfg, 16)
    try:
        pi = a.generate_trace(g1exp_case["ios"])
        assert np.array_equal(pi, g1exp_case["pi"])
        ref = a.prove()
        ptr = a.trace_device_ptr()
        assert ptr
        b.load_trace_device(ptr, pi)                         # another prover's buffer: copied device to device
        assert np.array_equal(b.read_trace(), g1exp_case["trace"])
        assert np.array_equal(b.prove().words, ref.words)
        b.load_trace_device(b.trace_device_ptr(), pi)        # the prover's own buffer, filled in place: no copy at all
        assert np.array_equal(b.prove().words, ref.words)
        with pytest.raises(gpu.SbnError):
            b.load_trace_device(ptr, pi[:-1])                # wrong number of public inputs
        bad = pi.copy()
        bad[0] = P                                           # not canonical
        with pytest.raises(gpu.SbnError) as e:
            b.load_trace_device(ptr, bad)
        assert e.value.code == -2
    finally:
        a.close()
        b.close()


def test_g1exp_2pow17_rows_full_oracle_proof_equality(gpu, O):
    """G1ExpStark(256): 2^17 rows, where the transforms take the 512 x 256 / 512 x 512 passes instead of 256 x 256 / 512 x 256
    and the range-check table column ends in copies of 65535.  Device witness == the oracle's trace, and every proof word
    == the oracle's prove() of it (about half a minute on the box)."""
    ios, _ = O.g1exp_inputs(256, 31)
    trace, pi = O.g1exp_trace(ios)
    stark = gpu.G1ExpStark(256)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 17)
    try:
        assert np.array_equal(prover.generate_trace(ios), pi)
        dev = prover.read_trace()
        bad = np.nonzero((dev != trace).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        del dev
        got = prover.prove()
    finally:
        prover.close()
    want, _ = O.prove(O.AIR_G1_EXP, 256, trace, pi)
    assert np.array_equal(got.words, want)
    assert O.verify(O.AIR_G1_EXP, 256, got.words) == (0, "")


def test_prover_argument_errors(gpu, g1op_case):
    stark = gpu.G1Stark()
    cfg = stark.config()
    tr = g1op_case["trace"].copy()
    tr[5, 7] = P                                                    # non-canonical trace word
    with pytest.raises(gpu.SbnError) as e:
        gpu.prove(stark, cfg, tr, g1op_case["pi"])
    assert e.value.code == -2
    with pytest.raises(gpu.SbnError) as e:                          # wrong public-input count
        gpu.prove(stark, cfg, g1op_case["trace"], np.zeros(3, dtype=np.uint64))
    assert e.value.code == -1
    with pytest.raises(gpu.SbnError) as e:                          # G1ExpStark below 2^16 rows
        gpu.Prover(gpu.G1ExpStark(4), cfg, 11)
    assert e.value.code in (-1, -7)
    with pytest.raises(gpu.SbnError):                               # prove before load
        gpu.Prover(stark, cfg, 9).prove()


def test_invalid_witness_yields_rejected_proof(gpu, O, g1op_case):
    tr = g1op_case["trace"].copy()
    tr[64 + 16, 9] = (int(tr[64 + 16, 9]) + 1) % 65536
    stark = gpu.G1Stark()
    proof = gpu.prove(stark, stark.config(), tr, g1op_case["pi"])
    ref, _ = O.prove(O.AIR_G1_OP, 0, tr, g1op_case["pi"])
    assert np.array_equal(proof.words, ref)                         # still the same bytes as the oracle...
    assert O.verify(O.AIR_G1_OP, 0, proof.words)[0] != 0           # ...and both verifiers reject them
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, proof, stark.config())


def test_g2exp_proof_matches_oracle_digest_and_verifies(gpu, O, g2exp_case, golden):
    """BASELINE config[3]: G2ExpStark(128), 2^16 rows x 2822 columns: GPU proof bytes == the CPU oracle's
    (committed sha256; the oracle run takes minutes), accepted by the oracle's and the product's verifier,
    tampering rejected."""
    stark = gpu.G2ExpStark(128)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 16)
    prover.load_trace(g2exp_case["trace"], g2exp_case["pi"])
    proof = prover.prove()
    prover.close()
    g = golden["proof_digests"]["g2exp_io128_seed2"]
    assert len(proof.words) == g["proof_words"]
    assert [int(x) for x in proof.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(O.AIR_G2_EXP, 128, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    t = proof.words.copy()
    t[12 + 3 * 64 + 9] = (int(t[12 + 3 * 64 + 9]) + 1) % P
    assert O.verify(O.AIR_G2_EXP, 128, t)[0] != 0
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, gpu.Proof(t, 16), cfg)


def test_g2exp_device_witness_generation_matches_oracle(gpu, O, g2exp_case, golden):
    """G2ExpStark::generate_trace on the device (src/curves/g2/exp.rs:271-342): trace and public inputs equal the CPU
    oracle's word for word; proving straight from the device-resident trace gives the committed proof digest."""
    stark = gpu.G2ExpStark(128)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 16)
    try:
        pi = prover.generate_trace(g2exp_case["ios"])
        assert np.array_equal(pi, g2exp_case["pi"])
        dev = prover.read_trace()
        bad = np.nonzero((dev != g2exp_case["trace"]).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        proof = prover.prove()
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == golden["proof_digests"]["g2exp_io128_seed2"]["proof_sha256"]
        ios = g2exp_case["ios"].copy()
        ios[:, 64:72] = 0                                          # exp_val = 0 everywhere but one instance
        ios[7, 64:72] = 0xFFFFFFFF
        pi2 = prover.generate_trace(ios)
        t_host, pi_host = stark.generate_trace_and_public_inputs(ios)
        assert np.array_equal(pi2, pi_host)
        assert np.array_equal(prover.read_trace(), t_host)
        _check_every_chain_placement(gpu, stark, cfg, 16, ios, pi_host, t_host)
        ios[5, 32:64] = ios[5, 0:32]                               # offset == x with bit 0 set: x1 == x2 in the first add
        ios[5, 64] = 1
        with pytest.raises(gpu.SbnError) as e:
            prover.generate_trace(ios)
        assert e.value.code == -8
    finally:
        prover.close()


def test_fq12exp_proof_matches_oracle_digest_and_verifies(gpu, O, fq12exp_case, golden):
    """Fq12ExpStark(16) (the reference's test_fq12_exp_raw size, 2^13 rows x 9802 columns, split range check):
    GPU proof bytes == the CPU oracle's (committed sha256), both verifiers accept, tampering rejected."""
    stark = gpu.Fq12ExpStark(16)
    cfg = stark.config()
    proof = gpu.prove(stark, cfg, fq12exp_case["trace"], fq12exp_case["pi"])
    g = golden["proof_digests"]["fq12exp_io16_seed3"]
    assert len(proof.words) == g["proof_words"]
    assert [int(x) for x in proof.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(O.AIR_FQ12_EXP, 16, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    t = proof.words.copy()
    t[12 + 3 * 64 + 11] = (int(t[12 + 3 * 64 + 11]) + 1) % P
    assert O.verify(O.AIR_FQ12_EXP, 16, t)[0] != 0
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, gpu.Proof(t, 13), cfg)


def _assert_stages_equal_kit(O, words, key):
    """The device proof against the committed ORACLE-PROVER run of the same instance list (tools/oracle_digest.py, run once on a
    GPU box: Fq12ExpStark(128) 27 s / 44 GB, Fq12ExpStark(512) 178 s / 167 GB on 16 threads): proof sha256 and every stage of
    prove() in order -- a mismatch names the first stage that differs."""
    import parity_kit as K
    kit = json.load(open(os.path.join(GOLD, "parity_kit", key + ".json")))
    got = json.loads(json.dumps(K.stage_digests(words, O.poseidon_permute)))
    assert K.first_difference(got, kit["stages"]["times_x"]) is None, (key, K.first_difference(got, kit["stages"]["times_x"]))


def test_fq12exp_2pow16_independent_verifier(gpu, O, golden):
    """Fq12ExpStark(128) (2^16 rows x 10250 columns, 74752 public inputs): GPU proof bytes == the CPU oracle PROVER's
    (committed sha256 and per-stage digests, tests/golden/proof_digests.json fq12exp_io128_seed11), the CPU
    oracle's VERIFIER (independent code) accepts the GPU proof, outputs equal offset * x^e computed with python
    tower arithmetic, and a flipped opening is rejected."""
    num_io = 128
    ios, native = O.fq12exp_inputs(num_io, 11)
    stark = gpu.Fq12ExpStark(num_io)
    cfg = stark.config()
    trace, pi = stark.generate_trace_and_public_inputs(ios)
    for k in (0, 127):
        x, off, e = native[k]
        got = [sum(int(pi[584 * k + 392 + 16 * c + i]) << (16 * i) for i in range(16)) for c in range(12)]
        assert got == O.fq12_mul(off, O.fq12_pow(x, e))
    proof = gpu.prove(stark, cfg, trace, pi)
    g = golden["proof_digests"]["fq12exp_io128_seed11"]
    assert hashlib.sha256(pi.tobytes()).hexdigest() == g["pi_sha256"]
    assert len(proof.words) == g["proof_words"] and [int(x) for x in proof.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    _assert_stages_equal_kit(O, proof.words, "fq12exp_io128_seed11")
    assert O.verify(O.AIR_FQ12_EXP, num_io, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    t = proof.words.copy()
    t[12 + 3 * 64 + 5] = (int(t[12 + 3 * 64 + 5]) + 1) % P
    assert O.verify(O.AIR_FQ12_EXP, num_io, t)[0] != 0


def test_fq12exp_2pow18_config4(gpu, O, golden):
    """BASELINE config[4] on one GPU: Fq12ExpStark(512), 2^18 rows x 11786 columns (24.7 GB trace, never on the host:
    the witness is generated on the device), reference workload src/fields/fq12/exp.rs:638-696.  GPU proof bytes == the CPU
    oracle PROVER's proof of the same 512 instances (committed sha256 and per-stage digests: tests/golden/proof_digests.json
    fq12exp_io512_seed3, tests/golden/parity_kit/fq12exp_io512_seed3.json, generated by tools/oracle_digest.py), plus the
    size-independent checks: outputs equal offset * x^e by python tower arithmetic, the oracle's VERIFIER (independent code)
    and the product's accept the proof, a flipped opening / cap / public input is rejected, and a second prove() is
    word-identical."""
    num_io = 512
    ios, native = O.fq12exp_inputs(num_io, 3)
    stark = gpu.Fq12ExpStark(num_io)
    cfg = stark.config()
    assert (stark.num_columns, stark.num_public_inputs, stark.num_permutation_zs(cfg)) == (11786, 299008, 5328)
    prover = gpu.Prover(stark, cfg, 18)
    try:
        pi = prover.generate_trace(ios)
        for k in (0, 257, 511):
            x, off, e = native[k]
            got = [sum(int(pi[584 * k + 392 + 16 * c + i]) << (16 * i) for i in range(16)) for c in range(12)]
            assert got == O.fq12_mul(off, O.fq12_pow(x, e))
        p1 = prover.prove()
        p2 = prover.prove()
        times = prover.stage_times()
    finally:
        prover.close()
    assert np.array_equal(p1.words, p2.words)
    g = golden["proof_digests"]["fq12exp_io512_seed3"]
    assert hashlib.sha256(np.asarray(pi, dtype=np.uint64).tobytes()).hexdigest() == g["pi_sha256"]
    assert len(p1.words) == g["proof_words"] and [int(x) for x in p1.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(p1.to_bytes()).hexdigest() == g["proof_sha256"]
    _assert_stages_equal_kit(O, p1.words, "fq12exp_io512_seed3")
    assert p1.recover_degree_bits(cfg) == 18
    assert np.array_equal(p1.words[-299008:], pi)
    assert O.verify(O.AIR_FQ12_EXP, num_io, p1.words) == (0, "")
    gpu.verify_stark_proof(stark, p1, cfg)
    for idx in (13, 12 + 3 * 64 + 5, len(p1.words) - 299008 + 700):      # trace cap, an opening, a public input
        t = p1.words.copy()
        t[idx] = (int(t[idx]) + 1) % P
        assert O.verify(O.AIR_FQ12_EXP, num_io, t)[0] != 0
        with pytest.raises(gpu.SbnError):
            gpu.verify_stark_proof(stark, gpu.Proof(t, 18), cfg)
    print("Fq12ExpStark(512) stage times (ms):", {k: round(v, 1) for k, v in times.items()})


def test_fq12exp_device_witness_generation_matches_oracle(gpu, O, fq12exp_case, golden):
    """Fq12ExpStark::generate_trace on the device (src/fields/fq12/exp.rs:283-319, split range check): trace and public
    inputs equal the CPU oracle's word for word; proving from the device-resident trace gives the committed digest."""
    stark = gpu.Fq12ExpStark(16)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 13)
    try:
        pi = prover.generate_trace(fq12exp_case["ios"])
        assert np.array_equal(pi, fq12exp_case["pi"])
        dev = prover.read_trace()
        bad = np.nonzero((dev != fq12exp_case["trace"]).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        proof = prover.prove()
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == golden["proof_digests"]["fq12exp_io16_seed3"]["proof_sha256"]
        ios = fq12exp_case["ios"].copy()
        ios[:, 192:200] = 0                                        # exponent 0 everywhere but two instances
        ios[3, 192:200] = 0xFFFFFFFF
        ios[9, 192] = 5
        ios[11, 0:192] = 0                                         # x = offset = 0: every product is 0 (constant byte columns)
        pi2 = prover.generate_trace(ios)
        t_host, pi_host = stark.generate_trace_and_public_inputs(ios)
        assert np.array_equal(pi2, pi_host)
        assert np.array_equal(prover.read_trace(), t_host)
    finally:
        prover.close()


def test_fqexp_proof_and_device_witness_match_oracle(gpu, O, fqexp_case, golden):
    """FqExpStark(128), 2^16 rows x 960 columns (the reference's test_fq_exp_raw): GPU proof bytes == the CPU oracle's
    (committed sha256), both verifiers accept, tampering rejected; the device-generated witness equals the oracle's and
    proves to the same bytes."""
    stark = gpu.FqExpStark(128)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 16)
    try:
        prover.load_trace(fqexp_case["trace"], fqexp_case["pi"])
        proof = prover.prove()
        g = golden["proof_digests"]["fqexp_io128_seed4"]
        assert len(proof.words) == g["proof_words"]
        assert [int(x) for x in proof.words[12:16]] == g["trace_cap0"]
        assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
        assert O.verify(O.AIR_FQ_EXP, 128, proof.words) == (0, "")
        gpu.verify_stark_proof(stark, proof, cfg)
        t = proof.words.copy()
        t[12 + 3 * 64 + 9] = (int(t[12 + 3 * 64 + 9]) + 1) % P
        assert O.verify(O.AIR_FQ_EXP, 128, t)[0] != 0
        with pytest.raises(gpu.SbnError):
            gpu.verify_stark_proof(stark, gpu.Proof(t, 16), cfg)
        pi = prover.generate_trace(fqexp_case["ios"])
        assert np.array_equal(pi, fqexp_case["pi"])
        bad = np.nonzero((prover.read_trace() != fqexp_case["trace"]).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        assert np.array_equal(prover.prove().words, proof.words)
    finally:
        prover.close()


def test_fq12expu64_proof_matches_oracle_digest_and_verifies(gpu, O, fq12expu64_case, golden):
    """Fq12ExpU64Stark(16), 2^11 rows x 9792 columns (the reference's test_fq12_exp_u64_raw): GPU proof bytes == the CPU
    oracle's (committed sha256), both verifiers accept, tampering rejected."""
    stark = gpu.Fq12ExpU64Stark(16)
    cfg = stark.config()
    prover = gpu.Prover(stark, cfg, 11)
    try:
        prover.load_trace(fq12expu64_case["trace"], fq12expu64_case["pi"])
        proof = prover.prove()
        # the same instances with the witness generated on the device
        assert np.array_equal(prover.generate_trace(fq12expu64_case["ios"]), fq12expu64_case["pi"])
        bad = np.nonzero((prover.read_trace() != fq12expu64_case["trace"]).any(axis=1))[0]
        assert bad.size == 0, f"first differing columns: {bad[:8].tolist()}"
        assert np.array_equal(prover.prove().words, proof.words)
    finally:
        prover.close()
    g = golden["proof_digests"]["fq12expu64_io16_seed5"]
    assert len(proof.words) == g["proof_words"]
    assert [int(x) for x in proof.words[12:16]] == g["trace_cap0"]
    assert hashlib.sha256(proof.to_bytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(O.AIR_FQ12_EXP_U64, 16, proof.words) == (0, "")
    gpu.verify_stark_proof(stark, proof, cfg)
    assert proof.recover_degree_bits(cfg) == 11
    t = proof.words.copy()
    t[12 + 3 * 64 + 9] = (int(t[12 + 3 * 64 + 9]) + 1) % P
    assert O.verify(O.AIR_FQ12_EXP_U64, 16, t)[0] != 0
    with pytest.raises(gpu.SbnError):
        gpu.verify_stark_proof(stark, gpu.Proof(t, 11), cfg)


def test_batch_prover_matches_single_proofs(gpu, g1exp_case, g1exp_gpu_proof):
    """sbn_batch_prover_* (batch mode, 3 proofs in flight, witness generated on the device): every unit's proof equals
    the proof of the same instance list proved alone; a failing unit fails the batch and frees everything."""
    stark, cfg, p1, _, _ = g1exp_gpu_proof
    ios = g1exp_case["ios"]
    other = ios.copy()
    other[:, 32:40] = np.random.default_rng(3).integers(0, 1 << 32, size=(128, 8), dtype=np.uint64).astype(np.uint32)
    bp = gpu.BatchProver(stark, cfg, 16, 3)
    try:
        proofs = bp.prove_ios(np.stack([ios, other, ios, other, ios]))
        assert len(proofs) == 5
        for k in (0, 2, 4):
            assert np.array_equal(proofs[k].words, p1.words)
        assert np.array_equal(proofs[1].words, proofs[3].words) and not np.array_equal(proofs[1].words, p1.words)
        gpu.verify_stark_proof(stark, proofs[1], cfg)
        bad = ios.copy()
        bad[5, 16:32] = bad[5, 0:16]
        bad[5, 32] |= 1                                            # x1 == x2 in the first addition of instance 5
        with pytest.raises(gpu.SbnError) as e:
            bp.prove_ios(np.stack([ios, bad, ios]))
        assert e.value.code == -8
    finally:
        bp.close()

