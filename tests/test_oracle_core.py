"""CPU tests of the oracle against the golden vectors (Poseidon KATs, field, BN254 witness arithmetic)."""
import numpy as np

P = 0xFFFFFFFF00000001


def test_goldilocks_constants(O):
    assert pow(7, (P - 1) >> 32, P) == 1753635133440165772        # POWER_OF_TWO_GENERATOR = 7^((p-1)/2^32)
    assert pow(1753635133440165772, 1 << 31, P) == P - 1           # order exactly 2^32
    assert pow(65536, -1, P) == 18446462594437939201               # addcy.rs:13
    rng = np.random.default_rng(0)
    for _ in range(200):
        a, b = int(rng.integers(0, P, dtype=np.uint64)), int(rng.integers(0, P, dtype=np.uint64))
        assert O.lib().orc_gf_mul(a, b) == a * b % P
    for a in (1, 2, P - 1, 0xFFFFFFFF, 1 << 63):
        assert O.lib().orc_gf_inv(a) == pow(a, -1, P)
    # edge products around the reduction carries
    for a in (P - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1 << 32):
        for b in (P - 1, 0xFFFFFFFF00000000, 0xFFFFFFFF, 1 << 32):
            assert O.lib().orc_gf_mul(a, b) == a * b % P


def test_poseidon_round_constants_and_kats(O, golden):
    kat = golden["poseidon_kat"]
    rc = O.round_constants()
    assert [hex(x) for x in rc[:16]] == [hex(int(x, 16)) for x in kat["round_constants_first16"]]
    assert len(rc) == 360 and max(rc) < 0xfffeeac900011537       # plonky2's documented bound for the AVX2 path
    for v in kat["vectors"]:
        out = O.poseidon_permute([int(x, 16) for x in v["input"]])
        assert out == [int(x, 16) for x in v["output"]]


def test_poseidon_fast_form_equals_definition(O, golden):
    """The oracle hashes Merkle leaves with the sparse-partial-round form of the permutation; it must be the same map as
    the plain definition (which the KATs pin), on the KAT inputs, edge values and random states."""
    import random
    rnd = random.Random(7)
    P = 0xFFFFFFFF00000001
    states = [[int(x, 16) for x in v["input"]] for v in golden["poseidon_kat"]["vectors"]]
    states += [[P - 1] * 12, [0] * 11 + [1], [0xFFFFFFFF] * 12, [0xFFFFFFFF00000000] * 12]
    states += [[rnd.randrange(P) for _ in range(12)] for _ in range(200)]
    for st in states:
        assert O.poseidon_permute_fast(st) == O.poseidon_permute(st)


def test_poseidon_constants_independent_derivation(O):
    """tools/_chacha8.py (python) and oracle/poseidon.hpp (C++) derive the same 360 constants."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from _chacha8 import ChaChaRng
    g = ChaChaRng(0, 8)
    assert [g.gen_range(P) for _ in range(360)] == O.round_constants()


def test_sponge_and_challenger(O):
    # hash_or_noop: <= 4 elements are returned zero-padded without a permutation
    assert O.hash_no_pad([5, 6]) == [5, 6, 0, 0]
    # 8 elements = one overwrite + permute
    assert O.hash_no_pad(list(range(1, 9))) == O.poseidon_permute(list(range(1, 9)) + [0, 0, 0, 0])[:4]
    # 9 elements: second block overwrites lane 0 only
    s1 = O.poseidon_permute(list(range(1, 9)) + [0, 0, 0, 0])
    s1[0] = 9
    assert O.hash_no_pad(list(range(1, 10))) == O.poseidon_permute(s1)[:4]
    # challenger: outputs are popped from the back of state[0..8] after a duplex
    st = O.poseidon_permute([11, 22, 33] + [0] * 9)
    assert O.challenger_probe([11, 22, 33], 3) == [st[7], st[6], st[5]]
    # observing exactly 8 elements duplexes immediately; the first challenge is then state[7]
    st = O.poseidon_permute(list(range(8)) + [0] * 4)
    assert O.challenger_probe(list(range(8)), 1) == [st[7]]


def test_fq_arithmetic(O):
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = int.from_bytes(rng.bytes(32), "little") % O.BN_P
        b = int.from_bytes(rng.bytes(32), "little") % O.BN_P
        assert O.fq_mul(a, b) == a * b % O.BN_P
        if a:
            assert O.fq_inv(a) == pow(a, -1, O.BN_P)
    assert O.fq_mul(O.BN_P - 1, O.BN_P - 1) == 1


def test_modular_witness_kats(O, golden):
    for case in golden["modular_kat"]["cases"]:
        got = O.modular_witness(case["pol_input"], case["zero_mode"])
        exp = case["expect"]
        for k in ("quot_abs", "sign", "aux_lo", "aux_hi"):
            assert got[k] == exp[k], k
        if not case["zero_mode"]:
            assert got["output"] == exp["output"] and got["out_aux_red"] == exp["out_aux_red"]
        assert all(0 <= v < 65536 for v in got["aux_lo"] + got["aux_hi"] + got["quot_abs"])


def test_flags_native(O, golden):
    """flags.rs:334-369: bits filtered by (a == 0 rows) reproduce the exponent LSB-first."""
    f = golden["flags_native"]
    t = O.flags_table(f["limbs"])
    assert t.shape == (512, 14)
    bits = [int(t[r, 5]) for r in range(0, 512, 2)]     # `bit` column on the add rows
    assert bits == f["bits_lsb_first"]
    assert int(t[511, 0]) == 1 and int(t[:511, 0].sum()) == 0      # is_final only on the last row
    assert [r for r in range(512) if t[r, 1] == 1] == [62 + 64 * k for k in range(8)]   # rotate rows


def test_lookup_fixed_input(O, golden):
    g = golden["lookup_fixed"]
    so, pt = O.permuted_cols(g["inputs"], g["table"])
    assert so == g["sorted_inputs"] and pt == g["permuted_table"]
    # Halo2 lookup invariant: each row either repeats the previous input or equals the permuted table
    for i in range(len(so)):
        assert so[i] == pt[i] or (i > 0 and so[i] == so[i - 1])
    assert sorted(pt) == sorted(g["table"])


def test_g1_scalar_mult_against_python(O, golden):
    """The G1ExpStark witness ends in b = x*s + offset (exp.rs:283-285) -- checked with python ints.
    Uses a 4-instance table, which is below the 2^16-row prover minimum but fine for witness generation
    of the main columns: the oracle asserts rows >= 65536 only in the range-check generator, so the
    check goes through the public inputs of a full 128-instance trace instead."""
    case = golden["g1_scalar_mult"]["cases"][0]
    ios = np.array(case["ios"], dtype=np.uint32)
    full, _ = O.g1exp_inputs(128, case["seed"])
    assert np.array_equal(full[:4], ios)            # same seeded stream
    trace, pi = O.g1exp_trace(full)
    for k, (ox, oy) in enumerate(case["outputs"]):
        gx = sum(int(pi[56 * k + 40 + i]) << (32 * i) for i in range(8))
        gy = sum(int(pi[56 * k + 48 + i]) << (32 * i) for i in range(8))
        assert (gx, gy) == (int(ox, 16), int(oy, 16))
    # last-row accumulator b equals the output limbs (16-bit columns 32..63)
    k = 3
    row = 512 * k + 511
    bx = sum(int(trace[32 + i, row]) << (16 * i) for i in range(16))
    assert bx == int(case["outputs"][k][0], 16)


def test_avx512_eight_lane_permutation_equals_the_definition(O):
    """oracle/poseidon_x8.hpp (eight permutations per AVX-512 register, the leaf hashing of the CPU baseline) == the plain
    definition on edge values and random states; the Merkle caps of a commitment are the same with it and without it
    (ORC_NO_AVX512=1 in a child process).  Skipped on CPUs without AVX-512F/DQ, where the scalar path is the only one."""
    import os
    import subprocess
    import sys
    P = O.GL_P
    rng = np.random.default_rng(77)
    st = rng.integers(0, P, size=(8, 12), dtype=np.uint64)
    st[0] = P - 1
    st[1] = 0
    st[2, :6] = [P - 1, 0, 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 63]
    a = st.copy()
    if O.lib().orc_poseidon_permute_x8(O.ptr(a)) != 0:
        pytest.skip("no AVX-512F/DQ on this CPU")
    for j in range(8):
        assert [int(x) for x in a[j]] == O.poseidon_permute([int(x) for x in st[j]]), j
    cols = rng.integers(0, P, size=(21, 1024), dtype=np.uint64)        # 21 columns: a ragged last sponge block
    cap, _, _ = O.commit_values(cols)
    code = ("import sys, numpy as np; sys.path.insert(0, %r); import oracle_lib as O; "
            "rng = np.random.default_rng(77); rng.integers(0, O.GL_P, size=(8, 12), dtype=np.uint64); "
            "cols = rng.integers(0, O.GL_P, size=(21, 1024), dtype=np.uint64); print(O.commit_values(cols)[0].tolist())") % os.path.dirname(os.path.abspath(__file__))
    out = subprocess.check_output([sys.executable, "-c", code], env=dict(os.environ, ORC_NO_AVX512="1"))
    assert eval(out.decode().strip().splitlines()[-1]) == cap.tolist()
