"""CPU tests: AIR constraints vanish on generated traces; oracle prove -> verify; tamper rejection;
committed proof digests."""
import hashlib

import numpy as np
import pytest

P = 0xFFFFFFFF00000001


def _trace_domain_consumer_args(n, i):
    """On the trace domain H: L_first = [i == 0], L_last = [i == n-1], z_last = w^i - w^(n-1)."""
    lg = n.bit_length() - 1
    w = pow(1753635133440165772, 1 << (32 - lg), P)
    x = pow(w, i, P)
    return (x - pow(w, n - 1, P)) % P, int(i == 0), int(i == n - 1)


def test_g1op_constraints_vanish_on_trace(O, g1op_case):
    tr = g1op_case["trace"]
    n = tr.shape[1]
    alphas = [0x1234567890abcdef, 0x0fedcba987654321]
    for i in (0, 1, 17, 255, 256, n - 2, n - 1):
        zl, lf, ll = _trace_domain_consumer_args(n, i)
        acc = O.eval_constraints(O.AIR_G1_OP, 0, tr[:, i], tr[:, (i + 1) % n], np.zeros(0, dtype=np.uint64), alphas, zl, lf, ll)
        assert acc == [0, 0], f"row {i}"
    # a corrupted limb must violate something
    bad = tr[:, 5].copy()
    bad[70] = (int(bad[70]) + 1) % 65536
    zl, lf, ll = _trace_domain_consumer_args(n, 5)
    assert O.eval_constraints(O.AIR_G1_OP, 0, bad, tr[:, 6], np.zeros(0, dtype=np.uint64), alphas, zl, lf, ll) != [0, 0]


def test_g1exp_constraints_vanish_on_trace(O, g1exp_case):
    tr, pi = g1exp_case["trace"], g1exp_case["pi"]
    n = tr.shape[1]
    alphas = [0xdeadbeefcafef00d, 0x1111111122222222]
    rows = [0, 1, 2, 61, 62, 63, 64, 510, 511, 512, 513, 1023, 1024, 40000, n - 2, n - 1]
    for i in rows:
        zl, lf, ll = _trace_domain_consumer_args(n, i)
        acc = O.eval_constraints(O.AIR_G1_EXP, 128, tr[:, i], tr[:, (i + 1) % n], pi, alphas, zl, lf, ll)
        assert acc == [0, 0], f"row {i}"
    # wrong public input (instance 0 output.x limb) breaks the binding on that instance's last row
    pi2 = pi.copy()
    pi2[40] = (int(pi2[40]) + 1) & 0xFFFFFFFF
    zl, lf, ll = _trace_domain_consumer_args(n, 511)
    assert O.eval_constraints(O.AIR_G1_EXP, 128, tr[:, 511], tr[:, 512], pi2, alphas, zl, lf, ll) != [0, 0]


def test_fqexp_constraints_vanish_on_trace_and_power(O, fqexp_case):
    """FqExpStark (src/fields/fq/exp.rs): every constraint vanishes on the generated trace (all three row kinds, block
    boundaries, wrap-around), outputs = offset * x^e with Python integers (exp.rs:239-243), a wrong output is caught."""
    tr, pi = fqexp_case["trace"], fqexp_case["pi"]
    n = tr.shape[1]
    L = O.lib()
    assert (L.orc_air_num_columns(O.AIR_FQ_EXP, 128), L.orc_air_num_public_inputs(O.AIR_FQ_EXP, 128), L.orc_air_num_permutation_zs(O.AIR_FQ_EXP, 128)) == (960, 4096, 286)
    alphas = [0x0123456789abcdef, 0x2222222211111111]
    for i in [0, 1, 2, 61, 62, 63, 64, 510, 511, 512, 513, 30000, n - 2, n - 1]:
        zl, lf, ll = _trace_domain_consumer_args(n, i)
        assert O.eval_constraints(O.AIR_FQ_EXP, 128, tr[:, i], tr[:, (i + 1) % n], pi, alphas, zl, lf, ll) == [0, 0], f"row {i}"
    for k in (0, 5, 127):
        x, off, e = fqexp_case["native"][k]
        assert sum(int(pi[32 * k + 24 + i]) << (32 * i) for i in range(8)) == off * pow(x, e, O.BN_P) % O.BN_P
    pi2 = pi.copy()
    pi2[24] = (int(pi2[24]) + 1) & 0xFFFFFFFF                      # instance 0 output limb 0
    zl, lf, ll = _trace_domain_consumer_args(n, 511)
    assert O.eval_constraints(O.AIR_FQ_EXP, 128, tr[:, 511], tr[:, 512], pi2, alphas, zl, lf, ll) != [0, 0]
    bad = tr[:, 7].copy()
    bad[40] = (int(bad[40]) + 1) % 65536                           # a limb of the product on a squaring row
    zl, lf, ll = _trace_domain_consumer_args(n, 7)
    assert O.eval_constraints(O.AIR_FQ_EXP, 128, bad, tr[:, 8], pi, alphas, zl, lf, ll) != [0, 0]


def test_fq12expu64_constraints_vanish_on_trace(O, fq12expu64_case):
    """Fq12ExpU64Stark (src/fields/fq12_u64/exp_u64.rs, flags_u64.rs): every constraint vanishes on the generated trace
    (multiply / square / idle rows, instance boundaries at multiples of 128, wrap-around); a wrong exponent is caught."""
    tr, pi = fq12expu64_case["trace"], fq12expu64_case["pi"]
    n = tr.shape[1]
    L = O.lib()
    assert (L.orc_air_num_columns(O.AIR_FQ12_EXP_U64, 16), L.orc_air_num_public_inputs(O.AIR_FQ12_EXP_U64, 16), L.orc_air_num_permutation_zs(O.AIR_FQ12_EXP_U64, 16)) == (9792, 9232, 5328)
    alphas = [0x0f0e0d0c0b0a0908, 0x1020304050607080]
    for i in [0, 1, 2, 3, 126, 127, 128, 129, 1000, n - 2, n - 1]:
        zl, lf, ll = _trace_domain_consumer_args(n, i)
        assert O.eval_constraints(O.AIR_FQ12_EXP_U64, 16, tr[:, i], tr[:, (i + 1) % n], pi, alphas, zl, lf, ll) == [0, 0], f"row {i}"
    pi2 = pi.copy()
    pi2[384] = (int(pi2[384]) + 2) % O.GL_P                         # instance 0 exponent
    zl, lf, ll = _trace_domain_consumer_args(n, 0)
    assert O.eval_constraints(O.AIR_FQ12_EXP_U64, 16, tr[:, 0], tr[:, 1], pi2, alphas, zl, lf, ll) != [0, 0]


def test_g1exp_shape(O):
    L = O.lib()
    assert L.orc_air_num_columns(O.AIR_G1_EXP, 128) == 1676
    assert L.orc_air_num_public_inputs(O.AIR_G1_EXP, 128) == 7168
    assert L.orc_air_num_permutation_zs(O.AIR_G1_EXP, 128) == 762
    assert L.orc_air_num_columns(O.AIR_G1_OP, 0) == 2283
    assert L.orc_air_num_permutation_zs(O.AIR_G1_OP, 0) == 1264


def test_g1op_prove_verify_and_digest(O, g1op_case, golden):
    w = g1op_case["proof"]
    g = golden["proof_digests"]["g1op_rows512_seed0"]
    assert hashlib.sha256(g1op_case["trace"].tobytes()).hexdigest() == g["trace_sha256"]
    assert len(w) == g["proof_words"]
    assert hashlib.sha256(w.astype("<u8").tobytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(O.AIR_G1_OP, 0, w) == (0, "")
    # header: degree_bits 9, 2283 columns, 1264 Zs, 4 quotient polys, 0 PI, cap 4, rate 1, 1 FRI layer, arity 4, 32 final coeffs, 84 queries
    assert [int(x) for x in w[1:12]] == [9, 2283, 1264, 4, 0, 4, 1, 1, 4, 32, 84]


def test_fri_final_poly_times_x_switch(O, S, g1op_case):
    """plonky2 0.1.x multiplies the FRI final polynomial by X (prove_openings: insert(0, ZERO); fri_combine_initial:
    sum * subgroup_x); later upstream versions do not.  The dependency is un-vendored ([DEP-RECALL]), so both forms are
    kept behind a switch: each verifier (oracle and product host verifier) accepts its own form and rejects the other."""
    stark = S.G1Stark()
    cfg1, cfg0 = stark.config(), stark.config()
    assert cfg1.fri_variant == S.api.FRI_TIMES_X
    cfg0.fri_variant = S.api.FRI_PLAIN
    w1 = g1op_case["proof"]
    try:
        O.set_final_poly_times_x(False)
        w0, _ = O.prove(O.AIR_G1_OP, 0, g1op_case["trace"], g1op_case["pi"])
        assert O.verify(O.AIR_G1_OP, 0, w0) == (0, "")
        assert O.verify(O.AIR_G1_OP, 0, w1)[0] != 0
    finally:
        O.set_final_poly_times_x(True)
    assert not np.array_equal(w0, w1)
    assert O.verify(O.AIR_G1_OP, 0, w1) == (0, "")
    assert O.verify(O.AIR_G1_OP, 0, w0)[0] != 0
    S.verify_stark_proof(stark, S.Proof(w1, 9), cfg1)
    S.verify_stark_proof(stark, S.Proof(w0, 9), cfg0)
    for w, cfg in ((w1, cfg0), (w0, cfg1)):
        with pytest.raises(S.SbnError):
            S.verify_stark_proof(stark, S.Proof(w, 9), cfg)
    # a zero-initialised field selects the default protocol, not the other one (include/sbn.h sbn_fri_variant)
    cfgz = stark.config()
    cfgz.fri_variant = S.api.FRI_DEFAULT
    S.verify_stark_proof(stark, S.Proof(w1, 9), cfgz)
    assert S.lib().sbn_abi_version() == 4


@pytest.mark.parametrize("where", ["trace_cap", "opening", "fri_cap", "query_leaf", "final_poly", "pow", "truncate", "noncanonical"])
def test_oracle_rejects_tampering(O, g1op_case, where):
    w = g1op_case["proof"].copy()
    ncol, nz = 2283, 1264
    caps = 12 + 3 * 64
    openings = caps
    fri_caps = openings + 2 * (2 * ncol + 2 * nz + 4)
    queries = fri_caps + 64
    if where == "trace_cap":
        w[12] ^= 1
    elif where == "opening":
        w[openings + 10] = (int(w[openings + 10]) + 1) % P
    elif where == "fri_cap":
        w[fri_caps + 3] ^= 1
    elif where == "query_leaf":
        w[queries + 7] = (int(w[queries + 7]) + 1) % P
    elif where == "final_poly":
        w[-2 - 5] = (int(w[-2 - 5]) + 1) % P
    elif where == "pow":
        w[-1] = (int(w[-1]) + 1) % P
    elif where == "truncate":
        w = w[:-3]
    elif where == "noncanonical":
        w[openings] = P
    rc, why = O.verify(O.AIR_G1_OP, 0, w)
    assert rc != 0, why


def test_oracle_rejects_invalid_witness(O, g1op_case):
    """A trace that violates a constraint yields a proof the verifier rejects (quotient not a polynomial)."""
    tr = g1op_case["trace"].copy()
    tr[64 + 16, 9] = (int(tr[64 + 16, 9]) + 1) % 65536      # new_x limb 0 of row 9
    w, _ = O.prove(O.AIR_G1_OP, 0, tr, np.zeros(0, dtype=np.uint64))
    rc, why = O.verify(O.AIR_G1_OP, 0, w)
    assert rc != 0


@pytest.mark.parametrize("table", ["modular", "fq12mul"])
def test_single_operation_tables_oracle(O, golden, table):
    """ModularStark (src/modular/modular.rs:361-537) and Fq12Stark (src/fields/fq12/mul.rs:355-517) as the oracle restates
    them: constraints vanish on every row pair of a seeded trace and not on a corrupted one; prove -> verify; the proof is the
    committed one; a tampered proof is rejected."""
    kind, inputs, gen, seed = {"modular": (O.AIR_MODULAR, O.modular_inputs, O.modular_trace, 6), "fq12mul": (O.AIR_FQ12_MUL, O.fq12mul_inputs, O.fq12mul_trace, 7)}[table]
    ops, native = inputs(512, seed)
    tr = gen(ops)
    n = tr.shape[1]
    alphas = [0x1234567890ABCDEF, 0x0FEDCBA987654321]
    nopi = np.zeros(0, dtype=np.uint64)
    for i in (0, 1, 255, n - 2, n - 1):
        zl = 1 if i != n - 1 else 0
        lf, ll = (1 if i == 0 else 0), (1 if i == n - 1 else 0)
        assert O.eval_constraints(kind, 0, tr[:, i], tr[:, (i + 1) % n], nopi, alphas, zl, lf, ll) == [0, 0], i
    bad = tr[:, 7].copy(); bad[33] = (int(bad[33]) + 1) % 65536          # an output limb
    assert O.eval_constraints(kind, 0, bad, tr[:, 8], nopi, alphas, 1, 0, 0) != [0, 0]
    w, _ = O.prove(kind, 0, tr, nopi)
    g = golden["proof_digests"][f"{table}_rows512_seed{seed}"]
    assert hashlib.sha256(tr.tobytes()).hexdigest() == g["trace_sha256"]
    assert len(w) == g["proof_words"] and hashlib.sha256(w.astype("<u8").tobytes()).hexdigest() == g["proof_sha256"]
    assert O.verify(kind, 0, w) == (0, "")
    t = w.copy(); t[100] ^= 1
    assert O.verify(kind, 0, t)[0] != 0
