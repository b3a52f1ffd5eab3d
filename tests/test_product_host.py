"""CPU tests of the product's host side: the C-ABI library loads and exports every declared symbol,
table shapes, witness generation (host code) bit-equal to the oracle, the host verifier accepts the
oracle's proofs and rejects tampered ones, and the prover refuses to run without a GPU (no fallback)."""
import ctypes as C
import hashlib
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = 0xFFFFFFFF00000001


def test_library_exports_every_declared_symbol(S):
    hdr = open(os.path.join(ROOT, "include", "sbn.h")).read()
    declared = set(re.findall(r"\b(sbn_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"sbn_status", "sbn_air_kind"}
    L = S.lib()
    missing = [d for d in sorted(declared) if not hasattr(L, d)]
    assert not missing, missing
    assert set(S.EXPORTS) == declared


def test_shapes_and_config(S):
    st = S.G1ExpStark(128)
    assert (st.num_columns, st.num_public_inputs, st.num_permutation_zs(), st.num_constraints) == (1676, 7168, 762, 9027)
    g = S.G1Stark()
    assert (g.num_columns, g.num_public_inputs, g.num_permutation_zs()) == (2283, 0, 1264)
    c = st.config()
    assert (c.security_bits, c.num_challenges, c.rate_bits, c.cap_height, c.proof_of_work_bits, c.fri_arity_bits,
            c.fri_final_poly_bits, c.num_query_rounds) == (100, 2, 1, 4, 16, 4, 5, 84)
    assert S.G1ExpStark(0).num_columns == 0 and S.G1ExpStark(129).num_columns == 0     # unknown shapes report 0


def test_host_transcript_permutation(S, O, golden):
    """The Fiat-Shamir hasher of prove()/verify(): sparse partial rounds == the plain definition == the oracle == KATs."""
    kat = golden["poseidon_kat"]["vectors"]
    states = np.array([[int(x, 16) for x in v["input"]] for v in kat], dtype=np.uint64)
    want = [[int(x, 16) for x in v["output"]] for v in kat]
    assert [[int(x) for x in r] for r in S.poseidon_permute_host(states)] == want
    assert [[int(x) for x in r] for r in S.poseidon_permute_host(states, use_definition=True)] == want
    rng = np.random.default_rng(11)
    st = rng.integers(0, P, size=(512, 12), dtype=np.uint64)
    st[0] = P - 1; st[1] = 0; st[2, :] = [P - 1, 0, 1, 0xFFFFFFFF, 0xFFFFFFFF00000000, 1 << 32, 1 << 63, 2, 3, 4, 5, 6]
    fast, plain = S.poseidon_permute_host(st), S.poseidon_permute_host(st, use_definition=True)
    assert np.array_equal(fast, plain)
    for r in (0, 1, 2, 3, 100, 511):
        assert [int(x) for x in fast[r]] == O.poseidon_permute([int(x) for x in st[r]])
    with pytest.raises(S.SbnError) as e:
        S.poseidon_permute_host(np.full((1, 12), P, dtype=np.uint64))
    assert e.value.code == -2
    # the scalar full rounds (CPUs without AVX-512) in a fresh process, where SBN_NO_AVX512 is read
    code = ("import numpy as np, starky_bn254_amd as S; st = np.random.default_rng(11).integers(0, 0xFFFFFFFF00000001, size=(64, 12), dtype=np.uint64); "
            "a = S.poseidon_permute_host(st); b = S.poseidon_permute_host(st, use_definition=True); assert np.array_equal(a, b); print(int(a[5, 3]))")
    env = dict(os.environ, SBN_NO_AVX512="1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert int(out.stdout.strip()) == int(S.poseidon_permute_host(st[:64])[5, 3])


def test_tracegen_g1op_matches_oracle(S, O):
    for rows, seed in ((512, 0), (1024, 3)):
        pts, _ = O.g1op_inputs(rows, seed)
        assert np.array_equal(S.G1Stark().generate_trace(pts), O.g1op_trace(pts))


@pytest.mark.parametrize("table,rows,seed", [("ModularStark", 512, 6), ("ModularStark", 1024, 16), ("Fq12Stark", 512, 7)])
def test_single_operation_tables_shape_tracegen_and_products(S, O, golden, table, rows, seed):
    """The reference's own test tables for the modular gadget and the Fq12 product (test_modular_stark modular.rs:540-560,
    test_fq12_mul fq12/mul.rs): column counts, host generator == the oracle's independent generator word for word, outputs
    == the products computed with Python integers (the reference asserts the same against arkworks, modular.rs:405-406,
    mul.rs:388-389), the product verifier accepts the oracle's proof."""
    stark = getattr(S, table)()
    kind, inputs, gen = {"ModularStark": (O.AIR_MODULAR, O.modular_inputs, O.modular_trace), "Fq12Stark": (O.AIR_FQ12_MUL, O.fq12mul_inputs, O.fq12mul_trace)}[table]
    L = O.lib()
    assert stark.num_columns == L.orc_air_num_columns(kind, 0) == {"ModularStark": 812, "Fq12Stark": 9722}[table]   # modular.rs:368, mul.rs:360
    assert stark.num_permutation_zs() == L.orc_air_num_permutation_zs(kind, 0) == {"ModularStark": 444, "Fq12Stark": 5328}[table]
    assert stark.num_public_inputs == 0 and stark.constraint_degree() == 3
    ops, native = inputs(rows, seed)
    trace = stark.generate_trace(ops)
    ref = gen(ops)
    assert np.array_equal(trace, ref)
    if rows == 512:
        g = golden["proof_digests"][f"{'modular' if table == 'ModularStark' else 'fq12mul'}_rows512_seed{seed}"]
        assert hashlib.sha256(trace.tobytes()).hexdigest() == g["trace_sha256"]
    limbs = lambda col0, r: sum(int(trace[col0 + i, r]) << (16 * i) for i in range(16))
    for r in (0, 1, rows // 2, rows - 1):
        if table == "ModularStark":
            a, b = native[r]
            assert limbs(0, r) == a and limbs(16, r) == b and limbs(32, r) == a * b % O.BN_P
        else:
            x, y = native[r]
            assert [limbs(384 + 16 * c, r) for c in range(12)] == O.fq12_mul(x, y)
    # every range-checked column holds 16-bit limbs; signs are +-1; the filter is 1
    main = {"ModularStark": 145, "Fq12Stark": 1729}[table]
    assert trace[main - 1].min() == 1 and trace[main - 1].max() == 1
    if rows == 512:
        w, _ = O.prove(kind, 0, ref, np.zeros(0, dtype=np.uint64))
        assert hashlib.sha256(w.astype("<u8").tobytes()).hexdigest() == g["proof_sha256"]
        S.verify_stark_proof(stark, S.Proof(w, 9), stark.config())
        t = w.copy(); t[len(t) // 2] ^= 1
        with pytest.raises(S.SbnError):
            S.verify_stark_proof(stark, S.Proof(t, 9), stark.config())
    # argument errors of the generators: a non-power-of-two height, an operand >= p
    with pytest.raises(S.SbnError):
        stark.generate_trace(ops[:300])
    bad = ops.copy(); bad[5, 0:8] = 0xFFFFFFFF
    with pytest.raises(S.SbnError):
        stark.generate_trace(bad)


def test_tracegen_g1exp_matches_oracle_and_golden(S, g1exp_case, golden):
    tr, pi = S.G1ExpStark(128).generate_trace_and_public_inputs(g1exp_case["ios"])
    assert np.array_equal(pi, g1exp_case["pi"])
    assert np.array_equal(tr, g1exp_case["trace"])
    g = golden["proof_digests"]["g1exp_io128_seed1"]
    assert hashlib.sha256(tr.tobytes()).hexdigest() == g["trace_sha256"]
    assert hashlib.sha256(pi.tobytes()).hexdigest() == g["pi_sha256"]


def test_tracegen_error_behaviour(S, O):
    pts, _ = O.g1op_inputs(512, 0)
    bad = pts.copy()
    bad[3, 16:32] = bad[3, 0:16]                       # b == a: affine add degenerates (x2 - x1 = 0)
    with pytest.raises(S.SbnError) as e:
        S.G1Stark().generate_trace(bad)
    assert e.value.code == -8
    bad = pts.copy()
    bad[0, 0:8] = 0xFFFFFFFF                           # coordinate >= p
    with pytest.raises(S.SbnError) as e:
        S.G1Stark().generate_trace(bad)
    assert e.value.code == -1
    with pytest.raises(S.SbnError):                    # rows not a power of two
        S.G1Stark().generate_trace(pts[:300])
    ios, _ = O.g1exp_inputs(4, 1)
    with pytest.raises(S.SbnError) as e:               # < 2^16 rows: range_check.rs:26
        S.G1ExpStark(4).generate_trace_and_public_inputs(ios)
    assert e.value.code == -7


def test_host_verifier_accepts_oracle_proof_and_rejects_tampering(S, g1op_case):
    stark = S.G1Stark()
    cfg = stark.config()
    w = g1op_case["proof"]
    S.verify_stark_proof(stark, S.Proof(w, 9), cfg)
    for idx, delta in ((12, 1), (12 + 3 * 64 + 5, 1), (len(w) - 1, 1), (12 + 3 * 64 + 2 * (2 * 2283 + 2 * 1264 + 4) + 64 + 11, 1)):
        t = w.copy()
        t[idx] = (int(t[idx]) + delta) % P
        with pytest.raises(S.SbnError) as e:
            S.verify_stark_proof(stark, S.Proof(t, 9), cfg)
        assert e.value.code == -6
    with pytest.raises(S.SbnError) as e:               # malformed: truncated
        S.verify_stark_proof(stark, S.Proof(w[:-1], 9), cfg)
    assert e.value.code == -5
    t = w.copy(); t[12 + 3 * 64] = P                   # non-canonical element
    with pytest.raises(S.SbnError) as e:
        S.verify_stark_proof(stark, S.Proof(t, 9), cfg)
    assert e.value.code == -5
    with pytest.raises(S.SbnError):                    # proof of another table
        S.verify_stark_proof(S.G1ExpStark(128), S.Proof(w, 9), cfg)


def test_no_cpu_fallback(S, g1op_case):
    """Without a HIP device the prover path must fail loudly (SBN_ERR_NO_DEVICE), never compute on the CPU."""
    if S.lib().sbn_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(S.SbnError) as e:
        S.prove(S.G1Stark(), S.G1Stark().config(), g1op_case["trace"], g1op_case["pi"])
    assert e.value.code == -3
    with pytest.raises(S.SbnError) as e:
        S.commit_values(np.zeros((2, 512), dtype=np.uint64))
    assert e.value.code == -3


def test_product_never_loads_the_oracle():
    """The shipped package must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "starky_bn254_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".hip", ".cuh", ".hpp", ".cpp", "Makefile")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in txt and "oracle/" not in txt and "oracle_lib" not in txt, os.path.join(dp, f)


def test_g2_shape_tracegen_and_scalar_mult(S, O, g2exp_case, golden):
    """G2ExpStark (BASELINE config[3]): shape = SURVEY section 8, host witness == oracle witness == golden digest,
    outputs == python-int x*s+offset on the twist, constraints vanish on sampled rows."""
    st = S.G2ExpStark(128)
    assert (st.num_columns, st.num_public_inputs, st.num_permutation_zs(), st.num_constraints) == (2822, 13312, 1524, 16455)
    tr, pi = st.generate_trace_and_public_inputs(g2exp_case["ios"])
    assert np.array_equal(pi, g2exp_case["pi"]) and np.array_equal(tr, g2exp_case["trace"])
    g = golden["proof_digests"]["g2exp_io128_seed2"]
    assert hashlib.sha256(tr.tobytes()).hexdigest() == g["trace_sha256"]
    gs = golden["g2_scalar_mult"]
    assert np.array_equal(np.array(gs["ios"], dtype=np.uint32), g2exp_case["ios"][:3])
    for k, out in enumerate(gs["outputs"]):
        got = [sum(int(pi[104 * k + 72 + 8 * q + i]) << (32 * i) for i in range(8)) for q in range(4)]
        assert got == [int(v, 16) for v in out]
    n = tr.shape[1]
    w = pow(1753635133440165772, 1 << (32 - 16), P)
    for i in (0, 1, 62, 63, 511, 512, 33333, n - 1):
        x = pow(w, i, P)
        acc = O.eval_constraints(O.AIR_G2_EXP, 128, tr[:, i], tr[:, (i + 1) % n], pi, [0x1234567, 0x7654321],
                                 (x - pow(w, n - 1, P)) % P, int(i == 0), int(i == n - 1))
        assert acc == [0, 0], i


def test_fq12_shape_tracegen_and_power(S, O, fq12exp_case, golden):
    """Fq12ExpStark: shapes = SURVEY section 8 (9802 cols @16, 11786 cols / 299008 PI / 310515 constraints @512),
    host witness == oracle witness == golden digest, outputs == offset * x^e by independent tower arithmetic,
    constraints vanish on sampled rows."""
    st = S.Fq12ExpStark(16)
    assert (st.num_columns, st.num_public_inputs, st.num_permutation_zs()) == (9802, 9344, 5328)
    big = S.Fq12ExpStark(512)
    assert (big.num_columns, big.num_public_inputs, big.num_permutation_zs(), big.num_constraints) == (11786, 299008, 5328, 310515)
    tr, pi = st.generate_trace_and_public_inputs(fq12exp_case["ios"])
    assert np.array_equal(pi, fq12exp_case["pi"]) and np.array_equal(tr, fq12exp_case["trace"])
    g = golden["proof_digests"]["fq12exp_io16_seed3"]
    assert hashlib.sha256(tr.tobytes()).hexdigest() == g["trace_sha256"]
    for k in (0, 7, 15):
        x, off, e = fq12exp_case["native"][k]
        expect = O.fq12_mul(off, O.fq12_pow(x, e))
        got = [sum(int(pi[584 * k + 392 + 16 * c + i]) << (16 * i) for i in range(16)) for c in range(12)]
        assert got == expect
    n = tr.shape[1]
    w = pow(1753635133440165772, 1 << (32 - 13), P)
    for i in (0, 1, 62, 63, 511, 512, 4097, n - 1):
        x = pow(w, i, P)
        acc = O.eval_constraints(O.AIR_FQ12_EXP, 16, tr[:, i], tr[:, (i + 1) % n], pi, [0x1234567, 0x7654321],
                                 (x - pow(w, n - 1, P)) % P, int(i == 0), int(i == n - 1))
        assert acc == [0, 0], i


def test_fqexp_shape_tracegen_and_power(S, O, fqexp_case, golden):
    """FqExpStark(128) (src/fields/fq/exp.rs, the reference's test_fq_exp_raw size): shape constants, host witness
    generation bit-equal to the oracle's, outputs = offset * x^e computed with Python integers (exp.rs:239-243)."""
    stark = S.FqExpStark(128)
    cfg = stark.config()
    assert (stark.num_columns, stark.num_public_inputs, stark.num_permutation_zs(cfg)) == (960, 4096, 286)
    trace, pi = stark.generate_trace_and_public_inputs(fqexp_case["ios"])
    assert np.array_equal(trace, fqexp_case["trace"]) and np.array_equal(pi, fqexp_case["pi"])
    g = golden["proof_digests"]["fqexp_io128_seed4"]
    assert hashlib.sha256(trace.tobytes()).hexdigest() == g["trace_sha256"]
    assert hashlib.sha256(pi.tobytes()).hexdigest() == g["pi_sha256"]
    for k in (0, 1, 63, 127):
        x, off, e = fqexp_case["native"][k]
        out = sum(int(pi[32 * k + 24 + i]) << (32 * i) for i in range(8))
        assert out == off * pow(x, e, O.BN_P) % O.BN_P
    bad = fqexp_case["ios"].copy()
    bad[3, 0:8] = 0xFFFFFFFF                                       # x >= p
    with pytest.raises(S.SbnError) as e:
        stark.generate_trace_and_public_inputs(bad)
    assert e.value.code == -1
    with pytest.raises(S.SbnError) as e:                           # fewer than 2^16 rows: u16 range check impossible
        S.FqExpStark(64).generate_trace_and_public_inputs(fqexp_case["ios"][:64])
    assert e.value.code == -7


def test_fq12expu64_shape_tracegen_and_power(S, O, fq12expu64_case, golden):
    """Fq12ExpU64Stark(16) (src/fields/fq12_u64/exp_u64.rs, the reference's test_fq12_exp_u64_raw size): shape constants,
    host witness generation bit-equal to the oracle's, outputs = offset * x^e in Fq12 with Python integers."""
    stark = S.Fq12ExpU64Stark(16)
    cfg = stark.config()
    assert (stark.num_columns, stark.num_public_inputs, stark.num_permutation_zs(cfg)) == (9792, 9232, 5328)
    trace, pi = stark.generate_trace_and_public_inputs(fq12expu64_case["ios"])
    assert np.array_equal(trace, fq12expu64_case["trace"]) and np.array_equal(pi, fq12expu64_case["pi"])
    g = golden["proof_digests"]["fq12expu64_io16_seed5"]
    assert hashlib.sha256(trace.tobytes()).hexdigest() == g["trace_sha256"]
    assert hashlib.sha256(pi.tobytes()).hexdigest() == g["pi_sha256"]
    for k in (0, 9, 15):
        x, off, e = fq12expu64_case["native"][k]
        got = [sum(int(pi[577 * k + 385 + 16 * c + i]) << (16 * i) for i in range(16)) for c in range(12)]
        assert got == O.fq12_mul(off, O.fq12_pow(x, e)) and int(pi[577 * k + 384]) == e
    bad = fq12expu64_case["ios"].copy()
    bad[2, 192:194] = 0xFFFFFFFF                                   # exponent >= the Goldilocks modulus
    with pytest.raises(S.SbnError) as e:
        stark.generate_trace_and_public_inputs(bad)
    assert e.value.code == -2


def test_generated_tables_and_instruction_streams_are_current(tmp_path):
    """Every committed *.inc under csrc/ (Poseidon constants, sparse partial-round tables, MDS / S-box / fold / multiply
    instruction streams) is what its generator in tools/ produces; the generators check their own algebra
    (rewritten partial rounds == plain permutation, plonky2 test vector) before writing."""
    import subprocess
    import sys
    env = dict(os.environ, SBN_GEN_OUT=str(tmp_path))
    for script in ("gen_poseidon_constants.py", "gen_poseidon_fast_partial.py", "gen_poseidon_mds_asm.py", "gen_poseidon_sbox_asm.py"):
        subprocess.run([sys.executable, script], cwd=os.path.join(ROOT, "tools"), env=env, check=True, capture_output=True)
    produced = sorted(os.listdir(tmp_path))
    assert len(produced) >= 9
    csrc = os.path.join(ROOT, "starky_bn254_amd", "csrc")
    assert sorted(f for f in os.listdir(csrc) if f.endswith(".inc")) == produced
    for f in produced:
        assert open(os.path.join(tmp_path, f)).read() == open(os.path.join(csrc, f)).read(), f
    for f in ("poseidon_eff_consts.inc", "poseidon_fast_consts.inc"):       # the oracle's own copies of the sparse-round tables
        assert open(os.path.join(tmp_path, f)).read() == open(os.path.join(ROOT, "oracle", f)).read(), f



def test_proof_fields_follow_the_documented_layout(S, g1op_case):
    """Proof.fields() (the struct view of include/sbn.h's word order, which the Rust shim's proof_from_words mirrors)
    consumes an oracle proof exactly and lands every field where the verifier reads it."""
    w = g1op_case["proof"]
    f = S.Proof(w, 9).fields()
    assert f["degree_bits"] == 9 and f["trace_cap"].shape == (16, 4) and f["permutation_zs_cap"].shape == (16, 4)
    op = f["openings"]
    assert op["local_values"].shape == (2283, 2) and op["permutation_zs_next"].shape == (1264, 2) and op["quotient_polys"].shape == (4, 2)
    fri = f["opening_proof"]
    assert len(fri["commit_phase_merkle_caps"]) == 1 and len(fri["query_round_proofs"]) == 84
    q = fri["query_round_proofs"][0]
    assert [len(v) for v, _ in q["initial_trees_proof"]] == [2283, 1264, 4]
    assert all(sib.shape == (10 - 4, 4) for _, sib in q["initial_trees_proof"])
    assert q["steps"][0]["evals"].shape == (16, 2) and q["steps"][0]["merkle_proof"].shape == (10 - 4 - 4, 4)
    assert fri["final_poly"].shape == (32, 2) and fri["pow_witness"] == int(w[-1]) and len(f["public_inputs"]) == 0
    assert np.array_equal(f["trace_cap"].ravel(), w[12:12 + 64])
    # the views alias the words: flipping a field through the view is what the verifier then rejects
    t = S.Proof(w.copy(), 9)
    t.fields()["opening_proof"]["final_poly"][3, 1] ^= 1
    with pytest.raises(S.SbnError):
        S.verify_stark_proof(S.G1Stark(), t, S.G1Stark().config())
    S.verify_stark_proof(S.G1Stark(), S.Proof(w, 9), S.G1Stark().config())
    with pytest.raises(ValueError):
        S.Proof(w[:-1], 9).fields()


def test_rust_shim_declarations_match_the_header():
    """integration/rust (source only, no Rust toolchain here): every `extern "C"` declaration names an entry point of
    include/sbn.h with the same number of arguments, the struct fields and table kinds agree, and the proof reader
    takes the fields in the header's documented order."""
    hdr = open(os.path.join(ROOT, "include", "sbn.h")).read()
    hdr_nc = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    shim = os.path.join(ROOT, "integration", "rust", "starky-bn254-amd", "src")
    ffi = open(os.path.join(shim, "ffi.rs")).read()
    decls = re.findall(r"pub fn (sbn_[a-z0-9_]+)\(([^)]*)\)", ffi)
    assert len(decls) >= 15
    for name, args in decls:
        m = re.search(r"\b%s\s*\(([^)]*)\)" % name, hdr_nc)
        assert m, name
        c_args = [a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]
        r_args = [a for a in args.split(",") if a.strip()]
        assert len(c_args) == len(r_args), (name, c_args, r_args)
    cfg_c = re.findall(r"uint32_t (\w+);", re.search(r"typedef struct sbn_config \{(.*?)\}", hdr_nc, re.S).group(1))
    cfg_r = re.findall(r"pub (\w+): u32", re.search(r"pub struct sbn_config \{(.*?)\}", ffi, re.S).group(1))
    assert cfg_c == cfg_r
    kinds_c = dict((k, int(v)) for k, v in re.findall(r"(SBN_AIR_\w+) = (\d+)", hdr_nc))
    kinds_r = dict((k, int(v)) for k, v in re.findall(r"pub const (SBN_AIR_\w+): i32 = (\d+);", ffi))
    assert kinds_c == kinds_r
    conv = open(os.path.join(shim, "convert.rs")).read()
    order = ["trace_cap = ", "permutation_zs_cap = ", "quotient_polys_cap = ", "local_values = ", "next_values = ",
             "(permutation_zs, permutation_zs_next) = ", "quotient_polys = ", "commit_phase_merkle_caps = ", "evals_proofs.push",
             "steps.push", "final_poly = ", "pow_witness = ", "public_inputs = "]
    at = [conv.index(s) for s in order]
    assert at == sorted(at)
    assert 'from_le_bytes(*b"SNBPROV1")' in conv and int.from_bytes(b"SNBPROV1", "little") == 0x31564F5250424E53


def _split_top_level(args):
    out, depth, cur = [], 0, ""
    for ch in args:
        if ch in "(<[{":
            depth += 1
        elif ch in ")>]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _call_args(src, opener):
    """argument text of every call `opener ... )` in src (balanced parentheses)"""
    out, at = [], 0
    while True:
        i = src.find(opener, at)
        if i < 0:
            return out
        j, depth = i + len(opener), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[j], 0)
            j += 1
        out.append(src[i + len(opener):j - 1])
        at = j


def test_rust_shim_prove_and_verify_have_the_reference_call_shape():
    """The shim's `prove` / `verify_stark_proof` take exactly the arguments the reference's call sites pass
    (prove::<F, C, _, D>(stark, &inner_config, trace, pi.try_into().unwrap(), &mut TimingTree::default()) and
    verify_stark_proof(stark, inner_proof.clone(), &inner_config): src/curves/g1/exp.rs:818-826,
    src/curves/g1/circuit.rs:192-200), with starky's generic parameter list <F, C, S, const D: usize>, so that those call
    expressions compile against the shim unchanged.  The reference files are read when present (this container)."""
    lib_rs = open(os.path.join(ROOT, "integration", "rust", "starky-bn254-amd", "src", "lib.rs")).read()
    m = re.search(r"^pub fn prove<([^>]*)>\((.*?)\)\s*->\s*Result<StarkProofWithPublicInputs<F, C, D>>\s*where(.*?)\{", lib_rs, re.S | re.M)
    assert m, "prove"
    assert [g.strip() for g in m.group(1).split(",")] == ["F", "C", "S", "const D: usize"]
    params = _split_top_level(m.group(2))
    assert [p.split(":")[0].strip() for p in params] == ["stark", "config", "trace_poly_values", "public_inputs", "timing"]
    assert [p.split(":", 1)[1].strip() for p in params] == ["S", "&StarkConfig", "Vec<PolynomialValues<F>>", "Vec<F>", "&mut TimingTree"]
    assert "S: Stark<F, D> + SbnTable" in m.group(3)
    v = re.search(r"^pub fn verify_stark_proof<([^>]*)>\((.*?)\)\s*->\s*Result<\(\)>", lib_rs, re.S | re.M)
    assert v, "verify_stark_proof"
    assert [g.strip() for g in v.group(1).split(",")] == ["F", "C", "S", "const D: usize"]
    vparams = _split_top_level(v.group(2))
    assert [p.split(":", 1)[1].strip() for p in vparams] == ["S", "StarkProofWithPublicInputs<F, C, D>", "&StarkConfig"]
    assert "words_from_proof" in open(os.path.join(ROOT, "integration", "rust", "starky-bn254-amd", "src", "convert.rs")).read()
    for rel in ("src/curves/g1/exp.rs", "src/curves/g1/circuit.rs", "src/curves/g2/circuit.rs", "src/fields/fq12/circuit.rs"):
        path = os.path.join("/root/reference", rel)
        if not os.path.exists(path):
            continue
        src = open(path).read()
        calls = _call_args(src, "prove::<F, C, _, D>(")
        assert calls, rel
        for c in calls:
            assert len(_split_top_level(c)) == len(params), (rel, c)
        vcalls = _call_args(src, " verify_stark_proof(")
        assert vcalls, rel
        for c in vcalls:
            assert len(_split_top_level(c)) == len(vparams), (rel, c)


@pytest.mark.parametrize("table,num_io", [("G1Stark", 0), ("G1ExpStark", 128), ("G2ExpStark", 128), ("Fq12ExpStark", 16), ("FqExpStark", 128),
                                           ("Fq12ExpU64Stark", 16), ("ModularStark", 0), ("Fq12Stark", 0), ("LookupStark", 0), ("FlagStark", 16), ("FlagU64Stark", 16)])
def test_regrouped_constraints_equal_the_oracle_on_random_rows(S, O, table, num_io):
    """air.cuh folds the constraints in regrouped form (local Horner sums, shared and factored limb convolutions, collapsed
    public-input block); the result must be the SAME field element as folding them one by one in the reference's order
    (the oracle's restatement of eval_packed_generic) -- on arbitrary rows, not only on valid traces: random field
    elements in every column, random public inputs, random challenges and selector values, several row pairs."""
    stark = getattr(S, table)(num_io) if num_io else getattr(S, table)()
    kind = {"G1Stark": O.AIR_G1_OP, "G1ExpStark": O.AIR_G1_EXP, "G2ExpStark": O.AIR_G2_EXP, "Fq12ExpStark": O.AIR_FQ12_EXP, "FqExpStark": O.AIR_FQ_EXP,
            "Fq12ExpU64Stark": O.AIR_FQ12_EXP_U64, "ModularStark": O.AIR_MODULAR, "Fq12Stark": O.AIR_FQ12_MUL, "LookupStark": O.AIR_LOOKUP,
            "FlagStark": O.AIR_FLAGS, "FlagU64Stark": O.AIR_FLAGS_U64}[table]
    rng = np.random.default_rng(hash(table) % 1000 + 17)
    ncol, npi = stark.num_columns, stark.num_public_inputs
    for trial in range(3):
        lv = rng.integers(0, P, size=ncol, dtype=np.uint64)
        nv = rng.integers(0, P, size=ncol, dtype=np.uint64)
        pi = rng.integers(0, P, size=npi, dtype=np.uint64)
        if trial == 1:                                   # small values as on a real trace: limbs, flags
            lv = rng.integers(0, 65536, size=ncol, dtype=np.uint64)
            nv = rng.integers(0, 2, size=ncol, dtype=np.uint64)
        alphas = [int(x) for x in rng.integers(1, P, size=2, dtype=np.uint64)]
        z_last, l_first, l_last = (int(x) for x in rng.integers(0, P, size=3, dtype=np.uint64))
        got = S.eval_constraints_host(stark, lv, nv, pi, alphas, z_last, l_first, l_last)
        want = O.eval_constraints(kind, num_io, lv, nv, pi, alphas, z_last, l_first, l_last)
        assert got == want, (table, trial)
    # a zero challenge (probability 2^-64 in a proof): the fold keeps only the last constraint
    got = S.eval_constraints_host(stark, lv, nv, pi, [0, alphas[1]], z_last, l_first, l_last)
    assert got == O.eval_constraints(kind, num_io, lv, nv, pi, [0, alphas[1]], z_last, l_first, l_last)


def test_split_exchange_sizes_and_argument_errors(S):
    """sbn_split_exchange_bytes (the staging memory a rank of the oversized-trace split needs): the send side holds two slots
    of one 64-column block per plane (two planes from four ranks up: the rows i + 2 travel as a second plane; one rank: no
    slot), the receive side the row-sharded trace and Z matrices plus the gather scratch; a world that is not a power of two
    <= 16 is refused when the prover is created (here: no device, so creation fails earlier with NO_DEVICE)."""
    from starky_bn254_amd import split
    stark = S.G1ExpStark(128)
    cfg = stark.config()
    C, Z, n, m = 1676, 762, 1 << 16, 1 << 17
    for world, planes in ((1, 1), (2, 1), (4, 2), (8, 2)):
        sb, rb = split.exchange_bytes(stark, cfg, 16, world)
        slot = planes * 64 * m if world > 1 else 0
        assert sb == max(2 * slot, 2 * (m // world), 2 * n) * 8
        assert rb == ((C + Z) * (m // world) * planes + max(2 * m, 2 * n * world)) * 8
    import ctypes as C_
    L = S.lib()
    comm = split._Comm(C_.sizeof(split._Comm), 0, None, 0, 3, None, None, 0, 0, split._A2A(lambda *a: 0), split._AGH(lambda *a: 0))
    h = C_.c_void_p()
    L.sbn_split_prover_create.argtypes = [C_.POINTER(S.api._AirDesc), C_.POINTER(S.api._Config), C_.c_uint32, C_.POINTER(split._Comm), C_.POINTER(C_.c_void_p)]
    rc = L.sbn_split_prover_create(C_.byref(stark._d), C_.byref(cfg._c), 16, C_.byref(comm), C_.byref(h))
    assert rc in (-1, -3) and not h.value          # BAD_ARG on a GPU box, NO_DEVICE here


def test_committed_profiles_json_parse():
    """Every profiles/*.json is one JSON document and every *.jsonl one document per line (collection scripts keep stderr --
    RCCL / gloo banners -- in a sibling .err file)."""
    import glob
    import json
    for f in glob.glob(os.path.join(ROOT, "profiles", "*.json")):
        json.load(open(f))
    for f in glob.glob(os.path.join(ROOT, "profiles", "*.jsonl")):
        for line in open(f):
            json.loads(line)


def test_design_numbers_point_at_committed_files():
    """DESIGN.md holds the current state with the profiles/ file behind every number -- every file it, README.md or
    INTEGRATION.md names must exist (a `*` stands for a family of files), and DESIGN.md stays wrapped outside its tables."""
    import glob
    import re
    for doc in ("DESIGN.md", "README.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, doc)).read()
        names = set(re.findall(r"`(?:profiles/)?(r[1-9]_[A-Za-z0-9_*{},.]+\.(?:json|jsonl|txt|csv|log))`", text))
        assert doc != "DESIGN.md" or len(names) >= 20
        for n in names:
            pats = [n]
            m = re.search(r"\{([^}]*)\}", n)
            if m:
                pats = [n[:m.start()] + alt + n[m.end():] for alt in m.group(1).split(",")]
            for pat in pats:
                assert glob.glob(os.path.join(ROOT, "profiles", pat)), (doc, n, pat)
    long_lines = [ln for ln in open(os.path.join(ROOT, "DESIGN.md")).read().splitlines() if len(ln) > 130 and not ln.startswith("|")]
    assert long_lines == [], long_lines[:3]
    assert os.path.exists(os.path.join(ROOT, "profiles", "HISTORY.md"))


def test_production_library_ignores_the_quotient_diagnostic_switch():
    """SBN_DIAG_QUOTIENT_SEGMASK (skip constraint segments, invalid proof) is compiled in only with -DSBN_DIAG."""
    src = open(os.path.join(ROOT, "starky_bn254_amd", "csrc", "prover.hip")).read()
    at = src.index('getenv("SBN_DIAG_QUOTIENT_SEGMASK")')
    assert src.rfind("#ifdef SBN_DIAG", 0, at) > src.rfind("#endif", 0, at)
    assert "SBN_DIAG" not in open(os.path.join(ROOT, "starky_bn254_amd", "csrc", "Makefile")).read().replace("SBN_DIAG_", "")
    lib = open(S_lib_path(), "rb").read()
    assert b"SBN_DIAG_QUOTIENT_SEGMASK" not in lib


def S_lib_path():
    import starky_bn254_amd as pkg
    return pkg.lib_path()


def test_unit_test_tables_shape_and_host_witness(S, O, golden):
    """The reference's two unit-test tables as product kinds: MyStark (lookup.rs:136-213; 4 columns, 2 pairs, 2 constraints)
    and FlagStark (flags.rs:379-547; 17 + 4 num_io columns, NO pairs): shapes, host witness == the oracle's, every constraint
    vanishes on the generated trace (both evaluators), a wrong table value does not; MyStark on the reference's fixed input
    reproduces the committed lookup fixture."""
    lk, fl, fu = S.LookupStark(), S.FlagStark(16), S.FlagU64Stark(16)
    assert (fu.num_columns, fu.num_public_inputs, fu.num_permutation_zs(), fu.num_constraints) == (71, 0, 0, 76)   # flags_u64.rs:285: 6 + 1 + 4 * 16
    exps, _ = O.flags_u64_inputs(16, 10)
    tu = fu.generate_trace(exps)
    assert np.array_equal(tu, O.flags_u64_trace(exps))
    assert [int(tu[3, 128 * 5 + 2 * k]) for k in range(64)] == [(int(exps[5]) >> k) & 1 for k in range(64)]        # filtered bits = the exponent's bits, LSB first
    assert (lk.num_columns, lk.num_public_inputs, lk.num_permutation_zs(), lk.num_constraints) == (4, 0, 2, 2)
    assert (fl.num_columns, fl.num_public_inputs, fl.num_permutation_zs(), fl.num_constraints) == (81, 0, 0, 98)   # flags.rs:376: MAIN_COLS + 2 + 1 + 4 * 16
    assert S.MyStark is S.LookupStark
    fx = golden["lookup_fixed"]
    t8 = lk.generate_trace(fx["inputs"], fx["table"])
    assert [int(x) for x in t8[2]] == fx["sorted_inputs"] and [int(x) for x in t8[3]] == fx["permuted_table"]
    ins, tab = O.lookup_inputs(512, 9)
    tr = lk.generate_trace(ins, tab)
    assert np.array_equal(tr, O.lookup_trace(ins, tab))
    limbs, _ = O.flags_inputs(16, 8)
    tf = fl.generate_trace(limbs)
    assert np.array_equal(tf, O.flags_trace(limbs))
    assert [int(tf[4, 1024 + 2 * k]) for k in range(256)] == [(int(limbs[2][k // 32]) >> (k % 32)) & 1 for k in range(256)]   # test_flag_native, flags.rs:354-368
    g = 0x185629DCDA58878C                                       # primitive_root_of_unity(13)^-1 is not needed: z_last = x - g^-1 is any nonzero value off the last row
    for stark, kind, num_io, t in ((lk, O.AIR_LOOKUP, 0, tr), (fl, O.AIR_FLAGS, 16, tf), (fu, O.AIR_FLAGS_U64, 16, tu)):
        n = t.shape[1]
        for r in (0, 1, 63, 127, 511, n - 2, n - 1):
            lv, nv = t[:, r], t[:, (r + 1) % n]
            sel = (1 if r != n - 1 else 0, 1 if r == 0 else 0, 1 if r == n - 1 else 0)   # transition, first row, last row selectors
            assert S.eval_constraints_host(stark, lv, nv, [], [3, 5], *sel) == [0, 0], (kind, r)
            assert O.eval_constraints(kind, num_io, lv, nv, [], [3, 5], *sel) == [0, 0]
    r = next(i for i in range(1, 512) if tr[2, i] != tr[2, i - 1])       # a row where a NEW value is looked up: it must sit beside itself
    bad = tr.copy()
    bad[3, r] = (int(bad[3, r]) + 1) % 512
    assert S.eval_constraints_host(lk, bad[:, r - 1], bad[:, r], [], [3, 5], 1, 0, 0) != [0, 0]
    with pytest.raises(S.SbnError):
        lk.generate_trace([1, 2, 3], [1, 2, 3])                  # not a power of two


@pytest.mark.parametrize("E,count", [(1, 128), (1, 13), (2, 16), (2, 5)])
def test_ifma_curve_chains_equal_the_scalar_chains(S, O, E, count):
    """The curve chains of the device witness (bn254w.cuh exp_chains; src/curves/g1/exp.rs:165-230, src/curves/g2/exp.rs:180-246) eight
    instances per AVX-512 IFMA register (csrc/chains_ifma.hpp: radix 2^52, R' = 2^260, every stored word converted back to the
    2^256 Montgomery form) against one instance at a time: the same [count][257][3][E][4] words, whole and ragged groups, all-zero and
    all-one scalars, and the degenerate instance (offset == x with bit 0 set) refused by both."""
    import ctypes as C
    L = S.lib()
    ios = (O.g1exp_inputs(count, 21)[0] if E == 1 else O.g2exp_inputs(count, 22)[0]).copy()
    ios[1, 32 * E:] = 0
    ios[2, 32 * E:] = 0xFFFFFFFF
    words = count * 257 * 3 * E * 4
    out = {}
    for form in (1, 2):
        ja, jb = np.zeros(words, dtype=np.uint64), np.full(words, 7, dtype=np.uint64)
        rc = L.sbn_host_curve_chains(E, ios.ctypes.data_as(C.c_void_p), count, ja.ctypes.data_as(C.c_void_p), jb.ctypes.data_as(C.c_void_p), form)
        if form == 2 and rc == -4:
            pytest.skip("this CPU has no AVX-512 IFMA")
        assert rc == 0, L.sbn_last_error()
        out[form] = (ja, jb)
    assert np.array_equal(out[1][0], out[2][0]) and np.array_equal(out[1][1], out[2][1])
    # the last stored point of chain b is x * s + offset (Jacobian, Montgomery form): check one instance with python integers
    P = O.BN_P
    if E == 1:
        k = min(3, count - 1)
        _, native = O.g1exp_inputs(count, 21)
        x, off, s = native[k]
        want = O.g1_add(O.g1_mul(x, s), off)
        w = out[2][1][((k * 257 + 256) * 3) * 4:((k * 257 + 256) * 3 + 3) * 4]
        X, Y, Z = (sum(int(w[4 * c + i]) << (64 * i) for i in range(4)) * pow(2, -256, P) % P for c in range(3))
        zi = pow(Z, -1, P)
        assert (X * zi * zi % P, Y * zi * zi * zi % P) == tuple(want)
    bad = ios.copy()
    bad[0, 16 * E:32 * E] = bad[0, 0:16 * E]
    bad[0, 32 * E] = 1
    for form in (1, 2):
        ja, jb = np.zeros(words, dtype=np.uint64), np.zeros(words, dtype=np.uint64)
        assert L.sbn_host_curve_chains(E, bad.ctypes.data_as(C.c_void_p), count, ja.ctypes.data_as(C.c_void_p), jb.ctypes.data_as(C.c_void_p), form) == -8


def test_settings_are_read_once_checked_and_experiments_need_their_switch(S):
    """csrc/settings.hpp: every SBN_* switch is parsed in one place; a value that is not understood is an
    error; experiment switches are ignored (and reported) without SBN_EXPERIMENTAL=1; the product sources read the environment
    nowhere else."""
    import subprocess
    import sys

    def run(env):
        code = ("import starky_bn254_amd as S\n"
                "try:\n    d = S.api.settings_check(); print('OK', d['ntt_chunk'], d['range_check'], d['ignored'], d['device_chain'], d['host_threads'])\n"
                "except S.SbnError as e:\n    print('ERR', e.code, e)\n")
        e = {k: v for k, v in os.environ.items() if not k.startswith("SBN_")}
        e.update(env, PYTHONPATH=ROOT)
        return subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=120).stdout.strip()
    assert run({}).startswith("OK 0 0 [] -1")
    assert run({"SBN_HOST_THREADS": "3"}) == "OK 0 0 [] -1 3"
    assert run({"SBN_NTT_CHUNK": "32", "SBN_RANGE_CHECK": "1"}) .startswith("OK 0 0 [SBN_NTT_CHUNK,SBN_RANGE_CHECK]")      # ignored, and said so
    assert run({"SBN_EXPERIMENTAL": "1", "SBN_NTT_CHUNK": "32", "SBN_RANGE_CHECK": "1"}).startswith("OK 32 1 []")
    assert run({"SBN_EXPERIMENTAL": "1", "SBN_NTT_CHUNK": "30"}).startswith("ERR -1")
    assert run({"SBN_TRACEGEN_DEVICE_CHAIN": "3"}).startswith("ERR -1")
    assert run({"SBN_TRACEGEN_DEVICE_CHAIN": "2"}).startswith("OK 0 0 [] 2")
    assert run({"SBN_COMM_TIMEOUT_S": "soon"}).startswith("ERR -1")
    assert run({"SBN_HOST_THREADS": "0"}).startswith("ERR -1")
    csrc = os.path.join(ROOT, "starky_bn254_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".cuh", ".hpp")) and f != "settings.hpp":
            src = open(os.path.join(csrc, f)).read()
            hits = [ln for ln in src.splitlines() if "getenv(" in ln and "SBN_DIAG_QUOTIENT_SEGMASK" not in ln]
            assert hits == [], (f, hits)
