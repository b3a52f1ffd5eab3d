#!/usr/bin/env python3
"""Generates the committed golden fixtures under tests/golden/.

What pins what (the reference itself has NO golden vectors -- SURVEY.md section 4 / 8c):
  poseidon_kat.json      plonky2 `poseidon_goldilocks.rs::test_vectors` (inputs 0^12, 0..11, (-1)^12) and the
                         first 16 ALL_ROUND_CONSTANTS -- dependency-published values, typed in here by hand;
                         the oracle must reproduce them from the ChaCha8(0) derivation.
  modular_kat.json       generate_modular_op / generate_modular_zero outputs computed HERE with Python big
                         integers following src/modular/modular.rs:38-100 line by line (independent of the C++).
  g1_scalar_mult.json    x*s+offset for seeded inputs computed with Python-int affine arithmetic.
  lookup_fixed.json      permuted_cols on the reference's only fixed input (src/utils/lookup.rs:154-161).
  flags_native.json      the reference's test_flag_native property inputs (flags.rs:334-369) for one exponent.
  proof_digests.json     sha256 of the ORACLE's canonical proof words for seeded G1Stark-512 / ModularStark-512 / Fq12Stark-512 / G1ExpStark-2^16
                         traces: pins GPU == oracle without re-running the 100 s CPU prover.
Run from the repo root:  python3 tests/golden/make_golden.py [--with-g1exp] [--with-g2exp] [--with-fq12exp] [--with-fqexp] [--with-fq12expu64]
"""
import hashlib, json, os, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

P = O.BN_P


def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def limbs16(v, n):
    return [(v >> (16 * i)) & 0xFFFF for i in range(n)]


def modular_witness_py(pol_input, zero_mode):
    """src/modular/modular.rs:38-100 (and modular_zero.rs:33-80) with Python ints."""
    inp = sum(c << (16 * i) for i, c in enumerate(pol_input))
    out = inp % P
    if zero_mode:
        assert out == 0
    quot = (inp - out) // P
    sign = -1 if quot < 0 else 1
    ml = limbs16(P, 16)
    quot_limbs = [sign * l for l in limbs16(abs(quot), 17)]
    constr = list(pol_input) + [0]
    if not zero_mode:
        for i, l in enumerate(limbs16(out, 16)):
            constr[i] -= l
    for i in range(17):
        for j in range(16):
            constr[i + j] -= quot_limbs[i] * ml[j]
    aux = [0] * 32
    aux[0] = -(constr[0] >> 16)
    for d in range(1, 31):
        aux[d] = (aux[d - 1] - constr[d]) >> 16
    aux = [a + (1 << 29) for a in aux]
    return {
        "output": limbs16(out, 16), "out_aux_red": limbs16((1 << 256) - P + out, 16),
        "quot_abs": limbs16(abs(quot), 17), "sign": sign,
        "aux_lo": [a & 0xFFFF for a in aux[:31]], "aux_hi": [(a >> 16) & 0xFFFF for a in aux[:31]],
    }


def conv(a, b):
    r = [0] * 31
    for i in range(16):
        for j in range(16):
            r[i + j] += a[i] * b[j]
    return r


def main():
    dump("poseidon_kat.json", {
        "source": "plonky2 plonky2/src/hash/poseidon_goldilocks.rs test_vectors; poseidon.rs ALL_ROUND_CONSTANTS[0..16]",
        "round_constants_first16": [
            "0xb585f766f2144405", "0x7746a55f43921ad7", "0xb2fb0d31cee799b4", "0x0f6760a4803427d7",
            "0xe10d666650f4e012", "0x8cae14cb07d09bf1", "0xd438539c95f63e9f", "0xef781c7ce35b4c3d",
            "0xcdc4a239b0c44426", "0x277fa208bf337bff", "0xe17653a29da578a1", "0xc54302f225db2c76",
            "0x86287821f722c881", "0x59cd1a8a41c18e55", "0xc3b919ad495dc574", "0xa484c4c5ef6a0781"],
        "vectors": [
            {"input": ["0x0"] * 12, "output": [
                "0x3c18a9786cb0b359", "0xc4055e3364a246c3", "0x7953db0ab48808f4", "0xc71603f33a1144ca",
                "0xd7709673896996dc", "0x46a84e87642f44ed", "0xd032648251ee0b3c", "0x1c687363b207df62",
                "0xdf8565563e8045fe", "0x40f5b37ff4254dae", "0xd070f637b431067c", "0x1792b1c4342109d7"]},
            {"input": [hex(i) for i in range(12)], "output": [
                "0xd64e1e3efc5b8e9e", "0x53666633020aaa47", "0xd40285597c6a8825", "0x613a4f81e81231d2",
                "0x414754bfebd051f0", "0xcb1f8980294a023f", "0x6eb2a9e4d54a9d0f", "0x1902bc3af467e056",
                "0xf045d5eafdc6021f", "0xe4150f77caaa3be5", "0xc9bfd01d39b50cce", "0x5c0a27fcb0e1459b"]},
            {"input": [hex(O.GL_P - 1)] * 12, "output": [
                "0xbe0085cfc57a8357", "0xd95af71847d05c09", "0xcf55a13d33c1c953", "0x95803a74f4530e82",
                "0xfcd99eb30a135df1", "0xe095905e913a3029", "0xde0392461b42919b", "0x7d3260e24e81d031",
                "0x10d3d0465d9deaa0", "0xa87571083dfc2a47", "0xe18263681e9958f8", "0xe28e96f1ae5e60d3"]},
        ]})

    rng = np.random.default_rng(2024)
    kats = []
    for t in range(8):
        a = int.from_bytes(rng.bytes(32), "little") % P
        b = int.from_bytes(rng.bytes(32), "little") % P
        c = int.from_bytes(rng.bytes(32), "little") % P
        al, bl, cl = limbs16(a, 16), limbs16(b, 16), limbs16(c, 16)
        if t % 2 == 0:   # a*b - c  (signed input, like lambda^2 - x1 - x2)
            pol = conv(al, bl)
            for i in range(16):
                pol[i] -= 3 * cl[i]
            kats.append({"pol_input": pol, "zero_mode": 0, "expect": modular_witness_py(pol, False)})
        else:            # a*b - (a*b mod p) = 0 mod p
            pol = conv(al, bl)
            ab = a * b % P
            for i, l in enumerate(limbs16(ab, 16)):
                pol[i] -= l
            kats.append({"pol_input": pol, "zero_mode": 1, "expect": modular_witness_py(pol, True)})
    dump("modular_kat.json", {"source": "python big-int restatement of src/modular/modular.rs:38-100", "cases": kats})

    cases = []
    for seed in (1, 7):
        ios, native = O.g1exp_inputs(4, seed)
        outs = []
        for (x, off, s) in native:
            r = O.g1_add(O.g1_mul(x, s), off)
            outs.append([hex(r[0]), hex(r[1])])
        cases.append({"seed": seed, "ios": ios.tolist(), "outputs": outs})
    dump("g1_scalar_mult.json", {"source": "python-int affine BN254 arithmetic: x*s + offset", "cases": cases})

    ios, native = O.g2exp_inputs(3, 2)
    outs = []
    for (x, off, sc) in native:
        r = O.g2_add(O.g2_mul(x, sc), off)
        outs.append([hex(r[0][0]), hex(r[0][1]), hex(r[1][0]), hex(r[1][1])])
    dump("g2_scalar_mult.json", {"source": "python-int affine arithmetic on the BN254 twist over Fq2: x*s + offset",
                                 "seed": 2, "ios": ios.tolist(), "outputs": outs})

    # lookup.rs:154-161 fixed input; expected columns from a python restatement of lookup.rs:60-111
    inputs = [6, 3, 1, 1, 0, 0, 0, 0]
    table = list(range(8))
    si, st = sorted(inputs), sorted(table)
    n = 8
    perm = [0] * n
    unused_inds, unused_vals = [], []
    i = j = 0
    while j < n and i < n:
        if si[i] > st[j]:
            unused_vals.append(st[j]); j += 1
        elif si[i] < st[j]:
            if unused_vals:
                perm[i] = unused_vals.pop()
            else:
                unused_inds.append(i)
            i += 1
        else:
            perm[i] = st[j]; i += 1; j += 1
    unused_vals += st[j:]
    unused_inds += list(range(i, n))
    for ind, val in zip(unused_inds, unused_vals):
        perm[ind] = val
    dump("lookup_fixed.json", {"source": "src/utils/lookup.rs:154-161", "inputs": inputs, "table": table,
                               "sorted_inputs": si, "permuted_table": perm})

    exp = [int(v) for v in np.random.default_rng(11).integers(0, 1 << 32, size=8, dtype=np.uint64)]
    dump("flags_native.json", {"source": "src/utils/flags.rs:334-369 test_flag_native", "limbs": exp,
                               "bits_lsb_first": [(exp[i // 32] >> (i % 32)) & 1 for i in range(256)]})

    digests = {}
    path = os.path.join(HERE, "proof_digests.json")
    if os.path.exists(path):
        digests = json.load(open(path))
    for rows, seed in ((512, 0), (1024, 3)):
        pts, _ = O.g1op_inputs(rows, seed)
        tr = O.g1op_trace(pts)
        w, secs = O.prove(O.AIR_G1_OP, 0, tr, np.zeros(0, dtype=np.uint64))
        assert O.verify(O.AIR_G1_OP, 0, w)[0] == 0
        digests[f"g1op_rows{rows}_seed{seed}"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "proof_words": int(len(w)),
            "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1])}
    # the reference's single-operation test tables (ModularStark modular.rs:361-537, Fq12Stark fq12/mul.rs:355-517), 512 rows as there
    for key, kind, inputs, gen in (("modular_rows512_seed6", O.AIR_MODULAR, O.modular_inputs, O.modular_trace),
                                   ("fq12mul_rows512_seed7", O.AIR_FQ12_MUL, O.fq12mul_inputs, O.fq12mul_trace)):
        ops, _ = inputs(512, int(key[-1]))
        tr = gen(ops)
        w, secs = O.prove(kind, 0, tr, np.zeros(0, dtype=np.uint64))
        assert O.verify(kind, 0, w)[0] == 0
        digests[key] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "proof_words": int(len(w)),
            "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1])}
    if "--with-g1exp" in sys.argv:
        ios, _ = O.g1exp_inputs(128, 1)
        tr, pi = O.g1exp_trace(ios)
        w, secs = O.prove(O.AIR_G1_EXP, 128, tr, pi)
        assert O.verify(O.AIR_G1_EXP, 128, w)[0] == 0
        digests["g1exp_io128_seed1"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest(),
            "proof_words": int(len(w)), "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1 - len(pi)]), "oracle_prove_seconds_8core": secs}
    if "--with-g2exp" in sys.argv:          # ~3 minutes on 8 cores
        ios, _ = O.g2exp_inputs(128, 2)
        tr, pi = O.g2exp_trace(ios)
        w, secs = O.prove(O.AIR_G2_EXP, 128, tr, pi)
        assert O.verify(O.AIR_G2_EXP, 128, w)[0] == 0
        digests["g2exp_io128_seed2"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest(),
            "proof_words": int(len(w)), "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1 - len(pi)]), "oracle_prove_seconds_8core": secs}
    if "--with-fq12exp" in sys.argv:        # ~1 minute on 8 cores
        ios, _ = O.fq12exp_inputs(16, 3)
        tr, pi = O.fq12exp_trace(ios)
        w, secs = O.prove(O.AIR_FQ12_EXP, 16, tr, pi)
        assert O.verify(O.AIR_FQ12_EXP, 16, w)[0] == 0
        digests["fq12exp_io16_seed3"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest(),
            "proof_words": int(len(w)), "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1 - len(pi)]), "oracle_prove_seconds_8core": secs}
    if "--with-fqexp" in sys.argv:          # ~1 minute on 8 cores
        ios, _ = O.fqexp_inputs(128, 4)
        tr, pi = O.fqexp_trace(ios)
        w, secs = O.prove(O.AIR_FQ_EXP, 128, tr, pi)
        assert O.verify(O.AIR_FQ_EXP, 128, w)[0] == 0
        digests["fqexp_io128_seed4"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest(),
            "proof_words": int(len(w)), "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1 - len(pi)]), "oracle_prove_seconds_8core": secs}
    if "--with-fq12expu64" in sys.argv:     # ~20 s on 8 cores
        ios, _ = O.fq12expu64_inputs(16, 5)
        tr, pi = O.fq12expu64_trace(ios)
        w, secs = O.prove(O.AIR_FQ12_EXP_U64, 16, tr, pi)
        assert O.verify(O.AIR_FQ12_EXP_U64, 16, w)[0] == 0
        digests["fq12expu64_io16_seed5"] = {
            "trace_sha256": hashlib.sha256(tr.tobytes()).hexdigest(), "pi_sha256": hashlib.sha256(pi.tobytes()).hexdigest(),
            "proof_words": int(len(w)), "proof_sha256": hashlib.sha256(w.astype("<u8").tobytes()).hexdigest(),
            "trace_cap0": [int(x) for x in w[12:16]], "pow_witness": int(w[-1 - len(pi)]), "oracle_prove_seconds_8core": secs}
    dump("proof_digests.json", digests)


if __name__ == "__main__":
    main()
