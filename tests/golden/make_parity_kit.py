#!/usr/bin/env python3
"""Generates tests/golden/parity_kit/<case>.json: the reference-parity hand-over kit (tests/parity_kit.py).

Each file holds, for one golden case,
  * `inputs`: the seeded inputs as the flat u32 arrays of include/sbn.h (the instance list of an Exp table, the operand
    rows of a single-operation table) -- everything the reference's `generate_trace` needs, so that a maintainer can feed
    the SAME inputs to the Rust reference (integration/rust/starky-bn254-amd/tests/parity.rs turns them into
    `G1ExpIONative` etc.);
  * `trace_sha256` / `public_inputs_sha256` of the witness the oracle's restatement of `generate_trace` builds from them;
  * `stages`: the per-stage digests (tests/parity_kit.py STAGES) of the ORACLE's proof, once per FRI variant
    (`times_x` = plonky2 0.1.x, the default; `plain` = later upstream, include/sbn.h sbn_fri_variant).
The oracle is a restatement (parity unpinned): these values pin GPU == oracle stage by stage today and
become reference-parity vectors the day parity.rs prints the same ones.

    python3 tests/golden/make_parity_kit.py            # the cases that take seconds
    python3 tests/golden/make_parity_kit.py --all      # + G1ExpStark / G2ExpStark / FqExpStark at 2^16 rows (minutes on 8 cores)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402
import parity_kit as K  # noqa: E402

# name -> (table, kind, num_io or rows, seed, inputs(), trace(), slow)
CASES = {
    "mystark_lookup_fixed": ("MyStark", "AIR_LOOKUP", 8, None, None, None, False),
    "mystark_rows512_seed9": ("MyStark", "AIR_LOOKUP", 512, 9, "lookup_inputs", "lookup_trace", False),
    "flagstark_io16_seed8": ("FlagStark", "AIR_FLAGS", 16, 8, "flags_inputs", "flags_trace", False),
    "flagu64stark_io16_seed10": ("FlagU64Stark", "AIR_FLAGS_U64", 16, 10, "flags_u64_inputs", "flags_u64_trace", False),
    "modular_rows512_seed6": ("ModularStark", "AIR_MODULAR", 512, 6, "modular_inputs", "modular_trace", False),
    "g1op_rows512_seed0": ("G1Stark", "AIR_G1_OP", 512, 0, "g1op_inputs", "g1op_trace", False),
    "fq12expu64_io16_seed5": ("Fq12ExpU64Stark", "AIR_FQ12_EXP_U64", 16, 5, "fq12expu64_inputs", "fq12expu64_trace", False),
    "fq12exp_io16_seed3": ("Fq12ExpStark", "AIR_FQ12_EXP", 16, 3, "fq12exp_inputs", "fq12exp_trace", True),
    "g1exp_io128_seed1": ("G1ExpStark", "AIR_G1_EXP", 128, 1, "g1exp_inputs", "g1exp_trace", True),
    "fqexp_io128_seed4": ("FqExpStark", "AIR_FQ_EXP", 128, 4, "fqexp_inputs", "fqexp_trace", True),
    "g2exp_io128_seed2": ("G2ExpStark", "AIR_G2_EXP", 128, 2, "g2exp_inputs", "g2exp_trace", True),
}
INPUT_LAYOUT = {
    "MyStark": "2 x rows: row 0 = the inputs column, row 1 = the table column of MyStark::generate_trace (sbn_generate_trace_lookup).  mystark_lookup_fixed = the reference's own "
               "fixed vectors [6, 3, 1, 1, 0, 0, 0, 0] / 0..7, src/utils/lookup.rs:154-161 (ORACLE ONLY: 8 rows are below the device prover's minimum; run the reference's "
               "test_mystark, lookup.rs:215-229, and compare)",
    "FlagStark": "num_io x 8 u32 limbs (sbn_generate_trace_flags; flags.rs:551-554 draws them at random)",
    "FlagU64Stark": "num_io x 2 u32 = the u64 exponents, low word first (sbn_generate_trace_flags_u64 takes them as u64; flags_u64.rs:427)",
    "ModularStark": "rows x 16 u32 = a[8] b[8] (sbn_generate_trace_modular)",
    "G1Stark": "rows x 32 u32 = a.x[8] a.y[8] b.x[8] b.y[8] (sbn_generate_trace_g1_op)",
    "G1ExpStark": "num_io x 40 u32 = x.x[8] x.y[8] offset.x[8] offset.y[8] exp_val[8] (sbn_generate_trace_g1_exp)",
    "G2ExpStark": "num_io x 72 u32 = x.x.c0 x.x.c1 x.y.c0 x.y.c1 offset.x.c0 offset.x.c1 offset.y.c0 offset.y.c1 exp_val, 8 limbs each (sbn_generate_trace_g2_exp)",
    "Fq12ExpStark": "num_io x 200 u32 = x[12] offset[12] (flat-basis coefficients, 8 limbs each) exp_val[8] (sbn_generate_trace_fq12_exp)",
    "Fq12ExpU64Stark": "num_io x 194 u32 = x[12] offset[12] (8 limbs each) exp_val (low, high) (sbn_generate_trace_fq12_exp_u64)",
    "FqExpStark": "num_io x 24 u32 = x[8] offset[8] exp_val[8] (sbn_generate_trace_fq_exp)",
}


def build_case(name, variants=("times_x", "plain")):
    table, kind, size, seed, fin, ftr, _ = CASES[name]
    if table == "MyStark":        # seed None: no generator, the reference's own fixed input
        ins, tab = O.lookup_fixed_inputs() if seed is None else O.lookup_inputs(size, seed)
        inputs = np.stack([ins, tab]).astype(np.uint32)
        trace, pi = O.lookup_trace(ins, tab), np.zeros(0, dtype=np.uint64)
    else:
        inputs, _ = getattr(O, fin)(size, seed)
        res = getattr(O, ftr)(inputs)
        trace, pi = res if isinstance(res, tuple) else (res, np.zeros(0, dtype=np.uint64))
    is_exp = table.endswith("ExpStark") or table in ("Fq12ExpU64Stark", "FlagStark", "FlagU64Stark")   # tables parametrised by an instance count
    out = {
        "about": "reference-parity hand-over kit: inputs + per-stage digests of the ORACLE's proof (a restatement; parity vs the Rust reference is unpinned)",
        "table": table, "air_kind": int(getattr(O, kind)), "num_io": size if is_exp else 0, "rows": int(trace.shape[1]), "seed": seed,
        "config": "StarkConfig::standard_fast_config: security 100, 2 challenges, rate_bits 1, cap_height 4, pow 16, ConstantArityBits(4, 5), 84 queries",
        "inputs_layout": INPUT_LAYOUT[table], "inputs_shape": [len(inputs), 2] if table == "FlagU64Stark" else list(np.asarray(inputs).shape),
        "inputs_u32": [int(x) for x in (np.asarray(inputs, dtype=np.uint64).view(np.uint32) if table == "FlagU64Stark" else np.asarray(inputs, dtype=np.uint32)).reshape(-1)],
        "num_columns": int(trace.shape[0]),
        "trace_sha256": hashlib.sha256(trace.tobytes()).hexdigest(),
        "public_inputs_sha256": hashlib.sha256(np.asarray(pi, dtype=np.uint64).tobytes()).hexdigest(),
        "stages": {},
    }
    for v in variants:
        try:
            O.set_final_poly_times_x(v == "times_x")
            w, secs = O.prove(getattr(O, kind), out["num_io"], trace, pi)
            assert O.verify(getattr(O, kind), out["num_io"], w)[0] == 0
        finally:
            O.set_final_poly_times_x(True)
        d = K.stage_digests(w, O.poseidon_permute)
        assert d["pow_ok"]
        out["stages"][v] = d
        print(f"{name} [{v}]: oracle prove {secs:.1f} s, proof sha {d['proof']['sha256'][:16]}")
    return out


def main():
    os.makedirs(os.path.join(HERE, "parity_kit"), exist_ok=True)
    names = [a for a in sys.argv[1:] if not a.startswith("--")] or [n for n, c in CASES.items() if "--all" in sys.argv or not c[6]]
    for name in names:
        kit = build_case(name)
        with open(os.path.join(HERE, "parity_kit", name + ".json"), "w") as f:
            json.dump(kit, f, separators=(",", ":"))
        print("wrote parity_kit/" + name + ".json")


if __name__ == "__main__":
    main()
