#!/usr/bin/env python3
"""Dynamic instruction mix of ONE Poseidon permutation of the leaf sponge (poseidon_permute_fast, poseidon.cuh) from its
hand-scheduled streams, priced with the measured issue costs of tools/microbench/valu_rates.hip
(profiles/r<N>_valu_rates.txt, wall-clock ns per wave-instruction per SIMD at W waves per SIMD).

usage: poseidon_mix.py profiles/r2_valu_rates.txt [waves=2] [valu_per_permutation] > profiles/r<N>_sponge_issue_model.json
The compiler-generated glue (the constant additions left outside the matrix layers, canonicalisation, loads, loop control)
is priced as carry adds; its count is the VALU instructions per permutation the PMC run measured (third argument: SQ_INSTS_VALU
of leaf_absorb_kernel / wave-permutations, default 12618 = the count from the kernel's disassembly, round 3) minus the VALU
instructions of the streams."""
import collections, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "starky_bn254_amd", "csrc")
# how often each stream runs in one permutation (poseidon.cuh poseidon_permute_fast)
# (the sponge's usual permutation: its last layer computes the four capacity rows only, poseidon_permute_keep)
RUNS = {"poseidon_sbox3_asm.inc": 8 * 4, "poseidon_mdsc_a_asm.inc": 7 * 2, "poseidon_mdsc_b_asm.inc": 7 * 2, "poseidon_mdsq8_asm.inc": 2,
        "poseidon_fold3_asm.inc": 7 * 4 + 1 + 6 * 4 + 2 * 4,
        "poseidon_sbox1_asm.inc": 6 * 3 + 2 * 2, "poseidon_mdsrow0_asm.inc": 6 + 2, "poseidon_mds2row0_asm.inc": 6, "poseidon_fold1_asm.inc": 6 * 2 + 2 + 1,
        "poseidon_mds3_asm.inc": 6 * 2, "poseidon_mds2_asm.inc": 2 * 2}
# microbenchmark row that prices an opcode
ROW = {"v_mad_u64_u32": "v_mad_u64_u32 v,v (+vcc)", "v_addc_co_u32": "v_addc_co_u32 sgpr->sgpr' (VOP3)", "v_add_co_u32": "v_add_co_u32 ->sgpr (VOP3)",
       "v_sub_co_u32": "v_add_co_u32 ->sgpr (VOP3)", "v_subbrev_co_u32": "v_addc_co_u32 sgpr->sgpr' (VOP3)", "v_subb_co_u32": "v_addc_co_u32 sgpr->sgpr' (VOP3)",
       "v_cndmask_b32": "v_cndmask_b32 sgpr (VOP3)", "v_mov_b32": "v_mov_b32", "v_lshl_add_u64": "v_lshl_add_u64", "v_add_u32": "v_add_u32",
       "s_nop": "v_mad_u64_u32 + s_nop 0 (2 instr)", "s_load": "s_mov_b32", "s_waitcnt": None}


def table(path):
    t = {}
    for ln in open(path):
        m = re.match(r"(.+?)\s{2,}(\d.*)", ln)
        if not m or ln.startswith("#") or ln.startswith("instruction"):
            continue
        ns = re.findall(r"\(\s*([0-9.]+)\)", m.group(2))
        if len(ns) == 5:
            t[m.group(1).strip()] = [float(x) for x in ns]
    return t


def main():
    rates = table(sys.argv[1])
    waves = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    col = {1: 0, 2: 1, 3: 2, 4: 3, 8: 4}[waves]
    mix = collections.Counter()
    for f, runs in RUNS.items():
        for op in re.findall(r"\b([vs]_[a-z0-9_]+)", open(os.path.join(CSRC, f)).read()):
            op = re.sub(r"_e(32|64)$", "", op)
            op = "s_load" if op.startswith("s_load") else op
            mix[op] += runs
    asm_total = sum(mix.values())
    valu_total = float(sys.argv[3]) if len(sys.argv) > 3 else 12618.0
    asm_valu = sum(c for op, c in mix.items() if op.startswith("v_"))
    glue = max(0, int(round(valu_total - asm_valu)))
    priced = 0.0
    rows = {}
    for op, cnt in sorted(mix.items(), key=lambda kv: -kv[1]):
        row = ROW.get(op, "v_mad_u64_u32 v,v (+vcc)")
        if row is None:
            ns = 0.0
        elif op == "s_nop":
            ns = max(0.0, 2 * rates[row][col] - rates["v_mad_u64_u32 v,v (+vcc)"][col])   # the pad's own share of the pair
        else:
            ns = rates[row][col]
        rows[op] = {"per_permutation": cnt, "ns_per_wave_instruction": round(ns, 3), "priced_as": row}
        priced += cnt * ns
    glue_ns = rates["v_add_co_u32 ->sgpr (VOP3)"][col]
    priced += glue * glue_ns
    total = asm_total + glue
    print(json.dumps({"source": os.path.relpath(sys.argv[1], ROOT), "waves_per_simd": waves, "instructions_per_permutation": total,
                      "hand_scheduled": asm_total, "compiler_glue_priced_as_carry_adds": glue, "mix": rows,
                      "issue_us_per_wave_permutation_per_simd": round(priced / 1e3, 3),
                      "weighted_ns_per_wave_instruction": round(priced / total, 4)}, indent=1))


if __name__ == "__main__":
    main()
