"""starky-bn254 on MI355X: Python host-side mirror of the reference's prover interface.

Thin ctypes layer over the C ABI in include/sbn.h (libsbn254.so, built in-tree by
`make -C starky_bn254_amd/csrc`).  There is NO CPU fallback: importing works without a GPU (so the
symbol table can be checked), but every prove()/commit call fails loudly without a HIP device.
"""
from .api import (  # noqa: F401
    AIR_G1_OP, AIR_G1_EXP, AIR_G2_EXP, AIR_FQ12_EXP, AIR_FQ_EXP, AIR_FQ12_EXP_U64, AIR_MODULAR, AIR_FQ12_MUL, AIR_LOOKUP, AIR_FLAGS, AIR_FLAGS_U64, LookupStark, MyStark, FlagStark, FlagU64Stark, SbnError, StarkConfig, G1Stark, ModularStark, Fq12Stark, G1ExpStark, G2ExpStark, Fq12ExpStark,
    FqExpStark, Fq12ExpU64Stark, Prover, BatchProver, Proof,
    prove, verify_stark_proof, commit_values, eval_constraints_host, poseidon_permute_batch, poseidon_permute_host, field_mul_batch, lib, lib_path, EXPORTS,
)
