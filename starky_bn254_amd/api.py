"""Host-side mirror of the reference's operator interface for the G1 prover path.

Names follow the reference (src/curves/g1/exp.rs:232-328, :811-826):
    stark = G1ExpStark(num_io);  config = stark.config()
    trace = stark.generate_trace(inputs);  pi = stark.generate_public_inputs(inputs)
    proof = prove(stark, config, trace, pi);  verify_stark_proof(stark, proof, config)
Errors surface as SbnError (the reference returns anyhow::Result and callers unwrap).
"""
import ctypes as C
import os
import numpy as np

AIR_G1_OP = 1
AIR_MODULAR = 7
AIR_FQ12_MUL = 8
AIR_LOOKUP = 9
AIR_FLAGS = 10
AIR_FLAGS_U64 = 11
AIR_G1_EXP = 2
AIR_G2_EXP = 3
AIR_FQ12_EXP = 4
AIR_FQ_EXP = 5
AIR_FQ12_EXP_U64 = 6

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

# every symbol include/sbn.h declares
EXPORTS = [
    "sbn_version", "sbn_last_error", "sbn_device_count", "sbn_set_device", "sbn_set_thread_device", "sbn_standard_fast_config",
    "sbn_air_num_columns", "sbn_air_num_public_inputs", "sbn_air_num_permutation_zs", "sbn_air_num_constraints",
    "sbn_generate_trace_g1_exp", "sbn_generate_trace_g2_exp", "sbn_generate_trace_fq12_exp", "sbn_generate_trace_fq_exp", "sbn_generate_trace_fq12_exp_u64",
    "sbn_generate_trace_g1_op", "sbn_generate_trace_modular", "sbn_generate_trace_fq12_mul", "sbn_generate_trace_lookup", "sbn_generate_trace_flags", "sbn_generate_trace_flags_u64",
    "sbn_prover_create", "sbn_prover_destroy", "sbn_prover_load_trace", "sbn_prover_load_trace_device",
    "sbn_prover_prove", "sbn_prover_stage_times", "sbn_prover_stage_name", "sbn_prover_describe", "sbn_settings_check", "sbn_prover_trace_device_ptr",
    "sbn_prover_generate_trace", "sbn_prover_read_trace",
    "sbn_batch_prover_create", "sbn_batch_prover_prove_ios", "sbn_batch_prover_destroy",
    "sbn_prove", "sbn_proof_num_words", "sbn_proof_words", "sbn_proof_serialize", "sbn_proof_degree_bits",
    "sbn_proof_free", "sbn_verify", "sbn_commit_values", "sbn_poseidon_permute_batch", "sbn_poseidon_permute_host", "sbn_field_mul_batch",
    "sbn_eval_constraints_host", "sbn_host_curve_chains", "sbn_split_exchange_bytes", "sbn_split_prover_create", "sbn_split_prover_destroy", "sbn_split_prover_generate_trace",
    "sbn_split_prover_load_trace", "sbn_split_prover_prove", "sbn_split_prover_stage_times",
    "sbn_abi_version", "sbn_rccl_unique_id", "sbn_rccl_comm_create", "sbn_rccl_comm_destroy",
    "sbn_local_comm_create", "sbn_local_comm_abort", "sbn_local_comm_destroy", "sbn_comm_selftest",
]


class SbnError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"sbn error {code}: {msg}")
        self.code = code


class _AirDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("num_io", C.c_uint32)]


class _Config(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("security_bits", "num_challenges", "rate_bits", "cap_height",
                                            "proof_of_work_bits", "fri_arity_bits", "fri_final_poly_bits", "num_query_rounds",
                                            "fri_variant")]


FRI_DEFAULT, FRI_TIMES_X, FRI_PLAIN = 0, 1, 2   # sbn_fri_variant (include/sbn.h)


def lib_path():
    """The in-tree HIP library; SBN_LIB selects another build of the same ABI (A/B measurements)."""
    return os.environ.get("SBN_LIB") or os.path.join(_HERE, "libsbn254.so")


def lib():
    """Loads libsbn254.so; raises if the HIP extension has not been built (no fallback)."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise SbnError(-100, f"{p} is missing: build it with `make -C starky_bn254_amd/csrc` (there is no CPU fallback)")
        L = C.CDLL(p)
        vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
        L.sbn_version.restype = C.c_char_p
        L.sbn_last_error.restype = C.c_char_p
        L.sbn_prover_stage_name.restype = C.c_char_p
        L.sbn_prover_stage_name.argtypes = [C.c_int]
        for f in ("sbn_air_num_columns", "sbn_air_num_public_inputs", "sbn_air_num_constraints"):
            getattr(L, f).restype = sz
            getattr(L, f).argtypes = [C.POINTER(_AirDesc)]
        L.sbn_air_num_permutation_zs.restype = sz
        L.sbn_air_num_permutation_zs.argtypes = [C.POINTER(_AirDesc), C.POINTER(_Config)]
        L.sbn_standard_fast_config.argtypes = [C.POINTER(_Config)]
        L.sbn_generate_trace_g1_exp.argtypes = [vp, sz, vp, vp]
        L.sbn_generate_trace_g2_exp.argtypes = [vp, sz, vp, vp]
        L.sbn_generate_trace_fq12_exp.argtypes = [vp, sz, vp, vp]
        L.sbn_generate_trace_fq_exp.argtypes = [vp, sz, vp, vp]
        L.sbn_generate_trace_fq12_exp_u64.argtypes = [vp, sz, vp, vp]
        L.sbn_generate_trace_g1_op.argtypes = [vp, sz, vp]
        L.sbn_generate_trace_modular.argtypes = [vp, sz, vp]
        L.sbn_generate_trace_fq12_mul.argtypes = [vp, sz, vp]
        L.sbn_generate_trace_lookup.argtypes = [vp, vp, sz, vp]
        L.sbn_generate_trace_flags.argtypes = [vp, sz, vp]
        L.sbn_generate_trace_flags_u64.argtypes = [vp, sz, vp]
        L.sbn_prover_create.argtypes = [C.POINTER(_AirDesc), C.POINTER(_Config), u32, C.POINTER(vp)]
        L.sbn_prover_destroy.argtypes = [vp]
        L.sbn_prover_load_trace.argtypes = [vp, vp, vp, sz]
        L.sbn_prover_load_trace_device.argtypes = [vp, vp, vp, sz]
        L.sbn_prover_prove.argtypes = [vp, C.POINTER(vp)]
        L.sbn_prover_stage_times.argtypes = [vp, C.POINTER(C.c_float), C.c_int]
        L.sbn_prover_trace_device_ptr.restype = vp
        L.sbn_prover_trace_device_ptr.argtypes = [vp]
        L.sbn_prover_generate_trace.argtypes = [vp, vp, sz, vp]
        L.sbn_prover_read_trace.argtypes = [vp, vp]
        L.sbn_batch_prover_create.argtypes = [C.POINTER(_AirDesc), C.POINTER(_Config), u32, u32, C.POINTER(vp)]
        L.sbn_batch_prover_prove_ios.argtypes = [vp, vp, sz, sz, sz, C.POINTER(vp)]
        L.sbn_batch_prover_destroy.argtypes = [vp]
        L.sbn_prove.argtypes = [C.POINTER(_AirDesc), C.POINTER(_Config), vp, u32, vp, sz, C.POINTER(vp)]
        L.sbn_proof_num_words.restype = sz
        L.sbn_proof_num_words.argtypes = [vp]
        L.sbn_proof_words.restype = C.POINTER(C.c_uint64)
        L.sbn_proof_words.argtypes = [vp]
        L.sbn_proof_serialize.restype = sz
        L.sbn_proof_serialize.argtypes = [vp, vp, sz]
        L.sbn_proof_degree_bits.restype = u32
        L.sbn_proof_degree_bits.argtypes = [vp]
        L.sbn_proof_free.argtypes = [vp]
        L.sbn_verify.argtypes = [C.POINTER(_AirDesc), C.POINTER(_Config), vp, sz]
        L.sbn_commit_values.argtypes = [vp, sz, sz, u32, u32, vp, vp, vp]
        L.sbn_poseidon_permute_batch.argtypes = [vp, sz]
        L.sbn_poseidon_permute_host.argtypes = [vp, sz, C.c_int]
        L.sbn_field_mul_batch.argtypes = [vp, vp, vp, sz, C.c_int]
        L.sbn_set_device.argtypes = [C.c_int]
        L.sbn_set_thread_device.argtypes = [C.c_int]
        L.sbn_host_curve_chains.argtypes = [C.c_int, vp, sz, vp, vp, C.c_int]
        L.sbn_prover_describe.argtypes = [vp, C.c_char_p, sz]
        L.sbn_settings_check.argtypes = [C.c_char_p, sz]
        _LIB = L
    return _LIB


def _check(rc):
    if rc != 0:
        raise SbnError(rc, lib().sbn_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class StarkConfig:
    """starky `StarkConfig` (+ FriConfig); `standard_fast_config()` as in exp.rs:250-253."""

    def __init__(self):
        self._c = _Config()
        lib().sbn_standard_fast_config(C.byref(self._c))

    @staticmethod
    def standard_fast_config(num_columns=None, num_public_inputs=None):
        return StarkConfig()

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_c"), name)

    def __setattr__(self, name, value):
        if name != "_c" and any(name == f[0] for f in _Config._fields_):
            setattr(self._c, name, value)
        else:
            object.__setattr__(self, name, value)


class _Stark:
    kind = 0

    def __init__(self, num_io=0):
        self.num_io = num_io
        self._d = _AirDesc(self.kind, num_io)

    def config(self):
        return StarkConfig.standard_fast_config(self.num_columns, self.num_public_inputs)

    @property
    def num_columns(self):
        return lib().sbn_air_num_columns(C.byref(self._d))

    @property
    def num_public_inputs(self):
        return lib().sbn_air_num_public_inputs(C.byref(self._d))

    def num_permutation_zs(self, config=None):
        config = config or StarkConfig()
        return lib().sbn_air_num_permutation_zs(C.byref(self._d), C.byref(config._c))

    @property
    def num_constraints(self):
        return lib().sbn_air_num_constraints(C.byref(self._d))

    def constraint_degree(self):
        return 3


class G1Stark(_Stark):
    """Reference `G1Stark` (src/curves/g1/muladd.rs:462-624): one affine add per row."""
    kind = AIR_G1_OP

    def __init__(self):
        super().__init__(0)

    def generate_trace(self, pts):
        """pts: (rows, 32) uint32 = a.x a.y b.x b.y as 8xu32 LE limbs -> column-major (ncols, rows) uint64."""
        pts = np.ascontiguousarray(pts, dtype=np.uint32)
        rows = pts.shape[0]
        trace = np.zeros((self.num_columns, rows), dtype=np.uint64)
        _check(lib().sbn_generate_trace_g1_op(_ptr(pts), rows, _ptr(trace)))
        return trace


class ModularStark(_Stark):
    """Reference `ModularStark` (src/modular/modular.rs:361-537): one a * b mod p per row -- the reference's test table for the modular gadget."""
    kind = AIR_MODULAR

    def __init__(self):
        super().__init__(0)

    def generate_trace(self, ops):
        """ops: (rows, 16) uint32 = a b as 8xu32 LE limbs (both < p) -> column-major (ncols, rows) uint64."""
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        rows = ops.shape[0]
        assert ops.shape == (rows, 16)
        trace = np.zeros((self.num_columns, rows), dtype=np.uint64)
        _check(lib().sbn_generate_trace_modular(_ptr(ops), rows, _ptr(trace)))
        return trace


class LookupStark(_Stark):
    """Reference `MyStark` (src/utils/lookup.rs:136-213), the unit-test table of the lookup argument: inputs, table, permuted
    inputs, permuted table; any power-of-two height >= 512 on the device (the reference's own 8-row instance: oracle only)."""
    kind = AIR_LOOKUP

    def __init__(self):
        super().__init__(0)

    def generate_trace(self, inputs, table):
        """inputs, table: (rows,) uint64 canonical field elements, every input value present in the table -> (4, rows) uint64."""
        inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
        table = np.ascontiguousarray(table, dtype=np.uint64)
        assert inputs.shape == table.shape and inputs.ndim == 1
        trace = np.zeros((4, inputs.shape[0]), dtype=np.uint64)
        _check(lib().sbn_generate_trace_lookup(_ptr(inputs), _ptr(table), inputs.shape[0], _ptr(trace)))
        return trace


MyStark = LookupStark   # the reference's name for it (a test-module local)


class FlagStark(_Stark):
    """Reference `FlagStark` (src/utils/flags.rs:379-547), the unit-test table of the exponent-bit flags: num_io inputs of 8 u32
    limbs, 512 rows each; no permutation pairs, so its proofs carry no permutation-Z commitment."""
    kind = AIR_FLAGS

    def generate_trace(self, limbs):
        """limbs: (num_io, 8) uint32 -> (17 + 4 num_io, 512 num_io) uint64."""
        limbs = np.ascontiguousarray(limbs, dtype=np.uint32)
        assert limbs.shape == (self.num_io, 8)
        trace = np.zeros((self.num_columns, 512 * self.num_io), dtype=np.uint64)
        _check(lib().sbn_generate_trace_flags(_ptr(limbs), self.num_io, _ptr(trace)))
        return trace


class FlagU64Stark(_Stark):
    """The `FlagStark` of src/fields/fq12_u64/flags_u64.rs:289-420 (u64 exponents, 128 rows per input, no permutation pairs)."""
    kind = AIR_FLAGS_U64

    def generate_trace(self, exps):
        """exps: (num_io,) uint64 < p -> (7 + 4 num_io, 128 num_io) uint64."""
        exps = np.ascontiguousarray(exps, dtype=np.uint64)
        assert exps.shape == (self.num_io,)
        trace = np.zeros((self.num_columns, 128 * self.num_io), dtype=np.uint64)
        _check(lib().sbn_generate_trace_flags_u64(_ptr(exps), self.num_io, _ptr(trace)))
        return trace


class Fq12Stark(_Stark):
    """Reference `Fq12Stark` (src/fields/fq12/mul.rs:355-517): one Fq12 product per row -- the reference's test table for eval_fq12_mul."""
    kind = AIR_FQ12_MUL

    def __init__(self):
        super().__init__(0)

    def generate_trace(self, ops):
        """ops: (rows, 192) uint32 = x[12] y[12] (flat-basis coefficients < p, 8xu32 LE limbs each)."""
        ops = np.ascontiguousarray(ops, dtype=np.uint32)
        rows = ops.shape[0]
        assert ops.shape == (rows, 192)
        trace = np.zeros((self.num_columns, rows), dtype=np.uint64)
        _check(lib().sbn_generate_trace_fq12_mul(_ptr(ops), rows, _ptr(trace)))
        return trace


class G1ExpStark(_Stark):
    """Reference `G1ExpStark` (src/curves/g1/exp.rs:232-742): 512 rows per scalar multiplication."""
    kind = AIR_G1_EXP

    def generate_trace_and_public_inputs(self, ios):
        """ios: (num_io, 40) uint32 = x.x x.y offset.x offset.y exp_val (8xu32 LE limbs each)."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        assert ios.shape == (self.num_io, 40)
        trace = np.zeros((self.num_columns, 512 * self.num_io), dtype=np.uint64)
        pi = np.zeros(self.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_generate_trace_g1_exp(_ptr(ios), self.num_io, _ptr(trace), _ptr(pi)))
        return trace, pi

    def generate_trace(self, ios):
        return self.generate_trace_and_public_inputs(ios)[0]

    def generate_public_inputs(self, ios):
        return self.generate_trace_and_public_inputs(ios)[1]


class G2ExpStark(_Stark):
    """Reference `G2ExpStark` (src/curves/g2/exp.rs:248-807): G1ExpStark's machine over Fq2 coordinates."""
    kind = AIR_G2_EXP

    def generate_trace_and_public_inputs(self, ios):
        """ios: (num_io, 72) uint32 = x.x.c0 x.x.c1 x.y.c0 x.y.c1 offset.(x.c0 x.c1 y.c0 y.c1) exp_val."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        assert ios.shape == (self.num_io, 72)
        trace = np.zeros((self.num_columns, 512 * self.num_io), dtype=np.uint64)
        pi = np.zeros(self.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_generate_trace_g2_exp(_ptr(ios), self.num_io, _ptr(trace), _ptr(pi)))
        return trace, pi

    def generate_trace(self, ios):
        return self.generate_trace_and_public_inputs(ios)[0]

    def generate_public_inputs(self, ios):
        return self.generate_trace_and_public_inputs(ios)[1]


class Fq12ExpStark(_Stark):
    """Reference `Fq12ExpStark` (src/fields/fq12/exp.rs:223-605): offset * x^e in Fq12, 512 rows per instance."""
    kind = AIR_FQ12_EXP

    def generate_trace_and_public_inputs(self, ios):
        """ios: (num_io, 200) uint32 = x[12] offset[12] (flat-basis coefficients, 8xu32 LE limbs each) exp_val[8]."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        assert ios.shape == (self.num_io, 200)
        trace = np.zeros((self.num_columns, 512 * self.num_io), dtype=np.uint64)
        pi = np.zeros(self.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_generate_trace_fq12_exp(_ptr(ios), self.num_io, _ptr(trace), _ptr(pi)))
        return trace, pi

    def generate_trace(self, ios):
        return self.generate_trace_and_public_inputs(ios)[0]

    def generate_public_inputs(self, ios):
        return self.generate_trace_and_public_inputs(ios)[1]


class FqExpStark(_Stark):
    """Reference `FqExpStark` (src/fields/fq/exp.rs:193-582): offset * x^e in the base field Fq, 512 rows per instance."""
    kind = AIR_FQ_EXP

    def generate_trace_and_public_inputs(self, ios):
        """ios: (num_io, 24) uint32 = x offset exp_val (8xu32 LE limbs each)."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        assert ios.shape == (self.num_io, 24)
        trace = np.zeros((self.num_columns, 512 * self.num_io), dtype=np.uint64)
        pi = np.zeros(self.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_generate_trace_fq_exp(_ptr(ios), self.num_io, _ptr(trace), _ptr(pi)))
        return trace, pi

    def generate_trace(self, ios):
        return self.generate_trace_and_public_inputs(ios)[0]

    def generate_public_inputs(self, ios):
        return self.generate_trace_and_public_inputs(ios)[1]


class Fq12ExpU64Stark(_Stark):
    """Reference `Fq12ExpU64Stark` (src/fields/fq12_u64/exp_u64.rs:243-571): offset * x^e in Fq12 for a u64 exponent,
    128 rows per instance."""
    kind = AIR_FQ12_EXP_U64
    rows_per_instance = 128

    def generate_trace_and_public_inputs(self, ios):
        """ios: (num_io, 194) uint32 = x[12] offset[12] (8xu32 LE limbs each) exp_val (low, high)."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        assert ios.shape == (self.num_io, 194)
        trace = np.zeros((self.num_columns, 128 * self.num_io), dtype=np.uint64)
        pi = np.zeros(self.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_generate_trace_fq12_exp_u64(_ptr(ios), self.num_io, _ptr(trace), _ptr(pi)))
        return trace, pi

    def generate_trace(self, ios):
        return self.generate_trace_and_public_inputs(ios)[0]

    def generate_public_inputs(self, ios):
        return self.generate_trace_and_public_inputs(ios)[1]


class Proof:
    """StarkProofWithPublicInputs as canonical proof words (layout: include/sbn.h)."""

    def __init__(self, words, degree_bits):
        self.words = words
        self.degree_bits = degree_bits

    def to_bytes(self):
        return self.words.astype("<u8").tobytes()

    def recover_degree_bits(self, config=None):
        return self.degree_bits

    def fields(self):
        """The proof as the reference's struct tree (starky proof.rs `StarkProof` / `StarkOpeningSet`, plonky2
        fri/proof.rs `FriProof`): a dict of numpy views into the words, read in the order include/sbn.h documents.
        The Rust shim's `proof_from_words` (integration/rust/starky-bn254-amd/src/convert.rs) reads the same way."""
        w = self.words
        if len(w) < 12 or int(w[0]) != 0x31564F5250424E53:
            raise ValueError("not a proof of this library")
        (degree_bits, ncol, nz, nq, npi, cap_h, rate_bits, nlayers, arity_bits, fpl, nqueries) = (int(x) for x in w[1:12])
        pos = [12]

        def take(n, shape=None):
            if pos[0] + n > len(w):
                raise ValueError("proof words truncated")
            v = w[pos[0]:pos[0] + n]
            pos[0] += n
            return v.reshape(shape) if shape else v

        capn, lde_bits = 1 << cap_h, degree_bits + rate_bits
        out = {"degree_bits": degree_bits, "trace_cap": take(4 * capn, (capn, 4))}
        out["permutation_zs_cap"] = take(4 * capn, (capn, 4)) if nz else None
        out["quotient_polys_cap"] = take(4 * capn, (capn, 4))
        op = {"local_values": take(2 * ncol, (ncol, 2)), "next_values": take(2 * ncol, (ncol, 2))}
        op["permutation_zs"] = take(2 * nz, (nz, 2)) if nz else None
        op["permutation_zs_next"] = take(2 * nz, (nz, 2)) if nz else None
        op["quotient_polys"] = take(2 * nq, (nq, 2))
        out["openings"] = op
        fri = {"commit_phase_merkle_caps": [take(4 * capn, (capn, 4)) for _ in range(nlayers)], "query_round_proofs": []}
        widths = [ncol] + ([nz] if nz else []) + [nq]
        for _ in range(nqueries):
            initial = [(take(wd), take(4 * (lde_bits - cap_h), (lde_bits - cap_h, 4))) for wd in widths]
            steps, bits = [], lde_bits
            for _ in range(nlayers):
                evals = take(2 << arity_bits, (1 << arity_bits, 2))
                bits -= arity_bits
                steps.append({"evals": evals, "merkle_proof": take(4 * (bits - cap_h), (bits - cap_h, 4))})
            fri["query_round_proofs"].append({"initial_trees_proof": initial, "steps": steps})
        fri["final_poly"] = take(2 * fpl, (fpl, 2))
        fri["pow_witness"] = int(take(1)[0])
        out["opening_proof"] = fri
        out["public_inputs"] = take(npi)
        if pos[0] != len(w):
            raise ValueError("trailing words after the proof")
        return out


def _take_proof(h):
    L = lib()
    n = L.sbn_proof_num_words(h)
    words = np.ctypeslib.as_array(L.sbn_proof_words(h), shape=(n,)).copy()
    db = L.sbn_proof_degree_bits(h)
    L.sbn_proof_free(h)
    return Proof(words, db)


def settings_check():
    """The SBN_* switches of this process as a prover created now would resolve them (csrc/settings.hpp), as a dict of strings;
    raises SbnError(-1) naming a value that is not understood.  Needs no device."""
    buf = C.create_string_buffer(1024)
    _check(lib().sbn_settings_check(buf, 1024))
    return dict(kv.partition("=")[::2] for kv in buf.value.decode().split(" "))


class Prover:
    """Device context for one (table, degree_bits): buffers stay allocated across proofs."""

    def __init__(self, stark, config, degree_bits):
        self.stark, self.config, self.degree_bits = stark, config, degree_bits
        self._h = C.c_void_p()
        _check(lib().sbn_prover_create(C.byref(stark._d), C.byref(config._c), degree_bits, C.byref(self._h)))

    def load_trace(self, trace, public_inputs):
        trace = np.ascontiguousarray(trace, dtype=np.uint64)
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        _check(lib().sbn_prover_load_trace(self._h, _ptr(trace), _ptr(pi), len(pi)))

    def load_trace_device(self, device_ptr, public_inputs):
        pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
        _check(lib().sbn_prover_load_trace_device(self._h, C.c_void_p(device_ptr), _ptr(pi), len(pi)))

    def trace_device_ptr(self):
        return lib().sbn_prover_trace_device_ptr(self._h)

    def generate_trace(self, ios):
        """On-device G1ExpStark::generate_trace + generate_public_inputs; returns the public inputs."""
        ios = np.ascontiguousarray(ios, dtype=np.uint32)
        pi = np.zeros(self.stark.num_public_inputs, dtype=np.uint64)
        _check(lib().sbn_prover_generate_trace(self._h, _ptr(ios), ios.shape[0], _ptr(pi)))
        return pi

    def read_trace(self):
        trace = np.zeros((self.stark.num_columns, 1 << self.degree_bits), dtype=np.uint64)
        _check(lib().sbn_prover_read_trace(self._h, _ptr(trace)))
        return trace

    def prove(self):
        h = C.c_void_p()
        _check(lib().sbn_prover_prove(self._h, C.byref(h)))
        return _take_proof(h)

    def stage_times(self):
        buf = (C.c_float * 32)()
        k = lib().sbn_prover_stage_times(self._h, buf, 32)
        return {lib().sbn_prover_stage_name(i).decode(): float(buf[i]) for i in range(k)}

    def describe(self):
        """The SBN_* switches this prover was created under, resolved (csrc/settings.hpp), as a dict of strings."""
        buf = C.create_string_buffer(1024)
        _check(lib().sbn_prover_describe(self._h, buf, 1024))
        out = {}
        for kv in buf.value.decode().split(" "):
            k, _, v = kv.partition("=")
            out[k] = v
        return out

    def close(self):
        if self._h:
            lib().sbn_prover_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class BatchProver:
    """`inflight` prover contexts on one GPU: proves a batch of instance lists (witness generated on the device)."""

    def __init__(self, stark, config, degree_bits, inflight=3):
        self.stark, self.config, self.degree_bits = stark, config, degree_bits
        self._h = C.c_void_p()
        _check(lib().sbn_batch_prover_create(C.byref(stark._d), C.byref(config._c), degree_bits, inflight, C.byref(self._h)))

    def prove_ios(self, ios_units):
        """ios_units: (count, num_io, words_per_instance) uint32 -> list of Proof, in unit order."""
        ios = np.ascontiguousarray(ios_units, dtype=np.uint32)
        count, num_io, w = ios.shape
        out = (C.c_void_p * count)()
        _check(lib().sbn_batch_prover_prove_ios(self._h, _ptr(ios), num_io * w, num_io, count, out))
        return [_take_proof(C.c_void_p(h)) for h in out]

    def close(self):
        if self._h:
            lib().sbn_batch_prover_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def prove(stark, config, trace_poly_values, public_inputs, timing=None):
    """starky `prove(stark, &config, trace_poly_values, public_inputs, &mut timing)`."""
    trace = np.ascontiguousarray(trace_poly_values, dtype=np.uint64)
    pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
    n = trace.shape[1]
    if trace.shape[0] != stark.num_columns or n & (n - 1):
        raise SbnError(-1, "trace shape does not match the table")
    h = C.c_void_p()
    _check(lib().sbn_prove(C.byref(stark._d), C.byref(config._c), _ptr(trace), n.bit_length() - 1, _ptr(pi), len(pi), C.byref(h)))
    return _take_proof(h)


def verify_stark_proof(stark, proof, config):
    """starky `verify_stark_proof(stark, proof, &config)`; raises SbnError when rejected."""
    b = proof.to_bytes() if isinstance(proof, Proof) else bytes(proof)
    buf = (C.c_uint8 * len(b)).from_buffer_copy(b)
    _check(lib().sbn_verify(C.byref(stark._d), C.byref(config._c), buf, len(b)))


def commit_values(cols, rate_bits=1, cap_height=4, want_coeffs=False, want_lde=False):
    """PolynomialBatch::from_values on the device -> (cap, coeffs, lde)."""
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    ncols, n = cols.shape
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    coeffs = np.zeros_like(cols) if want_coeffs else None
    lde = np.zeros((ncols, n << rate_bits), dtype=np.uint64) if want_lde else None
    _check(lib().sbn_commit_values(_ptr(cols), ncols, n, rate_bits, cap_height, _ptr(cap), _ptr(coeffs), _ptr(lde)))
    return cap, coeffs, lde


def poseidon_permute_batch(states):
    s = np.ascontiguousarray(states, dtype=np.uint64).copy()
    _check(lib().sbn_poseidon_permute_batch(_ptr(s), s.shape[0]))
    return s


def field_mul_batch(a, b, mode=0):
    """a[i] * b[i] mod p with the device multiply (mode 0: the canonical operator* of every kernel; 1: the transforms' weak product),
    on arbitrary 64-bit representatives; canonical results."""
    a = np.ascontiguousarray(a, dtype=np.uint64)
    b = np.ascontiguousarray(b, dtype=np.uint64)
    out = np.zeros_like(a)
    _check(lib().sbn_field_mul_batch(_ptr(a), _ptr(b), _ptr(out), a.shape[0], mode))
    return out


def eval_constraints_host(stark, local_row, next_row, public_inputs, alphas, z_last, l_first, l_last):
    """The table's AIR constraints folded into the two Horner accumulators on one row pair (host, base field)."""
    lv = np.ascontiguousarray(local_row, dtype=np.uint64)
    nv = np.ascontiguousarray(next_row, dtype=np.uint64)
    pi = np.ascontiguousarray(public_inputs, dtype=np.uint64)
    al = np.array(alphas, dtype=np.uint64)
    acc = np.zeros(2, dtype=np.uint64)
    L = lib()
    L.sbn_eval_constraints_host.argtypes = [C.POINTER(_AirDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_uint64, C.c_uint64,
                                            C.c_uint64, C.c_void_p]
    _check(L.sbn_eval_constraints_host(C.byref(stark._d), _ptr(lv), _ptr(nv), _ptr(pi), len(pi), _ptr(al), z_last, l_first, l_last, _ptr(acc)))
    return [int(x) for x in acc]


def poseidon_permute_host(states, use_definition=False):
    """The transcript's host permutation (no device): sparse partial rounds, or the plain definition."""
    s = np.ascontiguousarray(states, dtype=np.uint64).copy()
    _check(lib().sbn_poseidon_permute_host(_ptr(s), s.shape[0], 1 if use_definition else 0))
    return s
