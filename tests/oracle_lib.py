"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so) -- TEST INFRASTRUCTURE ONLY.

Also holds small pure-Python BN254 helpers (Python ints) used to make inputs and to cross-check the
oracle's witness arithmetic independently.
"""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")

GL_P = 0xFFFFFFFF00000001
BN_P = 21888242871839275222246405745257275088696311157297823662689037894645226208583
BN_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

AIR_G1_OP = 1
AIR_G1_EXP = 2

_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        u64p = C.POINTER(C.c_uint64)
        L.orc_gf_mul.restype = C.c_uint64
        L.orc_gf_mul.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_gf_inv.restype = C.c_uint64
        L.orc_gf_inv.argtypes = [C.c_uint64]
        for f in ("orc_air_num_columns", "orc_air_num_public_inputs", "orc_air_num_permutation_zs"):
            getattr(L, f).restype = C.c_size_t
            getattr(L, f).argtypes = [C.c_int, C.c_size_t]
        L.orc_prove.restype = C.c_int
        L.orc_prove.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_uint, C.c_void_p, C.c_size_t,
                                C.POINTER(u64p), C.POINTER(C.c_size_t), C.POINTER(C.c_double)]
        L.orc_verify.restype = C.c_int
        L.orc_verify.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_char_p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_commit_values.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_uint, C.c_uint, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_hash_no_pad.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_challenger_probe.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_ifft.argtypes = [C.c_void_p, C.c_size_t]
        L.orc_coset_lde.argtypes = [C.c_void_p, C.c_size_t, C.c_uint, C.c_void_p]
        L.orc_g1exp_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_g1op_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_permuted_cols.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_eval_constraints.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                           C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p]
        if "OMP_NUM_THREADS" not in os.environ:
            # a GPU box shows every hardware thread of the host (256) but grants the container a share (cgroup cpu.max: 16 CPUs);
            # OpenMP's default of one thread per visible CPU makes the oracle 3-4 x SLOWER there (profiles/r3_oracle_scaling*.jsonl)
            L.orc_set_threads(_effective_cpus())
        _lib = L
    return _lib


def _effective_cpus():
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = max(1, min(n, -(-int(q) // int(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and period > 0:
                n = max(1, min(n, -(-q // period)))
        except (OSError, ValueError):
            pass
    return n


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


# ---------------------------------------------------------------- Poseidon / field
def poseidon_permute(state):
    a = np.array(state, dtype=np.uint64)
    lib().orc_poseidon_permute(ptr(a))
    return [int(x) for x in a]


def poseidon_permute_fast(state):
    """The oracle's sparse-partial-round form (used by its Merkle hashing); must equal poseidon_permute."""
    a = np.array(state, dtype=np.uint64)
    lib().orc_poseidon_permute_fast(ptr(a))
    return [int(x) for x in a]


def round_constants():
    a = np.zeros(360, dtype=np.uint64)
    lib().orc_poseidon_round_constants(ptr(a))
    return [int(x) for x in a]


def hash_no_pad(vals):
    a = np.array(vals, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().orc_hash_no_pad(ptr(a), len(a), ptr(out))
    return [int(x) for x in out]


def commit_values(cols, rate_bits=1, cap_height=4, want_coeffs=False, want_lde=False):
    cols = np.ascontiguousarray(cols, dtype=np.uint64)
    ncols, n = cols.shape
    cap = np.zeros((1 << cap_height, 4), dtype=np.uint64)
    coeffs = np.zeros_like(cols) if want_coeffs else None
    lde = np.zeros((ncols, n << rate_bits), dtype=np.uint64) if want_lde else None
    lib().orc_commit_values(ptr(cols), ncols, n, rate_bits, cap_height, ptr(cap),
                            ptr(coeffs) if want_coeffs else None, ptr(lde) if want_lde else None)
    return cap, coeffs, lde


# ---------------------------------------------------------------- BN254 in Python ints
def g1_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if (y1 + y2) % BN_P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, BN_P) % BN_P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, BN_P) % BN_P
    x3 = (lam * lam - x1 - x2) % BN_P
    y3 = (lam * (x1 - x3) - y1) % BN_P
    return (x3, y3)


def g1_mul(p, k):
    r = None
    while k:
        if k & 1:
            r = g1_add(r, p)
        p = g1_add(p, p)
        k >>= 1
    return r


def g1_random(rng):
    """Random affine G1 point: random x until x^3+3 is a square (p = 3 mod 4)."""
    while True:
        x = int(rng.integers(0, 1 << 62)) * (1 << 192) + int(rng.integers(0, 1 << 62)) * (1 << 128) \
            + int(rng.integers(0, 1 << 62)) * (1 << 64) + int(rng.integers(0, 1 << 62))
        x %= BN_P
        rhs = (x * x * x + 3) % BN_P
        y = pow(rhs, (BN_P + 1) // 4, BN_P)
        if y * y % BN_P == rhs:
            if int(rng.integers(0, 2)):
                y = BN_P - y
            return (x, y)


def u32_limbs(v, n=8):
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def g1exp_inputs(num_io, seed):
    """Mirror of src/curves/g1/exp.rs:794-809: random x, offset, exp_val = 8 uniform u32 limbs."""
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 40), dtype=np.uint32)
    native = []
    for k in range(num_io):
        x = g1_random(rng)
        off = g1_random(rng)
        exp = [int(v) for v in rng.integers(0, 1 << 32, size=8, dtype=np.uint64)]
        ios[k, 0:8] = u32_limbs(x[0]); ios[k, 8:16] = u32_limbs(x[1])
        ios[k, 16:24] = u32_limbs(off[0]); ios[k, 24:32] = u32_limbs(off[1])
        ios[k, 32:40] = exp
        native.append((x, off, sum(e << (32 * i) for i, e in enumerate(exp))))
    return ios, native


def g1op_inputs(rows, seed):
    rng = np.random.default_rng(seed)
    pts = np.zeros((rows, 32), dtype=np.uint32)
    native = []
    for r in range(rows):
        a = g1_random(rng); b = g1_random(rng)
        pts[r, 0:8] = u32_limbs(a[0]); pts[r, 8:16] = u32_limbs(a[1])
        pts[r, 16:24] = u32_limbs(b[0]); pts[r, 24:32] = u32_limbs(b[1])
        native.append((a, b))
    return pts, native


def g1exp_trace(ios):
    num_io = ios.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_G1_EXP, num_io)
    npi = L.orc_air_num_public_inputs(AIR_G1_EXP, num_io)
    trace = np.zeros((ncols, 512 * num_io), dtype=np.uint64)
    pi = np.zeros(npi, dtype=np.uint64)
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_g1exp_generate_trace(ptr(ios), num_io, ptr(trace), ptr(pi))
    return trace, pi


def g1op_trace(pts):
    rows = pts.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_G1_OP, 0)
    trace = np.zeros((ncols, rows), dtype=np.uint64)
    pts = np.ascontiguousarray(pts, dtype=np.uint32)
    L.orc_g1op_generate_trace(ptr(pts), rows, ptr(trace))
    return trace


def prove(kind, num_io, trace, pi):
    trace = np.ascontiguousarray(trace, dtype=np.uint64)
    pi = np.ascontiguousarray(pi, dtype=np.uint64)
    n = trace.shape[1]
    degree_bits = n.bit_length() - 1
    out = C.POINTER(C.c_uint64)()
    nw = C.c_size_t()
    secs = C.c_double()
    rc = lib().orc_prove(kind, num_io, ptr(trace), degree_bits, ptr(pi), len(pi), C.byref(out), C.byref(nw), C.byref(secs))
    if rc != 0:
        raise RuntimeError(f"orc_prove failed rc={rc}")
    words = np.ctypeslib.as_array(out, shape=(nw.value,)).copy()
    lib().orc_free(out)
    return words, secs.value


def last_stage_seconds():
    """{stage name: wall seconds} of the last prove() (bench.py cpu_baseline)."""
    L = lib()
    L.orc_last_stage_seconds.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double)]
    out = {}
    n = L.orc_last_stage_seconds(-1, None, None)
    for k in range(n):
        name, secs = C.c_char_p(), C.c_double()
        L.orc_last_stage_seconds(k, C.byref(name), C.byref(secs))
        out[name.value.decode()] = secs.value
    return out


def set_final_poly_times_x(on):
    """Which plonky2 FRI the oracle speaks: True (default) = the 0.1.x 'multiply the final polynomial by X' step."""
    lib().orc_set_final_poly_times_x(1 if on else 0)


def verify(kind, num_io, words):
    words = np.ascontiguousarray(words, dtype=np.uint64)
    why = C.c_char_p()
    rc = lib().orc_verify(kind, num_io, ptr(words), len(words), C.byref(why))
    return rc, (why.value or b"").decode()


# ---------------------------------------------------------------- extra probes
def modular_witness(pol_input, zero_mode):
    a = np.array(pol_input, dtype=np.int64)
    out = np.zeros(112, dtype=np.int64)
    lib().orc_modular_witness(ptr(a), int(zero_mode), ptr(out))
    o = [int(x) for x in out]
    return {"output": o[0:16], "out_aux_red": o[16:32], "quot_abs": o[32:49], "aux_lo": o[49:80], "aux_hi": o[80:111], "sign": o[111]}


def fq_mul(a, b):
    A = np.array(u32_limbs(a), dtype=np.uint32); B = np.array(u32_limbs(b), dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    lib().orc_fq_mul(ptr(A), ptr(B), ptr(out))
    return sum(int(v) << (32 * i) for i, v in enumerate(out))


def fq_inv(a):
    A = np.array(u32_limbs(a), dtype=np.uint32)
    out = np.zeros(8, dtype=np.uint32)
    lib().orc_fq_inv(ptr(A), ptr(out))
    return sum(int(v) << (32 * i) for i, v in enumerate(out))


def flags_table(limbs):
    a = np.array(limbs, dtype=np.uint32)
    out = np.zeros((512, 14), dtype=np.uint64)
    lib().orc_flags_table(ptr(a), ptr(out))
    return out


def permuted_cols(inputs, table):
    i = np.array(inputs, dtype=np.uint64); t = np.array(table, dtype=np.uint64)
    so = np.zeros_like(i); pt = np.zeros_like(i)
    lib().orc_permuted_cols(ptr(i), ptr(t), len(i), ptr(so), ptr(pt))
    return [int(x) for x in so], [int(x) for x in pt]


def eval_constraints(kind, num_io, lv, nv, pi, alphas, z_last, l_first, l_last):
    lv = np.ascontiguousarray(lv, dtype=np.uint64); nv = np.ascontiguousarray(nv, dtype=np.uint64)
    pi = np.ascontiguousarray(pi, dtype=np.uint64); al = np.array(alphas, dtype=np.uint64)
    acc = np.zeros(len(al), dtype=np.uint64)
    rc = lib().orc_eval_constraints(kind, num_io, ptr(lv), ptr(nv), ptr(pi), ptr(al), len(al), z_last, l_first, l_last, ptr(acc))
    assert rc == 0
    return [int(x) for x in acc]


def challenger_probe(inputs, m):
    a = np.array(inputs, dtype=np.uint64)
    out = np.zeros(m, dtype=np.uint64)
    lib().orc_challenger_probe(ptr(a), len(a), ptr(out), m)
    return [int(x) for x in out]


def gf_pow(b, e):
    return pow(b, e, GL_P)


# ---------------------------------------------------------------- G2 (BN254 twist over Fq2 = Fq[i]/(i^2+1)) in Python ints
AIR_G2_EXP = 3


def fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % BN_P, (a[0] * b[1] + a[1] * b[0]) % BN_P)


def fq2_add(a, b):
    return ((a[0] + b[0]) % BN_P, (a[1] + b[1]) % BN_P)


def fq2_sub(a, b):
    return ((a[0] - b[0]) % BN_P, (a[1] - b[1]) % BN_P)


def fq2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, BN_P)
    return (a[0] * n % BN_P, (-a[1]) * n % BN_P)


def fq2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = fq2_mul(r, a)
        a = fq2_mul(a, a)
        e >>= 1
    return r


def fq2_sqrt(a):
    """Square root in Fq2 for p = 3 mod 4 (Adj-Rodriguez-Henriquez); None if `a` is not a square."""
    if a == (0, 0):
        return (0, 0)
    a1 = fq2_pow(a, (BN_P - 3) // 4)
    alpha = fq2_mul(fq2_mul(a1, a1), a)
    a0 = fq2_mul(fq2_pow(alpha, BN_P), alpha)
    if a0 == (BN_P - 1, 0):
        return None
    x0 = fq2_mul(a1, a)
    if alpha == (BN_P - 1, 0):
        return fq2_mul((0, 1), x0)
    b = fq2_pow(fq2_add((1, 0), alpha), (BN_P - 1) // 2)
    return fq2_mul(b, x0)


G2_B = fq2_mul((3, 0), fq2_inv((9, 1)))     # twist y^2 = x^3 + 3/(9+i)


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    (x1, y1), (x2, y2) = p, q
    if x1 == x2:
        if fq2_add(y1, y2) == (0, 0):
            return None
        lam = fq2_mul(fq2_mul((3, 0), fq2_mul(x1, x1)), fq2_inv(fq2_add(y1, y1)))
    else:
        lam = fq2_mul(fq2_sub(y2, y1), fq2_inv(fq2_sub(x2, x1)))
    x3 = fq2_sub(fq2_sub(fq2_mul(lam, lam), x1), x2)
    y3 = fq2_sub(fq2_mul(lam, fq2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(p, k):
    r = None
    while k:
        if k & 1:
            r = g2_add(r, p)
        p = g2_add(p, p)
        k >>= 1
    return r


def g2_random(rng):
    """Random affine point on the twist curve (not cofactor-cleared: the table only checks the group law)."""
    while True:
        x = (int.from_bytes(rng.bytes(32), "little") % BN_P, int.from_bytes(rng.bytes(32), "little") % BN_P)
        rhs = fq2_add(fq2_mul(fq2_mul(x, x), x), G2_B)
        y = fq2_sqrt(rhs)
        if y is not None and fq2_mul(y, y) == rhs:
            if int(rng.integers(0, 2)):
                y = ((-y[0]) % BN_P, (-y[1]) % BN_P)
            return (x, y)


def g2exp_inputs(num_io, seed):
    """Mirror of src/curves/g2/exp.rs:845-860 with seeded randomness."""
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 72), dtype=np.uint32)
    native = []
    for k in range(num_io):
        x = g2_random(rng)
        off = g2_random(rng)
        exp = [int(v) for v in rng.integers(0, 1 << 32, size=8, dtype=np.uint64)]
        vals = [x[0][0], x[0][1], x[1][0], x[1][1], off[0][0], off[0][1], off[1][0], off[1][1]]
        for j, v in enumerate(vals):
            ios[k, 8 * j:8 * j + 8] = u32_limbs(v)
        ios[k, 64:72] = exp
        native.append((x, off, sum(e << (32 * i) for i, e in enumerate(exp))))
    return ios, native


def g2exp_trace(ios):
    num_io = ios.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_G2_EXP, num_io)
    npi = L.orc_air_num_public_inputs(AIR_G2_EXP, num_io)
    trace = np.zeros((ncols, 512 * num_io), dtype=np.uint64)
    pi = np.zeros(npi, dtype=np.uint64)
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_g2exp_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_g2exp_generate_trace(ptr(ios), num_io, ptr(trace), ptr(pi))
    return trace, pi


# ---------------------------------------------------------------- Fq12 = Fq2[w]/(w^6 - (9+i)), flat basis r_0..r_5, s_0..s_5
AIR_FQ12_EXP = 4


def fq12_mul(a, b):
    """a, b: 12 ints [r_0..r_5, s_0..s_5] (coefficient of w^k is r_k + s_k i). Tower arithmetic over Fq2."""
    A = [(a[k], a[k + 6]) for k in range(6)]
    B = [(b[k], b[k + 6]) for k in range(6)]
    prod = [(0, 0)] * 11
    for i in range(6):
        for j in range(6):
            prod[i + j] = fq2_add(prod[i + j], fq2_mul(A[i], B[j]))
    xi = (9, 1)
    out = [fq2_add(prod[k], fq2_mul(xi, prod[k + 6])) if k < 5 else prod[k] for k in range(6)]
    return [c[0] for c in out] + [c[1] for c in out]


def fq12_pow(a, e):
    r = [1] + [0] * 11
    while e:
        if e & 1:
            r = fq12_mul(r, a)
        a = fq12_mul(a, a)
        e >>= 1
    return r


def fq12exp_inputs(num_io, seed):
    """Mirror of src/fields/fq12/exp.rs:647-660 with seeded randomness: random Fq12 x, offset; exponent < r."""
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 200), dtype=np.uint32)
    native = []
    for k in range(num_io):
        x = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        off = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        e = int.from_bytes(rng.bytes(32), "little") % BN_R
        for c in range(12):
            ios[k, 8 * c:8 * c + 8] = u32_limbs(x[c])
            ios[k, 96 + 8 * c:96 + 8 * c + 8] = u32_limbs(off[c])
        ios[k, 192:200] = u32_limbs(e)
        native.append((x, off, e))
    return ios, native


def fq12exp_trace(ios):
    num_io = ios.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_FQ12_EXP, num_io)
    npi = L.orc_air_num_public_inputs(AIR_FQ12_EXP, num_io)
    trace = np.zeros((ncols, 512 * num_io), dtype=np.uint64)
    pi = np.zeros(npi, dtype=np.uint64)
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_fq12exp_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_fq12exp_generate_trace(ptr(ios), num_io, ptr(trace), ptr(pi))
    return trace, pi


# ---- FqExpStark (src/fields/fq/exp.rs) ---------------------------------------------------------------------------------
AIR_FQ_EXP = 5


def fqexp_inputs(num_io, seed):
    """Mirror of src/fields/fq/exp.rs:594-607 with seeded randomness: random Fq x, offset; 256-bit exponent limbs."""
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 24), dtype=np.uint32)
    native = []
    for k in range(num_io):
        x = int.from_bytes(rng.bytes(32), "little") % BN_P
        off = int.from_bytes(rng.bytes(32), "little") % BN_P
        e = int.from_bytes(rng.bytes(32), "little")
        ios[k, 0:8] = u32_limbs(x)
        ios[k, 8:16] = u32_limbs(off)
        ios[k, 16:24] = u32_limbs(e)
        native.append((x, off, e))
    return ios, native


def fqexp_trace(ios):
    num_io = ios.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_FQ_EXP, num_io)
    npi = L.orc_air_num_public_inputs(AIR_FQ_EXP, num_io)
    trace = np.zeros((ncols, 512 * num_io), dtype=np.uint64)
    pi = np.zeros(npi, dtype=np.uint64)
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_fqexp_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_fqexp_generate_trace(ptr(ios), num_io, ptr(trace), ptr(pi))
    return trace, pi


# ---- Fq12ExpU64Stark (src/fields/fq12_u64/exp_u64.rs) ------------------------------------------------------------------
AIR_FQ12_EXP_U64 = 6
GL_P = 0xFFFFFFFF00000001


def fq12expu64_inputs(num_io, seed):
    """Mirror of src/fields/fq12_u64/exp_u64.rs:607-621 with seeded randomness: random Fq12 x, offset; the exponent is a
    canonical Goldilocks element (F::sample(..).to_canonical_u64())."""
    rng = np.random.default_rng(seed)
    ios = np.zeros((num_io, 194), dtype=np.uint32)
    native = []
    for k in range(num_io):
        x = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        off = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        e = int.from_bytes(rng.bytes(8), "little") % GL_P
        for c in range(12):
            ios[k, 8 * c:8 * c + 8] = u32_limbs(x[c])
            ios[k, 96 + 8 * c:96 + 8 * c + 8] = u32_limbs(off[c])
        ios[k, 192:194] = [e & 0xFFFFFFFF, e >> 32]
        native.append((x, off, e))
    return ios, native


def fq12expu64_trace(ios):
    num_io = ios.shape[0]
    L = lib()
    ncols = L.orc_air_num_columns(AIR_FQ12_EXP_U64, num_io)
    npi = L.orc_air_num_public_inputs(AIR_FQ12_EXP_U64, num_io)
    trace = np.zeros((ncols, 128 * num_io), dtype=np.uint64)
    pi = np.zeros(npi, dtype=np.uint64)
    ios = np.ascontiguousarray(ios, dtype=np.uint32)
    L.orc_fq12expu64_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    L.orc_fq12expu64_generate_trace(ptr(ios), num_io, ptr(trace), ptr(pi))
    return trace, pi


# ---- the reference's single-operation test tables: ModularStark (src/modular/modular.rs:361-537), Fq12Stark (src/fields/fq12/mul.rs:355-517)
AIR_MODULAR = 7
AIR_FQ12_MUL = 8
AIR_LOOKUP = 9      # oracle only: MyStark, the lookup unit-test table (src/utils/lookup.rs:136-213), 8 rows in the reference's test


AIR_FLAGS = 10      # FlagStark, the flags unit-test table (src/utils/flags.rs:379-547): no permutation pairs


AIR_FLAGS_U64 = 11  # the FlagStark of src/fields/fq12_u64/flags_u64.rs (u64 exponents, 128 rows each)


def flags_u64_inputs(num_io, seed):
    return np.random.default_rng(seed).integers(0, GL_P, size=num_io, dtype=np.uint64), None


def flags_u64_trace(exps):
    exps = np.ascontiguousarray(exps, dtype=np.uint64)
    k = exps.shape[0]
    trace = np.zeros((7 + 4 * k, 128 * k), dtype=np.uint64)
    lib().orc_flags_u64_generate_trace(ptr(exps), k, ptr(trace))
    return trace


def lookup_inputs(rows, seed):
    """A MyStark instance of `rows` rows: table 0 .. rows-1 shuffled, inputs drawn from a quarter of the table (repeats)."""
    rng = np.random.default_rng(seed)
    table = rng.permutation(rows).astype(np.uint64)
    inputs = rng.choice(table[: max(rows // 4, 1)], size=rows).astype(np.uint64)
    return inputs, table


def flags_inputs(num_io, seed):
    """num_io x 8 u32 limbs (flags.rs:551-554 draws them at random)."""
    return np.random.default_rng(seed).integers(0, 1 << 32, size=(num_io, 8), dtype=np.uint64).astype(np.uint32), None


def flags_trace(limbs):
    limbs = np.ascontiguousarray(limbs, dtype=np.uint32)
    k = limbs.shape[0]
    trace = np.zeros((17 + 4 * k, 512 * k), dtype=np.uint64)
    lib().orc_flags_generate_trace(ptr(limbs), k, ptr(trace))
    return trace


def lookup_fixed_inputs():
    """The crate's only fixed test input (src/utils/lookup.rs:154-161): inputs and table of MyStark::generate_trace."""
    return np.array([6, 3, 1, 1, 0, 0, 0, 0], dtype=np.uint64), np.arange(8, dtype=np.uint64)


def lookup_trace(inputs, table):
    """MyStark::generate_trace (lookup.rs:151-166): columns inputs, table, permuted inputs, permuted table."""
    inputs = np.ascontiguousarray(inputs, dtype=np.uint64)
    table = np.ascontiguousarray(table, dtype=np.uint64)
    trace = np.zeros((4, len(inputs)), dtype=np.uint64)
    lib().orc_lookup_generate_trace(ptr(inputs), ptr(table), len(inputs), ptr(trace))
    return trace


def modular_inputs(rows, seed):
    """Mirror of modular.rs:389-391 with seeded randomness: two random Fq per row."""
    rng = np.random.default_rng(seed)
    ops = np.zeros((rows, 16), dtype=np.uint32)
    native = []
    for r in range(rows):
        a = int.from_bytes(rng.bytes(32), "little") % BN_P
        b = int.from_bytes(rng.bytes(32), "little") % BN_P
        ops[r, 0:8] = u32_limbs(a); ops[r, 8:16] = u32_limbs(b)
        native.append((a, b))
    return ops, native


def modular_trace(ops):
    rows = ops.shape[0]
    L = lib()
    trace = np.zeros((L.orc_air_num_columns(AIR_MODULAR, 0), rows), dtype=np.uint64)
    ops = np.ascontiguousarray(ops, dtype=np.uint32)
    L.orc_modular_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_modular_generate_trace(ptr(ops), rows, ptr(trace))
    return trace


def fq12mul_inputs(rows, seed):
    """Mirror of fq12/mul.rs:378-380 with seeded randomness: two random Fq12 per row (12 coefficients each, flat basis)."""
    rng = np.random.default_rng(seed)
    ops = np.zeros((rows, 192), dtype=np.uint32)
    native = []
    for r in range(rows):
        x = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        y = [int.from_bytes(rng.bytes(32), "little") % BN_P for _ in range(12)]
        for c in range(12):
            ops[r, 8 * c:8 * c + 8] = u32_limbs(x[c])
            ops[r, 96 + 8 * c:96 + 8 * c + 8] = u32_limbs(y[c])
        native.append((x, y))
    return ops, native


def fq12mul_trace(ops):
    rows = ops.shape[0]
    L = lib()
    trace = np.zeros((L.orc_air_num_columns(AIR_FQ12_MUL, 0), rows), dtype=np.uint64)
    ops = np.ascontiguousarray(ops, dtype=np.uint32)
    L.orc_fq12mul_generate_trace.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.orc_fq12mul_generate_trace(ptr(ops), rows, ptr(trace))
    return trace
