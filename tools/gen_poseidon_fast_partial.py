#!/usr/bin/env python3
"""Regenerates starky_bn254_amd/csrc/poseidon_fast_consts.inc: the sparse form of the 22 partial rounds.

A partial round is x <- M * S(x + c_r), where S raises only x[0] to the 7th power.  Two rewrites (Poseidon paper,
appendix B; plonky2 ships equivalent tables) make it cheap without changing the permutation:

 * constants: everything in c_r except c_r[0] commutes with S and is pushed through M into c_{r+1}; the whole of c_0
   is added once up front, and what is left after the last partial round lands on the next full round's constants.
   Afterwards round r adds one scalar k_r to x[0].
 * matrix: with M = [[m00, v^T], [w, Mh]] and D_j = diag(1, Mh^j), M = [[m00, v^T Mh^-1], [w, I]] * D_1 and D_j
   commutes with S, so D is carried backwards through the rounds: round r multiplies by the sparse
   [[m00, vhat_r^T], [what_r, I]], vhat_r = v^T Mh^-(R-r), what_r = Mh^(R-r-1) w, and D_R is applied once before
   round 0 (state[0] untouched, an 11x11 product on the rest).

The script checks the rewritten permutation against the plain one on random states before writing the table.
"""
import os
import random
from _chacha8 import ChaChaRng, P

T, RF_HALF, RP = 12, 4, 22
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0] * 11

g = ChaChaRng(0, 8)
RC = [g.gen_range(P) for _ in range(360)]
# plonky2 mds_row_shf: out[r] = sum_i state[(i + r) % 12] * CIRC[i] + state[r] * DIAG[r]
M = [[0] * T for _ in range(T)]
for r in range(T):
    for i in range(T):
        M[r][(i + r) % T] = (M[r][(i + r) % T] + CIRC[i]) % P
    M[r][r] = (M[r][r] + DIAG[r]) % P


def matvec(A, x):
    return [sum(a * b for a, b in zip(row, x)) % P for row in A]


def matmul(A, B):
    n, m, k = len(A), len(B[0]), len(B)
    return [[sum(A[i][l] * B[l][j] for l in range(k)) % P for j in range(m)] for i in range(n)]


def matinv(A):
    n = len(A)
    a = [row[:] + [1 if i == j else 0 for j in range(n)] for i, row in enumerate(A)]
    for c in range(n):
        piv = next(r for r in range(c, n) if a[r][c])
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], P - 2, P)
        a[c] = [x * inv % P for x in a[c]]
        for r in range(n):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]
    return [row[n:] for row in a]


def permute_plain(s):
    s = s[:]
    for r in range(30):
        s = [(x + RC[12 * r + i]) % P for i, x in enumerate(s)]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = matvec(M, s)
    return s


# --- constants ------------------------------------------------------------------------------------------------------
c = [RC[12 * (RF_HALF + r):12 * (RF_HALF + r) + 12] for r in range(RP)]
first_const = c[0][:]
k = [0] * RP
carry = [0] * T                       # (0, chat) of the previous round pushed through M
for r in range(1, RP):
    eff = [(a + b) % P for a, b in zip(c[r], carry)]
    k[r] = eff[0]
    carry = matvec(M, [0] + eff[1:])
adj_next_full = carry                 # added to the constants of full round RF_HALF + RP

# --- matrices -------------------------------------------------------------------------------------------------------
m00 = M[0][0]
v = M[0][1:]
w = [M[i][0] for i in range(1, T)]
Mh = [row[1:] for row in M[1:]]
Mh_inv = matinv(Mh)


def matpow(A, e):
    R = [[1 if i == j else 0 for j in range(len(A))] for i in range(len(A))]
    for _ in range(e):
        R = matmul(R, A)
    return R


vhat, what = [], []
for r in range(RP):
    Ai = matpow(Mh_inv, RP - r)                      # Mh^-(R-r)
    vhat.append([sum(v[l] * Ai[l][j] for l in range(T - 1)) % P for j in range(T - 1)])
    what.append(matvec(matpow(Mh, RP - r - 1), w))
init = matpow(Mh, RP)


def permute_fast(s):
    s = s[:]
    for r in range(4):
        s = [pow((x + RC[12 * r + i]) % P, 7, P) for i, x in enumerate(s)]
        s = matvec(M, s)
    s = [(x + y) % P for x, y in zip(s, first_const)]
    s = [s[0]] + matvec(init, s[1:])
    for r in range(RP):
        s0 = pow((s[0] + k[r]) % P, 7, P)
        n0 = (m00 * s0 + sum(a * b for a, b in zip(vhat[r], s[1:]))) % P
        s = [n0] + [(x + wi * s0) % P for x, wi in zip(s[1:], what[r])]
    for r in range(26, 30):
        rc = RC[12 * r:12 * r + 12]
        if r == 26:
            rc = [(a + b) % P for a, b in zip(rc, adj_next_full)]
        s = [pow((x + rc[i]) % P, 7, P) for i, x in enumerate(s)]
        s = matvec(M, s)
    return s


rnd = random.Random(1)
for trial in range(20):
    st = [rnd.randrange(P) for _ in range(T)] if trial else [0] * T
    assert permute_plain(st) == permute_fast(st), "rewritten partial rounds disagree with the plain permutation"
assert permute_plain([0] * 12)[0] == 0x3c18a9786cb0b359      # plonky2 test vector (all-zero input)

csrc = os.environ.get("SBN_GEN_OUT") or os.path.join(os.path.dirname(__file__), "..", "starky_bn254_amd", "csrc")
# (1) "effective" round constants: same shape as the plain table, partial rounds carry one scalar (device + host)
eff = RC[:]
eff[12 * 4:12 * 5] = first_const
for r in range(1, RP):
    eff[12 * (4 + r):12 * (5 + r)] = [k[r]] + [0] * 11
eff[12 * 26:12 * 27] = [(a + b) % P for a, b in zip(RC[12 * 26:12 * 27], adj_next_full)]
out = os.path.join(csrc, "poseidon_eff_consts.inc")
with open(out, "w") as f:
    f.write("// Generated by tools/gen_poseidon_fast_partial.py -- do not edit.\n")
    f.write("// Round constants with the partial rounds' vector constants pushed through the MDS layer: rounds 5..25 add\n")
    f.write("// one scalar to state[0] (row = k, 0, ..., 0), round 4 keeps a full row, round 26 absorbs the remainder.\n")
    for r in range(30):
        f.write("  " + ", ".join("0x%016xULL" % x for x in eff[12 * r:12 * r + 12]) + ",\n")
print("wrote", os.path.normpath(out))

# (1b) the constants the device adds INSIDE the full MDS layers (poseidon.cuh mdsc): row j of the table belongs to the layer
# of full round j (j < 4: rounds 0..3, then rounds 26..29) and holds the NEXT round's constants split into zero-extended
# 32-bit halves, [row][lo | hi][12]; the layer of round 3 leaves element 0 to the partial-round loop, the last layer adds 0.
rows = [eff[12 * 1:12 * 2], eff[12 * 2:12 * 3], eff[12 * 3:12 * 4], [0] + eff[12 * 4 + 1:12 * 5],
        eff[12 * 27:12 * 28], eff[12 * 28:12 * 29], eff[12 * 29:12 * 30], [0] * 12]
out = os.path.join(csrc, "poseidon_kx_consts.inc")
with open(out, "w") as f:
    f.write("// Generated by tools/gen_poseidon_fast_partial.py -- do not edit.\n")
    f.write("// [8 layers][lo halves | hi halves][12] of the next round's constants, zero-extended to 64 bits.\n")
    for row in rows:
        f.write("  " + ", ".join("0x%08xULL" % (x & 0xFFFFFFFF) for x in row) + ",\n")
        f.write("  " + ", ".join("0x%08xULL" % (x >> 32) for x in row) + ",\n")
print("wrote", os.path.normpath(out))

# (2) sparse partial-round matrices (host transcript)
out = os.path.join(csrc, "poseidon_fast_consts.inc")
with open(out, "w") as f:
    f.write("// Generated by tools/gen_poseidon_fast_partial.py -- do not edit.\n")

    def arr(name, vals, per=4):
        f.write("static const u64 %s[%d] = {\n" % (name, len(vals)))
        for i in range(0, len(vals), per):
            f.write("  " + ", ".join("0x%016xULL" % x for x in vals[i:i + per]) + ",\n")
        f.write("};\n")
    arr("PFAST_INIT", [x for row in init for x in row])   # [11][11], new[1+i] = sum_j INIT[i][j] * old[1+j]
    arr("PFAST_VHAT", [x for row in vhat for x in row])   # [22][11]
    arr("PFAST_WHAT", [x for row in what for x in row])   # [22][11]
    f.write("static constexpr u64 PFAST_M00 = %d;\n" % m00)
print("wrote", os.path.normpath(out))
