#!/usr/bin/env python3
"""Generates the gfx950 instruction streams of the Goldilocks x^7 S-box and of the MDS output fold:

  poseidon_sbox3_asm.inc   three S-boxes interleaved instruction by instruction (full rounds)
  poseidon_sbox1_asm.inc   one S-box (partial rounds)
  poseidon_fold3_asm.inc   three MDS outputs  lo + hi*2^32 -> one weak u64 each
  poseidon_fold1_asm.inc   one MDS output
  gl_mul_asm.inc           one canonical multiply (kept for A/B; the kernels use gl.cuh's primitive form)

Why by hand: the compiler's Goldilocks multiply costs ~29 instructions (the low and high halves of the 128-bit
product are computed separately, every carry becomes a 64-bit compare + select, and each zero-extended addend of
v_mad_u64_u32 costs two moves).  Here a multiply is 13 instructions (round 3; 16 in rounds 1-2):

  P0 = a0*b0                       v_mad_u64_u32
  P1 = a0*b1 + hi(P0)              mov + v_mad_u64_u32      (the addend pair is {value, 0}: its odd register stays 0)
  P2 = a1*b0 + P1                  v_mad_u64_u32 with the WHOLE P1 as addend; its carry-out cM has weight 2^96 = -1
  P3 = a1*b1 + hi(P2)              mov + v_mad_u64_u32      ->  a*b = lo(P0) + 2^32 lo(P2) + 2^64 P3 - cM
  X  = (lo(P2):lo(P0)) - hi(P3) - cM    v_subb (cM is its borrow-IN: free), v_subbrev -> borrow b     [2^96 = -1]
  T  = lo(P3) - b                  v_subbrev -> borrow b2 (only if lo(P3) = 0): (lo(P3) - b) eps = T eps + b2  [2^32 eps = -1]
  R  = T*(2^32-1) + X              v_mad_u64_u32 with carry-out c                                      [2^64 = 2^32-1]
  r  = R + (c ? 2^32-1 : 0) + b2   cndmask, addc, addc      (b2 implies c, and the sum cannot wrap: see DESIGN section 3)

Rounds 1-2 formed hi(P1) + hi(P2) as a 64-bit column sum (add, addc), repaid the borrow of X with three instructions
(cndmask, addc, addc) before the reducing multiply-add and the carry with two after it.  Letting the cross-term carry
ride into the subtraction as a borrow-in, and the borrow into the multiplier of the reducing multiply-add, takes three
instructions off every multiply: 472 multiplies per permutation.

Registers: 64-bit operands must be even-aligned pairs whose halves are named separately, which inline-asm operands
cannot express, so every stream works in a fixed window of physical VGPRs (clobbered), carries live in two per-stream
SGPR pairs (asm operands).  gfx950 needs two wait states between a VALU writing an SGPR and a VALU reading it; the three
interleaved streams provide them (a single stream pads with s_nop).
"""
import os

CSRC = os.environ.get("SBN_GEN_OUT") or os.path.join(os.path.dirname(__file__), "..", "starky_bn254_amd", "csrc")
# MEASUREMENT variants of the two zero-extension moves of a multiply (tools/ab_sponge_moves.sh; never used for the library):
#   nomov = the moves left out (wrong results: what would the stream cost without them?), lshr = v_lshrrev_b64 {hi, 0} instead
VARIANT = os.environ.get("SBN_GEN_VARIANT", "")


class Stream:
    """Instruction list of one dependent chain; each entry = (text, carries read, carries written)."""

    def __init__(self, base, ca, cb):
        self.b, self.ca, self.cb, self.ins = base, ca, cb, []

    def pair(self, k):
        return f"v[{self.b + 2 * k}:{self.b + 2 * k + 1}]"

    def lo(self, k):
        return f"v{self.b + 2 * k}"

    def hi(self, k):
        return f"v{self.b + 2 * k + 1}"

    def emit(self, text, rd=(), wr=()):
        self.ins.append((text, tuple(rd), tuple(wr)))


ZY, P0, P1, P3, S2, S4 = range(6)   # pair indices inside a stream's window (12 VGPRs)
WIN = 12


def mul(s, a, b, dst):
    """a, b = (lo, hi) register names; dst = (lo, hi) register names of the destination (may be asm operands, may be a).
    Register reuse inside the window: P2 accumulates in place over P1, X and R in place over P0 (its high half is dead
    after the first move), T and the carry mask share the low half of the one zero-extended pair ZY."""
    ca, cb = s.ca, s.cb
    def zext(src_pair, src_hi):
        if VARIANT == "nomov":
            return
        if VARIANT == "lshr":
            s.emit(f"v_lshrrev_b64 {s.pair(ZY)}, 32, {src_pair}")
        else:
            s.emit(f"v_mov_b32 {s.lo(ZY)}, {src_hi}")
    s.emit(f"v_mad_u64_u32 {s.pair(P0)}, {ca}, {a[0]}, {b[0]}, 0", wr=[ca])
    zext(s.pair(P0), s.hi(P0))
    s.emit(f"v_mad_u64_u32 {s.pair(P1)}, {ca}, {a[0]}, {b[1]}, {s.pair(ZY)}", wr=[ca])
    s.emit(f"v_mad_u64_u32 {s.pair(P1)}, {cb}, {a[1]}, {b[0]}, {s.pair(P1)}", wr=[cb])            # P2 (in place); cb = cM
    zext(s.pair(P1), s.hi(P1))
    s.emit(f"v_mad_u64_u32 {s.pair(P3)}, {ca}, {a[1]}, {b[1]}, {s.pair(ZY)}", wr=[ca])
    s.emit(f"v_subb_co_u32_e64 {s.lo(P0)}, {cb}, {s.lo(P0)}, {s.hi(P3)}, {cb}", rd=[cb], wr=[cb])   # X.lo
    s.emit(f"v_subbrev_co_u32_e64 {s.hi(P0)}, {cb}, 0, {s.lo(P1)}, {cb}", rd=[cb], wr=[cb])        # X.hi; cb = b
    s.emit(f"v_subbrev_co_u32_e64 {s.lo(ZY)}, {cb}, 0, {s.lo(P3)}, {cb}", rd=[cb], wr=[cb])        # T; cb = b2
    s.emit(f"v_mad_u64_u32 {s.pair(P0)}, {ca}, {s.lo(ZY)}, -1, {s.pair(P0)}", wr=[ca])             # R (in place); ca = c
    s.emit(f"v_cndmask_b32_e64 {s.lo(ZY)}, 0, -1, {ca}", rd=[ca])
    s.emit(f"v_addc_co_u32_e64 {dst[0]}, {ca}, {s.lo(P0)}, {s.lo(ZY)}, {cb}", rd=[cb], wr=[ca])
    s.emit(f"v_addc_co_u32_e64 {dst[1]}, {ca}, {s.hi(P0)}, 0, {ca}", rd=[ca], wr=[ca])


def sbox(s, x0, x1, out):
    s.emit(f"v_mov_b32 {s.hi(ZY)}, 0")
    x = (x0, x1)
    x2 = (s.lo(S2), s.hi(S2))
    x4 = (s.lo(S4), s.hi(S4))
    mul(s, x, x, x2)
    mul(s, x2, x2, x4)
    mul(s, x2, x, x2)          # x^3 over x^2: the destination is written after the last read of the operands
    mul(s, x4, x2, out)


def fold(s, al, ah0, ah1, out):
    """out = al + ah*2^32 (al, ah < 2^59) as a weak u64:  W = ah1*(2^32-1) + al (no carry: < 2^60), W.hi += ah0 with
    carry k (weight 2^64 = 2^32-1), out = k*(2^32-1) + W (cannot wrap: after a carry W < 2^60).  al is a 64-bit operand
    (the accumulator pair itself is the addend: no move), ah0 / ah1 its partner's halves."""
    ca = s.ca
    W, U = 0, 1
    s.emit(f"v_mad_u64_u32 {s.pair(W)}, {ca}, {ah1}, -1, {al}", wr=[ca])
    s.emit(f"v_add_co_u32_e64 {s.hi(W)}, {ca}, {s.hi(W)}, {ah0}", wr=[ca])
    s.emit(f"v_cndmask_b32_e64 {s.lo(U)}, 0, 1, {ca}", rd=[ca])
    s.emit(f"v_mad_u64_u32 {out}, {ca}, {s.lo(U)}, -1, {s.pair(W)}", wr=[ca])


def schedule(streams):
    """Round-robin merge; pads with s_nop so that a carry reader sits >= 3 slots after the writer of that SGPR pair."""
    out, last_write = [], {}
    idx = [0] * len(streams)
    while any(idx[k] < len(s.ins) for k, s in enumerate(streams)):
        for k, s in enumerate(streams):
            if idx[k] >= len(s.ins):
                continue
            text, rd, wr = s.ins[idx[k]]
            need = 0
            for c in rd:
                if c in last_write:
                    gap = len(out) - last_write[c] - 1      # instructions between writer and reader
                    need = max(need, 2 - gap)
            if need > 0:
                out.append(f"s_nop {need - 1}")
            out.append(text)
            for c in wr:
                last_write[c] = len(out) - 1
            idx[k] += 1
    return out


def write(name, lines, header):
    path = os.path.join(CSRC, name)
    with open(path, "w") as f:
        f.write("// Generated by tools/gen_poseidon_sbox_asm.py -- do not edit.\n")
        for h in header:
            f.write("// " + h + "\n")
        for l in lines:
            f.write('"%s\\n\\t"\n' % l)
    print("wrote", os.path.normpath(path), len(lines), "instructions")


SB_BASE = [92, 104, 116]     # three windows of 12 VGPRs: v92..v127
# sbox3: operands %0..%5 = output halves (lo, hi) x3, %6..%11 = two carry SGPR pairs per stream (scratch outputs),
# %12..%17 = lo, hi of each input
st = [Stream(SB_BASE[k], f"%{6 + 2 * k}", f"%{7 + 2 * k}") for k in range(3)]
for k in range(3):
    sbox(st[k], f"%{12 + 2 * k}", f"%{13 + 2 * k}", (f"%{2 * k}", f"%{2 * k + 1}"))
write("poseidon_sbox3_asm.inc", schedule(st),
      ["three x^7 S-boxes, interleaved; %0-%5 out halves (lo,hi) x3, %6-%11 carry SGPR pairs (two per stream), %12-%17 in halves (lo,hi) x3; clobbers v92-v127"])
# sbox1: %0,%1 out halves, %2,%3 carry pairs, %4,%5 in halves
s1 = Stream(SB_BASE[2], "%2", "%3")
sbox(s1, "%4", "%5", ("%0", "%1"))
write("poseidon_sbox1_asm.inc", schedule([s1]), ["one x^7 S-box; %0,%1 out halves, %2,%3 carry SGPR pairs, %4,%5 in halves; clobbers v116-v127"])
# fold3: %0..%2 out (u64), %3..%5 carry pairs, %6..%14 = al (u64), ah.lo, ah.hi x3; windows of 4 VGPRs inside v116..v127
FB = [116, 120, 124]
sf = [Stream(FB[k], f"%{3 + k}", None) for k in range(3)]
for k in range(3):
    fold(sf[k], f"%{6 + 3 * k}", f"%{7 + 3 * k}", f"%{8 + 3 * k}", f"%{k}")
write("poseidon_fold3_asm.inc", schedule(sf),
      ["three MDS output folds, interleaved; %0-%2 out (u64), %3-%5 carry SGPR pairs, %6-%14 = al (u64) ah.lo ah.hi x3; clobbers v116-v127"])
# --- generic canonical multiply (kept for A/B measurements; gl.cuh builds operator* from single-instruction primitives) ----
# %0,%1 = result halves (canonical), %2,%3 = carry SGPR pairs, %4,%5 = a halves, %6,%7 = b halves; window v116..v127
sm = Stream(116, "%2", "%3")
sm.emit(f"v_mov_b32 {sm.hi(ZY)}, 0")
X = S2
mul(sm, ("%4", "%5"), ("%6", "%7"), (sm.lo(X), sm.hi(X)))            # weak product in X
c = sm.ca
sm.emit(f"v_add_co_u32_e64 {sm.lo(P0)}, {c}, {sm.lo(X)}, -1", wr=[c])        # t = w + (2^32 - 1): carries out iff w >= p,
sm.emit(f"v_addc_co_u32_e64 {sm.hi(P0)}, {c}, {sm.hi(X)}, 0, {c}", rd=[c], wr=[c])   # and then t mod 2^64 = w - p
sm.emit(f"v_cndmask_b32_e64 %0, {sm.lo(X)}, {sm.lo(P0)}, {c}", rd=[c])
sm.emit(f"v_cndmask_b32_e64 %1, {sm.hi(X)}, {sm.hi(P0)}, {c}", rd=[c])
write("gl_mul_asm.inc", schedule([sm]), ["canonical Goldilocks multiply; %0,%1 out halves, %2,%3 carry SGPR pairs, %4,%5 = a, %6,%7 = b; clobbers v116-v127"])

# fold1: one output (the single row of the fused partial rounds): %0 out, %1 carry pair, %2 = al (u64), %3, %4 = ah.lo ah.hi
s1f = Stream(124, "%1", None)
fold(s1f, "%2", "%3", "%4", "%0")
write("poseidon_fold1_asm.inc", schedule([s1f]), ["one MDS output fold; %0 out (u64), %1 carry SGPR pair, %2 = al (u64), %3,%4 = ah.lo ah.hi; clobbers v124-v127"])

with open(os.path.join(CSRC, "poseidon_asm_clobbers.inc"), "w") as f:
    f.write("// Generated by tools/gen_poseidon_sbox_asm.py -- do not edit.\n")
    for name, lo in (("PW_CLOBBER_SBOX3", 92), ("PW_CLOBBER_SBOX1", 116), ("PW_CLOBBER_FOLD3", 116), ("PW_CLOBBER_FOLD1", 124), ("GL_CLOBBER_MUL", 116)):
        f.write("#define %s %s\n" % (name, ", ".join('"v%d"' % r for r in range(lo, 128))))
