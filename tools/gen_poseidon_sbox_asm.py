#!/usr/bin/env python3
"""Generates the gfx950 instruction streams of the Goldilocks x^7 S-box and of the MDS output fold:

  poseidon_sbox3_asm.inc   three S-boxes interleaved instruction by instruction (full rounds)
  poseidon_sbox1_asm.inc   one S-box (partial rounds)
  poseidon_fold3_asm.inc   three MDS outputs  lo + hi*2^32 -> one weak u64 each

Why by hand: the compiler's Goldilocks multiply costs ~29 instructions (the low and high halves of the 128-bit
product are computed separately, every carry becomes a 64-bit compare + select, and each zero-extended addend of
v_mad_u64_u32 costs two moves).  Here a multiply is 16 instructions:

  P0 = a0*b0                       v_mad_u64_u32
  P1 = a0*b1 + hi(P0)              mov + v_mad_u64_u32      (the addend pair is {value, 0}: its odd register stays 0)
  P2 = a1*b0 + lo(P1)              mov + v_mad_u64_u32
  P3 = a1*b1 + hi(P1) + hi(P2)     add, addc, v_mad_u64_u32 ->  a*b = lo(P0) + 2^32 lo(P2) + 2^64 P3
  x  = (lo(P2):lo(P0)) - hi(P3)    sub, subb; on borrow add p = (0xffffffff:1) : cndmask, addc, addc   [2^96 = -1]
  r  = lo(P3)*(2^32-1) + x         v_mad_u64_u32 with carry-out                                       [2^64 = 2^32-1]
  r += carry ? 2^32-1 : 0          cndmask, v_lshl_add_u64

Registers: 64-bit operands must be even-aligned pairs whose halves are named separately, which inline-asm operands
cannot express, so every stream works in a fixed window of physical VGPRs (clobbered), carries live in per-stream SGPR
pairs (asm operands).  gfx950 needs two wait states between a VALU writing an SGPR and a VALU reading it; the three
interleaved streams provide them (a single stream pads with s_nop).
"""
import os

CSRC = os.environ.get("SBN_GEN_OUT") or os.path.join(os.path.dirname(__file__), "..", "starky_bn254_amd", "csrc")


class Stream:
    """Instruction list of one dependent chain; each entry = (text, reads_carry, writes_carry)."""

    def __init__(self, base, carry):
        self.b, self.c, self.ins = base, carry, []

    def pair(self, k):
        return f"v[{self.b + 2 * k}:{self.b + 2 * k + 1}]"

    def lo(self, k):
        return f"v{self.b + 2 * k}"

    def hi(self, k):
        return f"v{self.b + 2 * k + 1}"

    def emit(self, text, rd=False, wr=False):
        self.ins.append((text, rd, wr))


Z, Y, Q, P0, P1, P2, P3, X, S2, S3, S4 = range(11)   # pair indices inside a stream's window (22 VGPRs)


def mul(s, a, b, dst):
    """a, b = (lo, hi) register names; dst = destination pair text."""
    c = s.c
    s.emit(f"v_mad_u64_u32 {s.pair(P0)}, {c}, {a[0]}, {b[0]}, 0", wr=True)
    s.emit(f"v_mov_b32 {s.lo(Z)}, {s.hi(P0)}")
    s.emit(f"v_mad_u64_u32 {s.pair(P1)}, {c}, {a[0]}, {b[1]}, {s.pair(Z)}", wr=True)
    s.emit(f"v_mov_b32 {s.lo(Y)}, {s.lo(P1)}")
    s.emit(f"v_mad_u64_u32 {s.pair(P2)}, {c}, {a[1]}, {b[0]}, {s.pair(Y)}", wr=True)
    s.emit(f"v_add_co_u32_e64 {s.lo(Q)}, {c}, {s.hi(P1)}, {s.hi(P2)}", wr=True)
    s.emit(f"v_addc_co_u32_e64 {s.hi(Q)}, {c}, 0, 0, {c}", rd=True, wr=True)
    s.emit(f"v_mad_u64_u32 {s.pair(P3)}, {c}, {a[1]}, {b[1]}, {s.pair(Q)}", wr=True)
    s.emit(f"v_sub_co_u32_e64 {s.lo(X)}, {c}, {s.lo(P0)}, {s.hi(P3)}", wr=True)
    s.emit(f"v_subbrev_co_u32_e64 {s.hi(X)}, {c}, 0, {s.lo(P2)}, {c}", rd=True, wr=True)
    s.emit(f"v_cndmask_b32_e64 {s.lo(Z)}, 0, -1, {c}", rd=True)
    s.emit(f"v_addc_co_u32_e64 {s.lo(X)}, {c}, {s.lo(X)}, 0, {c}", rd=True, wr=True)
    s.emit(f"v_addc_co_u32_e64 {s.hi(X)}, {c}, {s.hi(X)}, {s.lo(Z)}, {c}", rd=True, wr=True)
    s.emit(f"v_mad_u64_u32 {s.pair(P1)}, {c}, {s.lo(P3)}, -1, {s.pair(X)}", wr=True)
    s.emit(f"v_cndmask_b32_e64 {s.lo(Z)}, 0, -1, {c}", rd=True)
    s.emit(f"v_lshl_add_u64 {dst}, {s.pair(P1)}, 0, {s.pair(Z)}")


def sbox(s, x0, x1, out):
    s.emit(f"v_mov_b32 {s.hi(Z)}, 0")
    s.emit(f"v_mov_b32 {s.hi(Y)}, 0")
    x = (x0, x1)
    mul(s, x, x, s.pair(S2))
    x2 = (s.lo(S2), s.hi(S2))
    mul(s, x2, x2, s.pair(S4))
    mul(s, x2, x, s.pair(S3))
    mul(s, (s.lo(S4), s.hi(S4)), (s.lo(S3), s.hi(S3)), out)


def fold(s, al0, al1, ah0, ah1, out):
    """out = al + ah*2^32 (al, ah < 2^42) as a weak u64: (al1 + ah0 : al0) + (ah1 + carry)*(2^32 - 1); the carry of that
    last product-sum is repaid by one more multiply-add with (2^32 - 1), which needs no zero-extended register pair."""
    c = s.c
    W, R = 0, 1
    s.emit(f"v_add_co_u32_e64 {s.hi(W)}, {c}, {al1}, {ah0}", wr=True)
    s.emit(f"v_mov_b32 {s.lo(W)}, {al0}")
    s.emit(f"v_addc_co_u32_e64 {s.lo(2)}, {c}, {ah1}, 0, {c}", rd=True, wr=True)
    s.emit(f"v_mad_u64_u32 {s.pair(R)}, {c}, {s.lo(2)}, -1, {s.pair(W)}", wr=True)
    s.emit(f"v_cndmask_b32_e64 {s.hi(2)}, 0, 1, {c}", rd=True)
    s.emit(f"v_mad_u64_u32 {out}, {c}, {s.hi(2)}, -1, {s.pair(R)}", wr=True)


def schedule(streams):
    """Round-robin merge; pads with s_nop so that a carry reader sits >= 3 slots after its writer."""
    out, last_write = [], {}
    idx = [0] * len(streams)
    while any(idx[k] < len(s.ins) for k, s in enumerate(streams)):
        for k, s in enumerate(streams):
            if idx[k] >= len(s.ins):
                continue
            text, rd, wr = s.ins[idx[k]]
            if rd and k in last_write:
                gap = len(out) - last_write[k] - 1      # instructions between writer and reader
                if gap < 2:
                    out.append(f"s_nop {1 - gap}")
            out.append(text)
            if wr:
                last_write[k] = len(out) - 1
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


SB_BASE = [62, 84, 106]     # three windows of 22 VGPRs: v62..v127
# sbox3: operands %0..%2 = outputs (u64), %3..%5 = carry SGPR pairs (scratch outputs), %6..%11 = lo,hi of each input
st = [Stream(SB_BASE[k], f"%{3 + k}") for k in range(3)]
for k in range(3):
    sbox(st[k], f"%{6 + 2 * k}", f"%{7 + 2 * k}", f"%{k}")
write("poseidon_sbox3_asm.inc", schedule(st),
      ["three x^7 S-boxes, interleaved; %0-%2 out (u64), %3-%5 carry SGPR pairs, %6-%11 in halves (lo,hi) x3; clobbers v62-v127"])
# sbox1: %0 out, %1 carry pair, %2,%3 in halves
s1 = Stream(SB_BASE[2], "%1")
sbox(s1, "%2", "%3", "%0")
write("poseidon_sbox1_asm.inc", schedule([s1]), ["one x^7 S-box; %0 out (u64), %1 carry SGPR pair, %2,%3 in halves; clobbers v106-v127"])
# fold3: %0..%2 out (u64), %3..%5 carry pairs, %6..%17 = al0,al1,ah0,ah1 x3; windows of 6 VGPRs inside v104..v127
FB = [104, 112, 120]
sf = [Stream(FB[k], f"%{3 + k}") for k in range(3)]
for k in range(3):
    fold(sf[k], f"%{6 + 4 * k}", f"%{7 + 4 * k}", f"%{8 + 4 * k}", f"%{9 + 4 * k}", f"%{k}")
write("poseidon_fold3_asm.inc", schedule(sf),
      ["three MDS output folds, interleaved; %0-%2 out (u64), %3-%5 carry SGPR pairs, %6-%17 = al.lo al.hi ah.lo ah.hi x3; clobbers v104-v127"])
# --- generic canonical multiply (gl.cuh operator* on the device): one stream, window v112..v127 --------------------------
# %0,%1 = result halves (canonical), %2 = carry SGPR pair, %3,%4 = a halves, %5,%6 = b halves
sm = Stream(112, "%2")
sm.emit(f"v_mov_b32 {sm.hi(Z)}, 0")
sm.emit(f"v_mov_b32 {sm.hi(Y)}, 0")
mul(sm, ("%3", "%4"), ("%5", "%6"), sm.pair(X))            # weak product in X
c = sm.c
sm.emit(f"v_add_co_u32_e64 {sm.lo(P0)}, {c}, {sm.lo(X)}, -1", wr=True)        # t = w + (2^32 - 1): carries out iff w >= p,
sm.emit(f"v_addc_co_u32_e64 {sm.hi(P0)}, {c}, {sm.hi(X)}, 0, {c}", rd=True, wr=True)   # and then t mod 2^64 = w - p
sm.emit(f"v_cndmask_b32_e64 %0, {sm.lo(X)}, {sm.lo(P0)}, {c}", rd=True)
sm.emit(f"v_cndmask_b32_e64 %1, {sm.hi(X)}, {sm.hi(P0)}, {c}", rd=True)
write("gl_mul_asm.inc", schedule([sm]), ["canonical Goldilocks multiply; %0,%1 out halves, %2 carry SGPR pair, %3,%4 = a, %5,%6 = b; clobbers v112-v127"])

# fold1: one output (the single row of the fused partial rounds): %0 out, %1 carry pair, %2..%5 = al0 al1 ah0 ah1
s1f = Stream(120, "%1")
fold(s1f, "%2", "%3", "%4", "%5", "%0")
write("poseidon_fold1_asm.inc", schedule([s1f]), ["one MDS output fold; %0 out (u64), %1 carry SGPR pair, %2-%5 = al.lo al.hi ah.lo ah.hi; clobbers v120-v127"])

with open(os.path.join(CSRC, "poseidon_asm_clobbers.inc"), "w") as f:
    f.write("// Generated by tools/gen_poseidon_sbox_asm.py -- do not edit.\n")
    for name, lo in (("PW_CLOBBER_SBOX3", 62), ("PW_CLOBBER_SBOX1", 106), ("PW_CLOBBER_FOLD3", 104), ("PW_CLOBBER_FOLD1", 120), ("GL_CLOBBER_MUL", 112)):
        f.write("#define %s %s\n" % (name, ", ".join('"v%d"' % r for r in range(lo, 128))))
