// BN254 base-field arithmetic and the modular-gadget witness, shared by the host trace generator (tracegen.hip) and
// the device trace generator (kernels_tracegen.cuh): one implementation compiled for both sides, so host and device
// witnesses agree by construction.  Fq = 4x64-bit Montgomery (R = 2^256); the signed quotient of a modular gadget
// (src/modular/modular.rs:38-100) is an exact division by p done limb-wise from the low end.
#pragma once
#include "gl.cuh"
#include "air.cuh"
#include <string.h>

namespace bnw {
typedef unsigned __int128 u128;


// ---- Fq ---------------------------------------------------------------------------------------------
struct Fq { u64 l[4]; };
#define BNW_PL {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}
static constexpr u64 BNW_N0INV = 0x87d20782e4866389ULL;  // -p^-1 mod 2^64
#define BNW_R1 {{0xd35d438dc58f0d9dULL, 0x0a78eb28f5c70b3dULL, 0x666ea36f7879462cULL, 0x0e0a77c19a07df2fULL}}  // 2^256 mod p
#define BNW_R2 {{0xf32cfc5b538afa89ULL, 0xb5e71911d44501fbULL, 0x47ab1eff0a417ff6ULL, 0x06d89f71cab8351fULL}}  // 2^512 mod p

GL_HD bool geq_p(const u64* a) {
  constexpr u64 PL[4] = BNW_PL;
  for (int i = 3; i >= 0; i--) { if (a[i] != PL[i]) return a[i] > PL[i]; }
  return true;
}
GL_HD u64 add4(u64* r, const u64* a, const u64* b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)a[i] + b[i]; r[i] = (u64)c; c >>= 64; } return (u64)c; }
GL_HD u64 sub4(u64* r, const u64* a, const u64* b) {
  u64 br = 0;
  for (int i = 0; i < 4; i++) { u128 t = (u128)a[i] - b[i] - br; r[i] = (u64)t; br = (u64)(t >> 64) & 1; }
  return br;
}
GL_HD Fq mmul(const Fq& a, const Fq& b) {
  constexpr u64 PL[4] = BNW_PL;
  const u64 n0 = BNW_N0INV;
  u64 t[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128 c = 0; u64 hi;
    for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (u64)c; c >>= 64; }
    c += t[4]; t[4] = (u64)c; hi = (u64)(c >> 64);
    u64 mq = t[0] * n0;
    c = ((u128)mq * PL[0] + t[0]) >> 64;
    for (int j = 1; j < 4; j++) { c += (u128)mq * PL[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
    c += t[4]; t[3] = (u64)c; t[4] = hi + (u64)(c >> 64);
  }
  Fq r; memcpy(r.l, t, 32);
  if (t[4] || geq_p(r.l)) sub4(r.l, r.l, PL);
  return r;
}
GL_HD Fq to_m(const u64* x) { Fq a; for (int i = 0; i < 4; i++) a.l[i] = x[i]; Fq r2 = BNW_R2; return mmul(a, r2); }
GL_HD void from_m(const Fq& a, u64* out) { Fq o = {{1, 0, 0, 0}}; Fq r = mmul(a, o); for (int i = 0; i < 4; i++) out[i] = r.l[i]; }
GL_HD Fq fq_one() { Fq r = BNW_R1; return r; }
GL_HD Fq fadd(const Fq& a, const Fq& b) { constexpr u64 PL[4] = BNW_PL; Fq r; u64 c = add4(r.l, a.l, b.l); if (c || geq_p(r.l)) sub4(r.l, r.l, PL); return r; }
GL_HD Fq fsub(const Fq& a, const Fq& b) { constexpr u64 PL[4] = BNW_PL; Fq r; if (sub4(r.l, a.l, b.l)) add4(r.l, r.l, PL); return r; }
GL_HD bool fzero(const Fq& a) { return !(a.l[0] | a.l[1] | a.l[2] | a.l[3]); }

// Inverse of a standard-form value via binary extended GCD; returns standard form (value < p, nonzero).
GL_HD void inv_std(const u64* a_in, u64* out) {
  constexpr u64 PL[4] = BNW_PL;
  u64 u[4], v[4], x1[4] = {1, 0, 0, 0}, x2[4] = {0, 0, 0, 0};
  for (int i = 0; i < 4; i++) { u[i] = a_in[i]; v[i] = PL[i]; }
  auto is_one = [](const u64* a) { return a[0] == 1 && !(a[1] | a[2] | a[3]); };
  auto shr1 = [](u64* a, u64 top) { for (int i = 0; i < 3; i++) a[i] = (a[i] >> 1) | (a[i + 1] << 63); a[3] = (a[3] >> 1) | (top << 63); };
  auto halve = [&](u64* x) { if (x[0] & 1) { u64 c = add4(x, x, PL); shr1(x, c); } else shr1(x, 0); };
  auto geq = [](const u64* a, const u64* b) { for (int i = 3; i >= 0; i--) if (a[i] != b[i]) return a[i] > b[i]; return true; };
  while (!is_one(u) && !is_one(v)) {
    while (!(u[0] & 1)) { shr1(u, 0); halve(x1); }
    while (!(v[0] & 1)) { shr1(v, 0); halve(x2); }
    if (geq(u, v)) { sub4(u, u, v); if (sub4(x1, x1, x2)) add4(x1, x1, PL); }
    else { sub4(v, v, u); if (sub4(x2, x2, x1)) add4(x2, x2, PL); }
  }
  const u64* res = is_one(u) ? x1 : x2;
  for (int i = 0; i < 4; i++) out[i] = res[i];
}
GL_HD void limbs16(const u64* x, int64_t* out) { for (int i = 0; i < 16; i++) out[i] = (int64_t)((x[i / 4] >> (16 * (i % 4))) & 0xffff); }

// ---- modular gadget witness (src/modular/modular.rs:38-100, modular_zero.rs:33-80) -----------------------
struct ModW { int64_t quot_abs[17]; int sign; int64_t aux_lo[31], aux_hi[31]; int64_t out_aux_red[16]; };


// pol_input: 31 signed coefficients; output: value of the gadget's output (standard form, < p; zero for mod-zero).
GL_HD bool mod_witness(const int64_t* pol_input, const u64* output, bool has_output, ModW& w) {
  constexpr u64 PL[4] = BNW_PL;
  int64_t ml[16];
  for (int j = 0; j < 16; j++) ml[j] = (int64_t)bn254_modulus_limb(j);
  // d = P(2^16) - output as 9-limb two's complement
  u64 d[9] = {0};
  for (int i = 0; i < 31; i++) {
    int64_t c = pol_input[i];
    unsigned sh = 16 * i, ls = sh / 64, bs = sh % 64;
    u64 ext = c < 0 ? ~0ULL : 0;
    u128 carry = 0;
    for (unsigned k = ls; k < 9; k++) {
      unsigned q = k - ls;
      u64 cur = q == 0 ? (u64)c : ext, prev = q == 0 ? 0 : (q == 1 ? (u64)c : ext);
      u64 word = bs ? ((cur << bs) | (q ? (prev >> (64 - bs)) : 0)) : cur;
      carry += (u128)d[k] + word; d[k] = (u64)carry; carry >>= 64;
    }
  }
  if (has_output) {
    u64 br = 0;
    for (int k = 0; k < 9; k++) { u64 o = k < 4 ? output[k] : 0; u128 t = (u128)d[k] - o - br; d[k] = (u64)t; br = (u64)(t >> 64) & 1; }
  }
  // exact division by p, low limb first: q_i = d_i * p^-1 mod 2^64 ; d -= q_i * p << 64 i
  const u64 pinv = 0 - BNW_N0INV;
  u64 q[5];
  for (int i = 0; i < 5; i++) {
    q[i] = d[i] * pinv;
    u128 br = 0;
    for (int j = 0; j < 4 && i + j < 9; j++) {
      u128 prod = (u128)q[i] * PL[j] + br;
      u64 lo = (u64)prod; br = prod >> 64;
      if (d[i + j] < lo) br += 1;
      d[i + j] -= lo;
    }
    for (int k = i + 4; k < 9 && br; k++) { u64 b = (u64)br; br = d[k] < b ? 1 : 0; d[k] -= b; }
  }
  // remaining high limbs: 0 if q >= 0, or -(p << 320) if q < 0 (q is two's complement in 320 bits)
  bool neg = q[4] >> 63;
  {
    u64 exp[4] = {0, 0, 0, 0};
    if (neg) { u64 z[4] = {0, 0, 0, 0}; sub4(exp, z, PL); }
    for (int k = 0; k < 5; k++) if (d[k]) return false;
    for (int k = 0; k < 4; k++) if (d[5 + k] != exp[k]) return false;  // input was not congruent to output mod p
  }
  if (neg) { u128 c = 1; for (int i = 0; i < 5; i++) { c += (u64)~q[i]; q[i] = (u64)c; c >>= 64; } }
  w.sign = neg ? -1 : 1;
  for (int i = 0; i < 17; i++) w.quot_abs[i] = (int64_t)((q[i / 4] >> (16 * (i % 4))) & 0xffff);
  for (int i = 17; i < 20; i++) if ((q[i / 4] >> (16 * (i % 4))) & 0xffff) return false;
  int64_t outl[16] = {0};
  if (has_output) {
    limbs16(output, outl);
    u64 t[4]; sub4(t, output, PL);  // 2^256 - p + output
    limbs16(t, w.out_aux_red);
  }
  int64_t constr[32];
  for (int i = 0; i < 31; i++) constr[i] = pol_input[i];
  constr[31] = 0;
  for (int i = 0; i < 16; i++) constr[i] -= outl[i];
  for (int i = 0; i < 17; i++) { int64_t qi = w.sign * w.quot_abs[i]; if (qi) for (int j = 0; j < 16; j++) constr[i + j] -= qi * ml[j]; }
  int64_t prev = 0;
  for (int k = 0; k < 31; k++) {
    int64_t a = k == 0 ? -(constr[0] >> 16) : (prev - constr[k]) >> 16;  // pol_remove_root_2exp
    prev = a;
    int64_t s = a + (1 << 29);
    if (s < 0 || s > (1 << 30)) return false;
    w.aux_lo[k] = s & 0xffff; w.aux_hi[k] = (s >> 16) & 0xffff;
  }
  return true;
}

GL_HD void conv16(const int64_t* a, const int64_t* b, int64_t* out) {
  for (int k = 0; k < 31; k++) out[k] = 0;
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) out[i + j] += a[i] * b[j];
}

// Writes the 320 G1Output columns (muladd.rs:79-94) for one add / double given lambda (standard form).
GL_HD bool g1_output_row(bool is_double, const u64* ax, const u64* ay, const u64* bx, const u64* by, const u64* lam, const u64* nx, const u64* ny, u64* lv) {
  int64_t l[16], axl[16], ayl[16], bxl[16], byl[16], nxl[16], t16[16], c[31], zero_pol[31], inx[31], iny[31];
  limbs16(lam, l); limbs16(ax, axl); limbs16(ay, ayl); limbs16(nx, nxl);
  if (is_double) { for (int i = 0; i < 16; i++) { bxl[i] = axl[i]; byl[i] = ayl[i]; } } else { limbs16(bx, bxl); limbs16(by, byl); }
  if (is_double) {
    conv16(l, ayl, zero_pol); for (int k = 0; k < 31; k++) zero_pol[k] *= 2;
    conv16(axl, axl, c); for (int k = 0; k < 31; k++) zero_pol[k] -= 3 * c[k];
  } else {
    for (int i = 0; i < 16; i++) t16[i] = bxl[i] - axl[i];
    conv16(l, t16, zero_pol); for (int i = 0; i < 16; i++) zero_pol[i] -= byl[i] - ayl[i];
  }
  conv16(l, l, inx); for (int i = 0; i < 16; i++) inx[i] -= axl[i] + bxl[i];
  for (int i = 0; i < 16; i++) t16[i] = axl[i] - nxl[i];
  conv16(l, t16, iny); for (int i = 0; i < 16; i++) iny[i] -= ayl[i];
  ModW wz, wx, wy;
  if (!mod_witness(zero_pol, nullptr, false, wz) || !mod_witness(inx, nx, true, wx) || !mod_witness(iny, ny, true, wy)) return false;
  int cur = 0;
  auto put16 = [&](const int64_t* v) { for (int i = 0; i < 16; i++) lv[cur++] = (u64)v[i]; };
  int64_t nyl[16]; limbs16(ny, nyl);
  put16(l); put16(nxl); put16(nyl);
  auto put_aux = [&](const ModW& w, bool oar) {
    if (oar) put16(w.out_aux_red);
    for (int i = 0; i < 17; i++) lv[cur++] = (u64)w.quot_abs[i];
    for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_lo[i];
    for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_hi[i];
  };
  put_aux(wz, false); put_aux(wx, true); put_aux(wy, true);
  lv[cur++] = wz.sign > 0 ? 1 : GLP - 1; lv[cur++] = wx.sign > 0 ? 1 : GLP - 1; lv[cur++] = wy.sign > 0 ? 1 : GLP - 1;
  return cur == 320;
}

// ---- curve chains of one G1ExpStark instance ---------------------------------------------------------------------
// Strided Fq access: limb i of entry idx lives at base[i * stride + idx].
GL_HD Fq ldq(const u64* base, size_t idx, size_t stride) { Fq r; for (int i = 0; i < 4; i++) r.l[i] = base[i * stride + idx]; return r; }
GL_HD void stq(u64* base, size_t idx, size_t stride, const Fq& v) { for (int i = 0; i < 4; i++) base[i * stride + idx] = v.l[i]; }
GL_HD void u32x8_to_u64x4(const uint32_t* w, u64* out) { for (int i = 0; i < 4; i++) out[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32); }
enum { TG_ERR_DEGENERATE = 1, TG_ERR_WITNESS = 2, TG_ERR_RANGE = 4 };

struct Jac { Fq X, Y, Z; };
// dbl-2009-l for y^2 = x^3 + b: 2M + 5S.
GL_HD Jac jac_double(const Jac& p) {
  Fq A = mmul(p.X, p.X), B = mmul(p.Y, p.Y), C = mmul(B, B);
  Fq t = fadd(p.X, B); t = mmul(t, t); t = fsub(fsub(t, A), C);
  Fq D = fadd(t, t), E = fadd(fadd(A, A), A), F = mmul(E, E);
  Jac r;
  r.X = fsub(F, fadd(D, D));
  Fq C8 = fadd(C, C); C8 = fadd(C8, C8); C8 = fadd(C8, C8);
  r.Y = fsub(mmul(E, fsub(D, r.X)), C8);
  Fq yz = mmul(p.Y, p.Z);
  r.Z = fadd(yz, yz);
  return r;
}
// add-2007-bl (both Jacobian): 11M + 5S.  *degenerate is set when the x coordinates agree (H = 0).
GL_HD Jac jac_add(const Jac& p, const Jac& q, bool* degenerate) {
  Fq Z1Z1 = mmul(p.Z, p.Z), Z2Z2 = mmul(q.Z, q.Z);
  Fq U1 = mmul(p.X, Z2Z2), U2 = mmul(q.X, Z1Z1);
  Fq S1 = mmul(mmul(p.Y, q.Z), Z2Z2), S2 = mmul(mmul(q.Y, p.Z), Z1Z1);
  Fq H = fsub(U2, U1);
  *degenerate = fzero(H);
  Fq I = fadd(H, H); I = mmul(I, I);
  Fq J = mmul(H, I);
  Fq r = fsub(S2, S1); r = fadd(r, r);
  Fq V = mmul(U1, I);
  Jac o;
  o.X = fsub(fsub(mmul(r, r), J), fadd(V, V));
  Fq sj = mmul(S1, J);
  o.Y = fsub(mmul(r, fsub(V, o.X)), fadd(sj, sj));
  Fq zz = fadd(p.Z, q.Z); zz = mmul(zz, zz); zz = fsub(fsub(zz, Z1Z1), Z2Z2);
  o.Z = mmul(zz, H);
  return o;
}


// Jacobian chain storage, instance-major (one host thread or one lane writes a contiguous 24 KB run):
// entry (t, c) of instance k, t = 0..256, c = X,Y,Z -> 4 consecutive words at base + jac_at(k, t, c)
GL_HD size_t jac_at(size_t k, int t, int c) { return (k * 257 + (size_t)t) * 12 + (size_t)c * 4; }
// io: x.x x.y offset.x offset.y exp_val (8 u32 each).  A[t] = 2^t x (t = 0..256), B[0] = offset,
// B[t+1] = bit_t ? B[t] + A[t] : B[t]   (g1/exp.rs:165-230: even rows add-if-bit, odd rows double), all in Jacobian
// coordinates and Montgomery form, so no step needs an inversion.  Returns TG_ERR_* flags.
GL_HD int g1_chains(const uint32_t* io, size_t k, u64* ja, u64* jb) {
  u64 t4[4];
  Jac a, b;
  u32x8_to_u64x4(io, t4); a.X = to_m(t4);
  u32x8_to_u64x4(io + 8, t4); a.Y = to_m(t4);
  u32x8_to_u64x4(io + 16, t4); b.X = to_m(t4);
  u32x8_to_u64x4(io + 24, t4); b.Y = to_m(t4);
  a.Z = fq_one(); b.Z = fq_one();
  int bad = 0;
  for (int t = 0;; t++) {
    stq(ja + jac_at(k, t, 0), 0, 1, a.X); stq(ja + jac_at(k, t, 1), 0, 1, a.Y); stq(ja + jac_at(k, t, 2), 0, 1, a.Z);
    stq(jb + jac_at(k, t, 0), 0, 1, b.X); stq(jb + jac_at(k, t, 1), 0, 1, b.Y); stq(jb + jac_at(k, t, 2), 0, 1, b.Z);
    if (t == 256) break;
    const bool bit = (io[32 + (t >> 5)] >> (t & 31)) & 1;
    if (bit) { bool deg; b = jac_add(b, a, &deg); if (deg) bad |= TG_ERR_DEGENERATE; }
    if (fzero(a.Y)) bad |= TG_ERR_DEGENERATE;
    a = jac_double(a);
  }
  return bad;
}

}  // namespace bnw
