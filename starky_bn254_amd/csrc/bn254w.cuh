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
#if defined(__HIP_DEVICE_COMPILE__)
// The same Montgomery product (R = 2^256, canonical result) on eight 32-bit limbs: every inner step is one v_mad_u64_u32
// (a_j b_i + t_j + carry never exceeds 2^64 - 1) plus the add that forms its 64-bit addend, fully unrolled, where the
// 64-bit-limb form below compiles to ~700 instructions of emulated 128-bit arithmetic.  The device witness kernels wait for
// ONE chain of ~350 dependent products (the Fermat inversion of a batch), so the length of a product is their latency.
GL_HD Fq mmul(const Fq& a, const Fq& b) {   // (host-device only so that the host functions parsed in this pass resolve it)
  constexpr u64 PL[4] = BNW_PL;
  const uint32_t n0 = (uint32_t)BNW_N0INV;   // -p^-1 mod 2^32
  uint32_t al[8], bl[8], pl[8], t[10];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    al[2 * i] = (uint32_t)a.l[i]; al[2 * i + 1] = (uint32_t)(a.l[i] >> 32);
    bl[2 * i] = (uint32_t)b.l[i]; bl[2 * i + 1] = (uint32_t)(b.l[i] >> 32);
    pl[2 * i] = (uint32_t)PL[i]; pl[2 * i + 1] = (uint32_t)(PL[i] >> 32);
  }
#pragma unroll
  for (int i = 0; i < 10; i++) t[i] = 0;
#pragma unroll
  for (int i = 0; i < 8; i++) {
    u64 c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { const u64 cs = (u64)al[j] * bl[i] + ((u64)t[j] + c); t[j] = (uint32_t)cs; c = cs >> 32; }
    u64 cs = (u64)t[8] + c; t[8] = (uint32_t)cs; t[9] = (uint32_t)(cs >> 32);
    const uint32_t mq = t[0] * n0;
    c = ((u64)mq * pl[0] + t[0]) >> 32;
#pragma unroll
    for (int j = 1; j < 8; j++) { cs = (u64)mq * pl[j] + ((u64)t[j] + c); t[j - 1] = (uint32_t)cs; c = cs >> 32; }
    cs = (u64)t[8] + c; t[7] = (uint32_t)cs; t[8] = t[9] + (uint32_t)(cs >> 32);
  }
  Fq r;
#pragma unroll
  for (int i = 0; i < 4; i++) r.l[i] = (u64)t[2 * i] | ((u64)t[2 * i + 1] << 32);
  if (t[8] || geq_p(r.l)) sub4(r.l, r.l, PL);
  return r;
}
#else
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
#endif
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

// ---- Fq2 limb products and the G2 gadget row ---------------------------------------------------------------------------
// Fq2 limb product (src/fields/fq2.rs:41-58): c0 = x0*y0 - x1*y1, c1 = x0*y1 + x1*y0.
GL_HD void conv16_fq2(const int64_t x[2][16], const int64_t y[2][16], int64_t out[2][31]) {
  int64_t t[31];
  conv16(x[0], y[0], out[0]); conv16(x[1], y[1], t); for (int k = 0; k < 31; k++) out[0][k] -= t[k];
  conv16(x[0], y[1], out[1]); conv16(x[1], y[0], t); for (int k = 0; k < 31; k++) out[1][k] += t[k];
}
// Writes the 640 G2Output columns (src/curves/g2/muladd.rs:56-80) for one add / double.
// Each value is an Fq2 given as two standard-form Fq (c0 = v, c1 = v + 4).
GL_HD bool g2_output_row(bool is_double, const u64* ax, const u64* ay, const u64* bx, const u64* by, const u64* lam, const u64* nx, const u64* ny, u64* lv) {
  int64_t l[2][16], axl[2][16], ayl[2][16], bxl[2][16], byl[2][16], nxl[2][16], nyl[2][16], t16[2][16];
  int64_t zero_pol[2][31], inx[2][31], iny[2][31], c[2][31];
  for (int q = 0; q < 2; q++) {
    limbs16(lam + 4 * q, l[q]); limbs16(ax + 4 * q, axl[q]); limbs16(ay + 4 * q, ayl[q]); limbs16(nx + 4 * q, nxl[q]); limbs16(ny + 4 * q, nyl[q]);
    if (is_double) { for (int i = 0; i < 16; i++) { bxl[q][i] = axl[q][i]; byl[q][i] = ayl[q][i]; } } else { limbs16(bx + 4 * q, bxl[q]); limbs16(by + 4 * q, byl[q]); }
  }
  if (is_double) {
    conv16_fq2(l, ayl, zero_pol); conv16_fq2(axl, axl, c);
    for (int q = 0; q < 2; q++) for (int k = 0; k < 31; k++) zero_pol[q][k] = 2 * zero_pol[q][k] - 3 * c[q][k];
  } else {
    for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) t16[q][i] = bxl[q][i] - axl[q][i];
    conv16_fq2(l, t16, zero_pol);
    for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) zero_pol[q][i] -= byl[q][i] - ayl[q][i];
  }
  conv16_fq2(l, l, inx);
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) inx[q][i] -= axl[q][i] + bxl[q][i];
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) t16[q][i] = axl[q][i] - nxl[q][i];
  conv16_fq2(l, t16, iny);
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) iny[q][i] -= ayl[q][i];
  ModW wz[2], wx[2], wy[2];
  for (int q = 0; q < 2; q++)
    if (!mod_witness(zero_pol[q], nullptr, false, wz[q]) || !mod_witness(inx[q], nx + 4 * q, true, wx[q]) || !mod_witness(iny[q], ny + 4 * q, true, wy[q])) return false;
  int cur = 0;
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) lv[cur++] = (u64)l[q][i];
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) lv[cur++] = (u64)nxl[q][i];
  for (int q = 0; q < 2; q++) for (int i = 0; i < 16; i++) lv[cur++] = (u64)nyl[q][i];
  for (int blk = 0; blk < 3; blk++) {
    const ModW* ws = blk == 0 ? wz : (blk == 1 ? wx : wy);
    for (int q = 0; q < 2; q++) {
      const ModW& w = ws[q];
      if (blk) for (int i = 0; i < 16; i++) lv[cur++] = (u64)w.out_aux_red[i];
      for (int i = 0; i < 17; i++) lv[cur++] = (u64)w.quot_abs[i];
      for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_lo[i];
      for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_hi[i];
    }
  }
  for (int q = 0; q < 2; q++) lv[cur++] = wz[q].sign > 0 ? 1 : GLP - 1;
  for (int q = 0; q < 2; q++) lv[cur++] = wx[q].sign > 0 ? 1 : GLP - 1;
  for (int q = 0; q < 2; q++) lv[cur++] = wy[q].sign > 0 ? 1 : GLP - 1;
  return cur == 640;
}

// FqOutput columns (src/fields/fq/mul.rs:49-54) of x*y given the product (standard form): 112 words
GL_HD bool fq_output_row(const u64* x, const u64* y, const u64* out, u64* lv) {
  int64_t xl[16], yl[16], ol[16], pol[31];
  limbs16(x, xl); limbs16(y, yl); limbs16(out, ol);
  conv16(xl, yl, pol);
  ModW w;
  if (!mod_witness(pol, out, true, w)) return false;
  int cur = 0;
  for (int i = 0; i < 16; i++) lv[cur++] = (u64)ol[i];
  for (int i = 0; i < 16; i++) lv[cur++] = (u64)w.out_aux_red[i];
  for (int i = 0; i < 17; i++) lv[cur++] = (u64)w.quot_abs[i];
  for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_lo[i];
  for (int i = 0; i < 31; i++) lv[cur++] = (u64)w.aux_hi[i];
  lv[cur++] = w.sign > 0 ? 1 : GLP - 1;
  return cur == 112;
}

// ---- Fq12 (flat basis of plonky2-bn254 MyFq12: coefficient of w^k is c[k] + c[k+6] i, w^6 = 9 + i) ------------------------
// flat-basis product (src/fields/fq12/mul.rs:24-87) in Montgomery Fq
GL_HD void fq12_mul_m(const Fq* a, const Fq* b, Fq* out) {
  Fq z = {{0, 0, 0, 0}}, d[11], s[11];
  for (int m = 0; m < 11; m++) { d[m] = z; s[m] = z; }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      d[i + j] = fadd(d[i + j], fsub(mmul(a[i], b[j]), mmul(a[i + 6], b[j + 6])));
      s[i + j] = fadd(s[i + j], fadd(mmul(a[i], b[j + 6]), mmul(a[i + 6], b[j])));
    }
  for (int m = 0; m < 6; m++) {
    if (m < 5) {
      Fq t = fadd(d[m + 6], d[m + 6]); t = fadd(t, t); t = fadd(t, t); const Fq d9 = fadd(t, d[m + 6]);
      t = fadd(s[m + 6], s[m + 6]); t = fadd(t, t); t = fadd(t, t); const Fq s9 = fadd(t, s[m + 6]);
      out[m] = fsub(fadd(d[m], d9), s[m + 6]); out[m + 6] = fadd(fadd(s[m], d[m + 6]), s9);
    } else { out[m] = d[m]; out[m + 6] = s[m]; }
  }
}
// Fq12Output columns (mul.rs:217-231) of x*y given the product `out` (standard form): 1344 words, handed one by one to
// put(index, value) -- the host writes them into a row buffer, the device straight into the column-major trace.
template <typename Put>
GL_HD bool fq12_output_row(const u64 (*x)[4], const u64 (*y)[4], const u64 (*out)[4], Put put) {
  int64_t xl[12][16], yl[12][16], d[11][31], s[11][31], t[31];
  for (int c = 0; c < 12; c++) { limbs16(x[c], xl[c]); limbs16(y[c], yl[c]); }
  for (int m = 0; m < 11; m++) for (int k = 0; k < 31; k++) { d[m][k] = 0; s[m][k] = 0; }
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      conv16(xl[i], yl[j], t); for (int k = 0; k < 31; k++) d[i + j][k] += t[k];
      conv16(xl[i + 6], yl[j + 6], t); for (int k = 0; k < 31; k++) d[i + j][k] -= t[k];
      conv16(xl[i], yl[j + 6], t); for (int k = 0; k < 31; k++) s[i + j][k] += t[k];
      conv16(xl[i + 6], yl[j], t); for (int k = 0; k < 31; k++) s[i + j][k] += t[k];
    }
  for (int c = 0; c < 12; c++) {
    const int m = c < 6 ? c : c - 6;
    for (int k = 0; k < 31; k++) {
      if (m < 5) t[k] = c < 6 ? d[m][k] + 9 * d[m + 6][k] - s[m + 6][k] : s[m][k] + d[m + 6][k] + 9 * s[m + 6][k];
      else t[k] = c < 6 ? d[m][k] : s[m][k];
    }
    ModW w;
    if (!mod_witness(t, out[c], true, w)) return false;
    int64_t ol[16]; limbs16(out[c], ol);
    for (int i = 0; i < 16; i++) put(16 * c + i, (u64)ol[i]);
    int cur = 192 + 95 * c;
    for (int i = 0; i < 16; i++) put(cur++, (u64)w.out_aux_red[i]);
    for (int i = 0; i < 17; i++) put(cur++, (u64)w.quot_abs[i]);
    for (int i = 0; i < 31; i++) put(cur++, (u64)w.aux_lo[i]);
    for (int i = 0; i < 31; i++) put(cur++, (u64)w.aux_hi[i]);
    put(1332 + c, w.sign > 0 ? (u64)1 : GLP - 1);
  }
  return true;
}

// ---- curve chains of one G1ExpStark / G2ExpStark instance -----------------------------------------------------------------
// Coordinates: E = 1 -> Fq, E = 2 -> Fq2 = Fq[i]/(i^2+1), Montgomery form.
template <int E> struct Co { Fq c[E]; };
template <int E> GL_HD Co<E> cadd(const Co<E>& a, const Co<E>& b) { Co<E> r; for (int q = 0; q < E; q++) r.c[q] = fadd(a.c[q], b.c[q]); return r; }
template <int E> GL_HD Co<E> csub(const Co<E>& a, const Co<E>& b) { Co<E> r; for (int q = 0; q < E; q++) r.c[q] = fsub(a.c[q], b.c[q]); return r; }
GL_HD Co<1> cmul(const Co<1>& a, const Co<1>& b) { Co<1> r; r.c[0] = mmul(a.c[0], b.c[0]); return r; }
GL_HD Co<2> cmul(const Co<2>& a, const Co<2>& b) {
  Co<2> r;
  r.c[0] = fsub(mmul(a.c[0], b.c[0]), mmul(a.c[1], b.c[1]));
  r.c[1] = fadd(mmul(a.c[0], b.c[1]), mmul(a.c[1], b.c[0]));
  return r;
}
template <int E> GL_HD bool czero(const Co<E>& a) { for (int q = 0; q < E; q++) if (!fzero(a.c[q])) return false; return true; }
template <int E> GL_HD Co<E> cone() { Co<E> r; r.c[0] = fq_one(); for (int q = 1; q < E; q++) r.c[q] = Fq{{0, 0, 0, 0}}; return r; }

// Strided Fq access: limb i of entry idx lives at base[i * stride + idx].
GL_HD Fq ldq(const u64* base, size_t idx, size_t stride) { Fq r; for (int i = 0; i < 4; i++) r.l[i] = base[i * stride + idx]; return r; }
GL_HD void stq(u64* base, size_t idx, size_t stride, const Fq& v) { for (int i = 0; i < 4; i++) base[i * stride + idx] = v.l[i]; }
GL_HD void u32x8_to_u64x4(const uint32_t* w, u64* out) { for (int i = 0; i < 4; i++) out[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32); }
enum { TG_ERR_DEGENERATE = 1, TG_ERR_WITNESS = 2, TG_ERR_RANGE = 4 };

template <int E> struct Jac { Co<E> X, Y, Z; };
// dbl-2009-l for y^2 = x^3 + b (no curve constant involved): 2M + 5S.
template <int E> GL_HD Jac<E> jac_double(const Jac<E>& p) {
  Co<E> A = cmul(p.X, p.X), B = cmul(p.Y, p.Y), C = cmul(B, B);
  Co<E> t = cadd(p.X, B); t = cmul(t, t); t = csub(csub(t, A), C);
  Co<E> D = cadd(t, t), Ee = cadd(cadd(A, A), A), F = cmul(Ee, Ee);
  Jac<E> r;
  r.X = csub(F, cadd(D, D));
  Co<E> C8 = cadd(C, C); C8 = cadd(C8, C8); C8 = cadd(C8, C8);
  r.Y = csub(cmul(Ee, csub(D, r.X)), C8);
  Co<E> yz = cmul(p.Y, p.Z);
  r.Z = cadd(yz, yz);
  return r;
}
// add-2007-bl (both Jacobian): 11M + 5S.  *degenerate is set when the x coordinates agree (H = 0).
template <int E> GL_HD Jac<E> jac_add(const Jac<E>& p, const Jac<E>& q, bool* degenerate) {
  Co<E> Z1Z1 = cmul(p.Z, p.Z), Z2Z2 = cmul(q.Z, q.Z);
  Co<E> U1 = cmul(p.X, Z2Z2), U2 = cmul(q.X, Z1Z1);
  Co<E> S1 = cmul(cmul(p.Y, q.Z), Z2Z2), S2 = cmul(cmul(q.Y, p.Z), Z1Z1);
  Co<E> H = csub(U2, U1);
  *degenerate = czero(H);
  Co<E> I = cadd(H, H); I = cmul(I, I);
  Co<E> J = cmul(H, I);
  Co<E> r = csub(S2, S1); r = cadd(r, r);
  Co<E> V = cmul(U1, I);
  Jac<E> o;
  o.X = csub(csub(cmul(r, r), J), cadd(V, V));
  Co<E> sj = cmul(S1, J);
  o.Y = csub(cmul(r, csub(V, o.X)), cadd(sj, sj));
  Co<E> zz = cadd(p.Z, q.Z); zz = cmul(zz, zz); zz = csub(csub(zz, Z1Z1), Z2Z2);
  o.Z = cmul(zz, H);
  return o;
}

// Jacobian chain storage, instance-major (one host thread or one lane writes a contiguous run): coordinate c (X,Y,Z)
// of step t (0..256) of instance k = E consecutive Fq (4 words each) at base + jac_at<E>(k, t, c).
template <int E> GL_HD size_t jac_at(size_t k, int t, int c) { return ((k * 257 + (size_t)t) * 3 + (size_t)c) * 4 * E; }
template <int E> GL_HD Co<E> ldc(const u64* p) { Co<E> r; for (int q = 0; q < E; q++) r.c[q] = ldq(p + 4 * q, 0, 1); return r; }
template <int E> GL_HD void stc(u64* p, const Co<E>& v) { for (int q = 0; q < E; q++) stq(p + 4 * q, 0, 1, v.c[q]); }
// io: x.x x.y offset.x offset.y (E x 8 u32 each) exp_val (8 u32).  A[t] = 2^t x (t = 0..256), B[0] = offset,
// B[t+1] = bit_t ? B[t] + A[t] : B[t]   (g1/exp.rs:165-230, g2/exp.rs:180-246: even rows add-if-bit, odd rows double),
// all in Jacobian coordinates and Montgomery form, so no step needs an inversion.  Returns TG_ERR_* flags.
template <int E> GL_HD int exp_chains(const uint32_t* io, size_t k, u64* ja, u64* jb) {
  u64 t4[4];
  Jac<E> a, b;
  Co<E>* dst[4] = {&a.X, &a.Y, &b.X, &b.Y};
  for (int v = 0; v < 4; v++) for (int q = 0; q < E; q++) { u32x8_to_u64x4(io + 8 * (v * E + q), t4); dst[v]->c[q] = to_m(t4); }
  a.Z = cone<E>(); b.Z = cone<E>();
  const uint32_t* e = io + 32 * E;
  int bad = 0;
  for (int t = 0;; t++) {
    stc<E>(ja + jac_at<E>(k, t, 0), a.X); stc<E>(ja + jac_at<E>(k, t, 1), a.Y); stc<E>(ja + jac_at<E>(k, t, 2), a.Z);
    stc<E>(jb + jac_at<E>(k, t, 0), b.X); stc<E>(jb + jac_at<E>(k, t, 1), b.Y); stc<E>(jb + jac_at<E>(k, t, 2), b.Z);
    if (t == 256) break;
    const bool bit = (e[t >> 5] >> (t & 31)) & 1;
    if (bit) { bool deg; b = jac_add<E>(b, a, &deg); if (deg) bad |= TG_ERR_DEGENERATE; }
    if (czero<E>(a.Y)) bad |= TG_ERR_DEGENERATE;
    a = jac_double<E>(a);
  }
  return bad;
}

}  // namespace bnw
