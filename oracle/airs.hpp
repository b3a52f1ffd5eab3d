// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the reference's AIR layer for the G1 path: limb-polynomial utilities, the
// bigint-mod-p gadgets, flags / pulses / range-check lookups, the G1 add/double gadget, and the two
// tables G1Stark (plumbing config, 2^9..2^10 rows) and G1ExpStark (2^16 rows).  Every function cites
// the reference file:line it follows.  Constraint EMISSION ORDER is part of the result (Horner in
// alpha) and is kept exactly, including the duplicated eval_flags call (g1/exp.rs:462,467).
#pragma once
#include "stark.hpp"
#include "bn254.hpp"
#include <array>

namespace orc {

static const int N_LIMBS = 16;         // src/constants.rs:1-2
static const int LIMB_BITS = 16;
static const int NUM_INPUT_LIMBS = 8;  // src/utils/flags.rs:41-42
static const int INPUT_LIMB_BITS = 32;
static const int NUM_FLAGS_COLS = 14;  // flags.rs:30
static const int64_t AUX_COEFF_ABS_MAX = 1 << 29;  // modular.rs:29

template <class P> static inline P cst(u64 v) { return P(GF(v)); }
template <class P, size_t N> using Arr = std::array<P, N>;

// ---- src/modular/pol_utils.rs -------------------------------------------------------------------
template <class T> static inline Arr<T, 31> pol_mul_wide(const Arr<T, 16>& a, const Arr<T, 16>& b) {  // :221
  Arr<T, 31> r; for (auto& x : r) x = T();
  for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) r[i + j] = r[i + j] + a[i] * b[j];
  return r;
}
template <class T> static inline Arr<T, 32> pol_mul_wide2(const Arr<T, 17>& a, const Arr<T, 16>& b) {  // :274
  Arr<T, 32> r; for (auto& x : r) x = T();
  for (int i = 0; i < 17; i++) for (int j = 0; j < 16; j++) r[i + j] = r[i + j] + a[i] * b[j];
  return r;
}
template <class T> static inline Arr<T, 31> pol_add(const Arr<T, 16>& a, const Arr<T, 16>& b) {  // :47
  Arr<T, 31> r; for (auto& x : r) x = T();
  for (int i = 0; i < 16; i++) r[i] = a[i] + b[i];
  return r;
}
template <class T> static inline Arr<T, 31> pol_sub(const Arr<T, 16>& a, const Arr<T, 16>& b) {  // :131
  Arr<T, 31> r; for (auto& x : r) x = T();
  for (int i = 0; i < 16; i++) r[i] = a[i] - b[i];
  return r;
}
template <class T, size_t N> static inline Arr<T, N> pol_sub_normal(const Arr<T, N>& a, const Arr<T, N>& b) {  // :142
  Arr<T, N> r; for (size_t i = 0; i < N; i++) r[i] = a[i] - b[i]; return r;
}
template <class T, size_t N> static inline Arr<T, N> pol_mul_scalar(const Arr<T, N>& a, T c) {  // :235
  Arr<T, N> r; for (size_t i = 0; i < N; i++) r[i] = c * a[i]; return r;
}
template <class T> static inline Arr<T, 31> widen(const Arr<T, 16>& a) {
  Arr<T, 31> r; for (auto& x : r) x = T(); for (int i = 0; i < 16; i++) r[i] = a[i]; return r;
}
// pol_adjoin_root :348  -- (x - root) * a(x)
template <class T> static inline Arr<T, 32> pol_adjoin_root(const Arr<T, 32>& a, T root) {
  Arr<T, 32> r;
  r[0] = (T() - root) * a[0];
  for (int d = 1; d < 32; d++) r[d] = a[d - 1] - root * a[d];
  return r;
}

static inline const Arr<int64_t, 16>& bn254_modulus_limbs() {  // modular.rs:298-309
  static const Arr<int64_t, 16> m = u256_to_limbs16(BN254_P);
  return m;
}
template <class P> static inline Arr<P, 16> bn254_modulus_p() {
  Arr<P, 16> r; const auto& m = bn254_modulus_limbs();
  for (int i = 0; i < 16; i++) r[i] = cst<P>((u64)m[i]);
  return r;
}

// ---- src/modular/modular.rs, modular_zero.rs, addcy.rs --------------------------------------------
template <class T> struct ModulusAux {      // modular.rs:31-36
  Arr<T, 16> out_aux_red; Arr<T, 17> quot_abs; Arr<T, 31> aux_input_lo, aux_input_hi;
};
template <class T> struct ModulusAuxZero {  // modular_zero.rs:27-31
  Arr<T, 17> quot_abs; Arr<T, 31> aux_input_lo, aux_input_hi;
};

// addcy.rs:16-58
template <class P>
static inline void eval_addcy(Consumer<P>& yc, P filter, const Arr<P, 16>& x, const Arr<P, 16>& y, const Arr<P, 16>& z, const Arr<P, 16>& given_cy) {
  P overflow = cst<P>(1ULL << LIMB_BITS);
  P overflow_inv = cst<P>(18446462594437939201ULL);  // addcy.rs:13
  P cy = P();
  for (int i = 0; i < 16; i++) {
    P t = cy + x[i] + y[i] - z[i];
    yc.constraint(filter * t * (overflow - t));
    cy = t * overflow_inv;
  }
  yc.constraint(filter * given_cy[0] * (given_cy[0] - cst<P>(1)));
  yc.constraint(filter * (cy - given_cy[0]));
  for (int i = 1; i < 16; i++) yc.constraint(filter * given_cy[i]);
}

template <class P>
static inline Arr<P, 32> aux_adjoined(const Arr<P, 31>& lo, const Arr<P, 31>& hi) {
  P base = cst<P>(1ULL << LIMB_BITS), offset = cst<P>((u64)AUX_COEFF_ABS_MAX);
  Arr<P, 32> aux_poly; aux_poly[31] = P();
  for (int i = 0; i < 31; i++) aux_poly[i] = (lo[i] - offset) + base * hi[i];
  return pol_adjoin_root(aux_poly, base);
}
// modular.rs:102-153
template <class P>
static inline Arr<P, 32> modular_constr_poly(Consumer<P>& yc, P filter, const Arr<P, 16>& modulus, const Arr<P, 16>& output, P quot_sign, const ModulusAux<P>& aux) {
  Arr<P, 16> is_less_than; for (auto& x : is_less_than) x = P(); is_less_than[0] = cst<P>(1);
  eval_addcy(yc, filter, modulus, aux.out_aux_red, output, is_less_than);
  yc.constraint(filter * (quot_sign * quot_sign - cst<P>(1)));
  Arr<P, 17> quot; for (int i = 0; i < 17; i++) quot[i] = quot_sign * aux.quot_abs[i];
  Arr<P, 32> constr = pol_mul_wide2(quot, modulus);
  for (int i = 0; i < 16; i++) constr[i] = constr[i] + output[i];
  Arr<P, 32> adj = aux_adjoined(aux.aux_input_lo, aux.aux_input_hi);
  for (int i = 0; i < 32; i++) constr[i] = constr[i] + adj[i];
  return constr;
}
// modular.rs:215-230
template <class P>
static inline void eval_modular_op(Consumer<P>& yc, P filter, const Arr<P, 16>& modulus, const Arr<P, 31>& input, const Arr<P, 16>& output, P quot_sign, const ModulusAux<P>& aux) {
  Arr<P, 32> c = modular_constr_poly(yc, filter, modulus, output, quot_sign, aux);
  for (int i = 0; i < 31; i++) c[i] = c[i] - input[i];
  for (int i = 0; i < 32; i++) yc.constraint(filter * c[i]);
}
// modular_zero.rs:82-120
template <class P>
static inline void eval_modular_zero(Consumer<P>& yc, P filter, const Arr<P, 16>& modulus, const Arr<P, 31>& input, P quot_sign, const ModulusAuxZero<P>& aux) {
  yc.constraint(filter * (quot_sign * quot_sign - cst<P>(1)));
  Arr<P, 17> quot; for (int i = 0; i < 17; i++) quot[i] = quot_sign * aux.quot_abs[i];
  Arr<P, 32> c = pol_mul_wide2(quot, modulus);
  Arr<P, 32> adj = aux_adjoined(aux.aux_input_lo, aux.aux_input_hi);
  for (int i = 0; i < 32; i++) c[i] = c[i] + adj[i];
  for (int i = 0; i < 31; i++) c[i] = c[i] - input[i];
  for (int i = 0; i < 32; i++) yc.constraint(filter * c[i]);
}

// Witness side.  Common core of generate_modular_op (modular.rs:38-100) and generate_modular_zero
// (modular_zero.rs:33-80): given P(x) (31 signed coeffs), out = P(beta) mod p, quot, aux.
struct ModWitness {
  Arr<int64_t, 16> output, out_aux_red;
  int quot_sign;  // +1 / -1
  Arr<int64_t, 17> quot_abs;
  Arr<int64_t, 31> aux_lo, aux_hi;
};
static inline ModWitness generate_modular_witness(const Arr<int64_t, 31>& pol_input, bool zero_mode) {
  const auto& ml = bn254_modulus_limbs();
  // input mod p by Horner in Fq
  Fq acc = fq_from_u64(0), b16 = fq_from_u64(65536);
  for (int i = 30; i >= 0; i--) {
    acc = fq_mul(acc, b16);
    int64_t c = pol_input[i];
    if (c >= 0) acc = fq_add(acc, fq_from_u64((uint64_t)c)); else acc = fq_sub(acc, fq_from_u64((uint64_t)(-c)));
  }
  U256 out = fq_to_u256(acc);
  if (zero_mode) assert(out.is_zero());  // modular_zero.rs:39
  // quot = (input - output) / p, exact: multiply by p^-1 mod 2^320
  I320 in = i320_zero();
  for (int i = 0; i < 31; i++) i320_add_shifted_i64(in, pol_input[i], 16 * i);
  I320 diff = i320_sub_u256(in, out);
  I320 q = i320_mul(diff, bn254_p_inv_320());
  ModWitness w;
  w.quot_sign = q.neg() ? -1 : 1;
  I320 qa = q.neg() ? i320_neg(q) : q;
  for (int i = 0; i < 17; i++) w.quot_abs[i] = (int64_t)((qa.l[i / 4] >> (16 * (i % 4))) & 0xffff);
  // |quot| must fit 17 limbs (bigint_to_columns assert, utils.rs:152)
  for (int i = 17; i < 20; i++) assert(((qa.l[i / 4] >> (16 * (i % 4))) & 0xffff) == 0);
  w.output = u256_to_limbs16(out);
  {
    // out_aux_red = 2^256 - p + output  (modular.rs:62)
    U256 t; u256_sub(t, out, BN254_P);  // wraps mod 2^256
    w.out_aux_red = u256_to_limbs16(t);
    assert(u256_cmp(out, BN254_P) < 0);
  }
  // constr_poly = input - output - quot*m ; aux = constr/(x - beta)   (modular.rs:65-72)
  int64_t constr[32];
  for (int i = 0; i < 31; i++) constr[i] = pol_input[i];
  constr[31] = 0;
  if (!zero_mode) for (int i = 0; i < 16; i++) constr[i] -= w.output[i];
  for (int i = 0; i < 17; i++) for (int j = 0; j < 16; j++) constr[i + j] -= (int64_t)w.quot_sign * w.quot_abs[i] * ml[j];
  int64_t aux[32];
  aux[0] = -(constr[0] >> LIMB_BITS);  // pol_remove_root_2exp, pol_utils.rs:390
  for (int d = 1; d < 31; d++) aux[d] = (aux[d - 1] - constr[d]) >> LIMB_BITS;
  aux[31] = 0;
  for (int i = 0; i < 32; i++) { aux[i] += AUX_COEFF_ABS_MAX; assert(aux[i] >= 0 && aux[i] <= 2 * AUX_COEFF_ABS_MAX); }
  for (int i = 0; i < 31; i++) { w.aux_lo[i] = aux[i] & 0xffff; w.aux_hi[i] = (aux[i] >> LIMB_BITS) & 0xffff; }
  return w;
}

// ---- src/curves/g1/muladd.rs -----------------------------------------------------------------------
template <class T> struct G1Output {  // :49-59 ; column layout :79-94 (320 columns)
  Arr<T, 16> lambda, new_x, new_y;
  ModulusAuxZero<T> aux_zero; ModulusAux<T> aux_x, aux_y;
  T quot_sign_zero, quot_sign_x, quot_sign_y;
};
static const int G1_OUTPUT_COLS = 20 * N_LIMBS;

template <class T> static inline Arr<T, 16> read16(const T* lv, int& cur) { Arr<T, 16> r; for (int i = 0; i < 16; i++) r[i] = lv[cur + i]; cur += 16; return r; }
template <class T, size_t N> static inline Arr<T, N> readn(const T* lv, int& cur) { Arr<T, N> r; for (size_t i = 0; i < N; i++) r[i] = lv[cur + i]; cur += (int)N; return r; }
template <class T> static inline G1Output<T> read_g1_output(const T* lv, int& cur) {  // :102-122
  G1Output<T> o;
  o.lambda = read16(lv, cur); o.new_x = read16(lv, cur); o.new_y = read16(lv, cur);
  o.aux_zero.quot_abs = readn<T, 17>(lv, cur); o.aux_zero.aux_input_lo = readn<T, 31>(lv, cur); o.aux_zero.aux_input_hi = readn<T, 31>(lv, cur);
  for (ModulusAux<T>* a : {&o.aux_x, &o.aux_y}) {
    a->out_aux_red = read16(lv, cur); a->quot_abs = readn<T, 17>(lv, cur); a->aux_input_lo = readn<T, 31>(lv, cur); a->aux_input_hi = readn<T, 31>(lv, cur);
  }
  o.quot_sign_zero = lv[cur++]; o.quot_sign_x = lv[cur++]; o.quot_sign_y = lv[cur++];
  return o;
}
// eval_g1_add :179-230
template <class P>
static inline void eval_g1_add(Consumer<P>& yc, P filter, const Arr<P, 16>& a_x, const Arr<P, 16>& a_y, const Arr<P, 16>& b_x, const Arr<P, 16>& b_y, const G1Output<P>& o) {
  Arr<P, 16> modulus = bn254_modulus_p<P>();
  Arr<P, 16> delta_x = pol_sub_normal(b_x, a_x);
  Arr<P, 31> delta_y = pol_sub(b_y, a_y);
  Arr<P, 31> zero_pol = pol_sub_normal(pol_mul_wide(o.lambda, delta_x), delta_y);
  eval_modular_zero(yc, filter, modulus, zero_pol, o.quot_sign_zero, o.aux_zero);
  Arr<P, 31> new_x_input = pol_sub_normal(pol_mul_wide(o.lambda, o.lambda), pol_add(a_x, b_x));
  eval_modular_op(yc, filter, modulus, new_x_input, o.new_x, o.quot_sign_x, o.aux_x);
  Arr<P, 16> x1_minus_new_x = pol_sub_normal(a_x, o.new_x);
  Arr<P, 31> new_y_input = pol_sub_normal(pol_mul_wide(o.lambda, x1_minus_new_x), widen(a_y));
  eval_modular_op(yc, filter, modulus, new_y_input, o.new_y, o.quot_sign_y, o.aux_y);
}
// eval_g1_double :291-342
template <class P>
static inline void eval_g1_double(Consumer<P>& yc, P filter, const Arr<P, 16>& x, const Arr<P, 16>& y, const G1Output<P>& o) {
  Arr<P, 16> modulus = bn254_modulus_p<P>();
  Arr<P, 31> lambda_y_double = pol_mul_scalar(pol_mul_wide(o.lambda, y), cst<P>(2));
  Arr<P, 31> x_sq_triple = pol_mul_scalar(pol_mul_wide(x, x), cst<P>(3));
  Arr<P, 31> zero_pol = pol_sub_normal(lambda_y_double, x_sq_triple);
  eval_modular_zero(yc, filter, modulus, zero_pol, o.quot_sign_zero, o.aux_zero);
  Arr<P, 31> new_x_input = pol_sub_normal(pol_mul_wide(o.lambda, o.lambda), pol_add(x, x));
  eval_modular_op(yc, filter, modulus, new_x_input, o.new_x, o.quot_sign_x, o.aux_x);
  Arr<P, 16> x1_minus_new_x = pol_sub_normal(x, o.new_x);
  Arr<P, 31> new_y_input = pol_sub_normal(pol_mul_wide(o.lambda, x1_minus_new_x), widen(y));
  eval_modular_op(yc, filter, modulus, new_y_input, o.new_y, o.quot_sign_y, o.aux_y);
}

// Witness for one add / double (generate_g1_add :124-177, generate_g1_double :409-460).
// Writes the 320 G1Output columns as canonical field values.
static inline void write_mod_aux(GF* lv, int& cur, const ModWitness& w, bool with_out_aux_red) {
  if (with_out_aux_red) for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)w.out_aux_red[i]);
  for (int i = 0; i < 17; i++) lv[cur++] = GF((u64)w.quot_abs[i]);
  for (int i = 0; i < 31; i++) lv[cur++] = GF((u64)w.aux_lo[i]);
  for (int i = 0; i < 31; i++) lv[cur++] = GF((u64)w.aux_hi[i]);
}
static inline void write_g1_output_default(GF* lv) {  // G1Output::default :61-75
  for (int i = 0; i < G1_OUTPUT_COLS - 3; i++) lv[i] = GF();
  for (int i = G1_OUTPUT_COLS - 3; i < G1_OUTPUT_COLS; i++) lv[i] = GF::one();
}
static inline Arr<int64_t, 31> i64_mul_wide(const Arr<int64_t, 16>& a, const Arr<int64_t, 16>& b) { return pol_mul_wide<int64_t>(a, b); }

// Returns (new_x, new_y); `is_double` selects the formula.  Writes lv[0..320).
static inline void generate_g1_op(bool is_double, const U256& ax, const U256& ay, const U256& bx, const U256& by, GF* lv, U256& nx, U256& ny) {
  Fq x1 = fq_from_u256(ax), y1 = fq_from_u256(ay);
  Fq lambda;
  if (is_double) {
    Fq x2 = fq_mul(x1, x1);
    Fq num = fq_add(fq_add(x2, x2), x2);
    Fq den = fq_add(y1, y1);
    assert(!den.m.is_zero());
    lambda = fq_mul(num, fq_inv(den));
  } else {
    Fq x2 = fq_from_u256(bx), y2 = fq_from_u256(by);
    Fq den = fq_sub(x2, x1);
    assert(!den.m.is_zero());
    lambda = fq_mul(fq_sub(y2, y1), fq_inv(den));
  }
  Arr<int64_t, 16> l = u256_to_limbs16(fq_to_u256(lambda));
  Arr<int64_t, 16> axl = u256_to_limbs16(ax), ayl = u256_to_limbs16(ay);
  Arr<int64_t, 16> bxl = is_double ? axl : u256_to_limbs16(bx), byl = is_double ? ayl : u256_to_limbs16(by);
  Arr<int64_t, 31> zero_pol;
  if (is_double) {
    zero_pol = pol_sub_normal(pol_mul_scalar(i64_mul_wide(l, ayl), (int64_t)2), pol_mul_scalar(i64_mul_wide(axl, axl), (int64_t)3));
  } else {
    zero_pol = pol_sub_normal(i64_mul_wide(l, pol_sub_normal(bxl, axl)), pol_sub(byl, ayl));
  }
  ModWitness wz = generate_modular_witness(zero_pol, true);
  Arr<int64_t, 31> new_x_input = pol_sub_normal(i64_mul_wide(l, l), pol_add(axl, bxl));
  ModWitness wx = generate_modular_witness(new_x_input, false);
  Arr<int64_t, 31> new_y_input = pol_sub_normal(i64_mul_wide(l, pol_sub_normal(axl, wx.output)), widen(ayl));
  ModWitness wy = generate_modular_witness(new_y_input, false);
  int cur = 0;
  for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)l[i]);
  for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)wx.output[i]);
  for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)wy.output[i]);
  write_mod_aux(lv, cur, wz, false);
  write_mod_aux(lv, cur, wx, true);
  write_mod_aux(lv, cur, wy, true);
  lv[cur++] = GF::from_i64(wz.quot_sign); lv[cur++] = GF::from_i64(wx.quot_sign); lv[cur++] = GF::from_i64(wy.quot_sign);
  assert(cur == G1_OUTPUT_COLS);
  nx = limbs16_to_u256(wx.output.data()); ny = limbs16_to_u256(wy.output.data());
}

// ---- src/utils/flags.rs -------------------------------------------------------------------------------
static inline void generate_flags_first_row(GF* lv, int s, const uint32_t* limbs) {  // :46-75
  uint32_t first_bit = limbs[0] & 1, rest = limbs[0] >> 1;
  lv[s] = GF(); lv[s + 1] = GF(); lv[s + 2] = GF(); lv[s + 3] = GF::one();
  lv[s + 4] = GF(first_bit); lv[s + 5] = GF(first_bit);
  lv[s + 6] = GF(rest);
  for (int i = 1; i < NUM_INPUT_LIMBS; i++) lv[s + 6 + i] = GF(limbs[i]);
}
static inline void generate_flags_next_row(const GF* lv, GF* nv, size_t cur_row, int s) {  // :77-134
  int a = s + 2, b = s + 3, fb = s + 4, bit = s + 5, sl = s + 6, el = sl + NUM_INPUT_LIMBS;
  nv[a] = GF::one() - lv[a];
  nv[b] = GF::one() - lv[b];
  size_t num_rows = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
  nv[s] = cur_row == num_rows - 2 ? GF::one() : GF();
  nv[s + 1] = (cur_row % (2 * INPUT_LIMB_BITS) == 2 * INPUT_LIMB_BITS - 3) ? GF::one() : GF();
  if (lv[a] == GF::one()) {
    u64 fl = lv[sl].v; u64 nb = fl & 1;
    nv[bit] = GF(nb); nv[sl] = GF(fl >> 1);
  } else { nv[bit] = lv[bit]; nv[sl] = lv[sl]; }
  if (lv[s + 1] == GF::one()) {
    for (int c = sl + 1; c < el; c++) nv[c - 1] = lv[c];
    nv[el - 1] = GF();
  } else {
    for (int c = sl + 1; c < el; c++) nv[c] = lv[c];
  }
  nv[fb] = nv[bit] * nv[b];
}
template <class P>
static inline void eval_flags(Consumer<P>& yc, const P* lv, const P* nv, int s) {  // :136-195
  int fin = s, rot = s + 1, a = s + 2, b = s + 3, fb = s + 4, bitc = s + 5, sl = s + 6, el = sl + NUM_INPUT_LIMBS;
  P one = cst<P>(1);
  yc.constraint_first_row(lv[a]);
  yc.constraint_first_row(lv[b] - one);
  P bit = lv[bitc];
  yc.constraint(bit * bit - bit);
  yc.constraint(bit * lv[b] - lv[fb]);
  yc.constraint(lv[rot] * lv[a]);
  yc.constraint(lv[fin] * lv[rot]);
  yc.constraint_transition(lv[a] + nv[a] - one);
  yc.constraint_transition(lv[b] + nv[b] - one);
  P first_limb = lv[sl], next_first_limb = nv[sl], next_bit = nv[bitc];
  P is_split = lv[a], is_final = lv[fin], is_not_final = one - is_final;
  yc.constraint_transition(is_not_final * is_split * (first_limb - cst<P>(2) * next_first_limb - next_bit));
  P is_not_split = one - is_split, is_rotate = lv[rot];
  P is_not_rotate_nor_final = one - is_rotate - is_final;
  yc.constraint_transition(is_not_split * (next_bit - bit));
  yc.constraint_transition(is_not_rotate_nor_final * is_not_split * (first_limb - next_first_limb));
  for (int c = sl + 1; c < el; c++) yc.constraint_transition(is_rotate * (nv[c - 1] - lv[c]));
  yc.constraint_transition(is_rotate * nv[el - 1]);
  for (int c = sl + 1; c < el; c++) yc.constraint_transition(is_not_rotate_nor_final * (nv[c] - lv[c]));
}

// ---- src/utils/pulse.rs ---------------------------------------------------------------------------------
static inline int get_witness_col(int start, int i) { return start + 1 + 2 * i; }  // :14
static inline int get_pulse_col(int start, int i) { return start + 1 + 2 * i + 1; }  // :10
static inline void generate_pulse(std::vector<std::vector<GF>>& cols, const std::vector<size_t>& positions) {  // :20-43
  size_t rows = cols[0].size();
  std::vector<GF> counter(rows);
  for (size_t i = 0; i < rows; i++) counter[i] = GF((u64)i);
  cols.push_back(counter);
  size_t base = cols.size();
  cols.resize(base + 2 * positions.size());
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t k = 0; k < positions.size(); k++) {
    size_t pos = positions[k];
    std::vector<GF> diff(rows), inv(rows);
    for (size_t i = 0; i < rows; i++) diff[i] = i == pos ? GF::one() : counter[i] - GF((u64)pos);
    gf_batch_inv(diff.data(), inv.data(), rows);
    inv[pos] = GF();
    std::vector<GF> pulse(rows, GF());
    pulse[pos] = GF::one();
    cols[base + 2 * k].swap(inv);
    cols[base + 2 * k + 1].swap(pulse);
  }
}
template <class P>
static inline void eval_pulse(Consumer<P>& yc, const P* lv, const P* nv, int start, const std::vector<size_t>& positions) {  // :45-63
  P counter = lv[start], one = cst<P>(1);
  yc.constraint_first_row(counter);
  yc.constraint_transition(nv[start] - counter - one);
  for (size_t i = 0; i < positions.size(); i++) {
    P cmp = counter - cst<P>((u64)positions[i]);
    P witness = lv[get_witness_col(start, (int)i)], pulse = lv[get_pulse_col(start, (int)i)];
    yc.constraint(cmp * witness + pulse - one);
    yc.constraint(cmp * pulse);
  }
}
static inline void generate_periodic_pulse_witness(std::vector<std::vector<GF>>& cols, size_t pulse_col, size_t period, size_t first_pulse) {  // :100-144
  size_t rows = cols[0].size();
  std::vector<GF> counter(rows), diff(rows), wit(rows);
  size_t c = period - first_pulse - 1;
  for (size_t i = 0; i < rows; i++) {
    counter[i] = GF((u64)c);
    assert((c == period - 1) == (cols[pulse_col][i] == GF::one()));
    diff[i] = c == period - 1 ? GF::one() : GF((u64)c) - GF((u64)(period - 1));
    c = (c + 1) % period;
  }
  gf_batch_inv(diff.data(), wit.data(), rows);
  for (size_t i = 0; i < rows; i++) if (counter[i] == GF((u64)(period - 1))) wit[i] = GF();
  cols.push_back(counter); cols.push_back(wit);
}
template <class P>
static inline void eval_periodic_pulse(Consumer<P>& yc, const P* lv, const P* nv, int pulse_col, int start, size_t period, size_t first_pulse) {  // :146-170
  P counter = lv[start], witness = lv[start + 1], is_reset = lv[pulse_col], next_counter = nv[start], one = cst<P>(1);
  yc.constraint_first_row(counter - cst<P>((u64)(period - first_pulse - 1)));
  yc.constraint_transition((one - is_reset) * (next_counter - counter - one));
  yc.constraint_transition(is_reset * next_counter);
  P delta = counter - cst<P>((u64)(period - 1));
  yc.constraint(delta * witness + is_reset - one);
  yc.constraint(delta * is_reset);
}

// ---- src/utils/lookup.rs, range_check.rs ----------------------------------------------------------------------
template <class P>
static inline void eval_lookups(Consumer<P>& yc, const P* lv, const P* nv, int col_in, int col_tab) {  // lookup.rs:13-34
  P diff_input_prev = nv[col_in] - lv[col_in];
  P diff_input_table = nv[col_in] - nv[col_tab];
  yc.constraint(diff_input_prev * diff_input_table);
  yc.constraint_last_row(diff_input_table);
}
static inline void permuted_cols(const std::vector<GF>& inputs, const std::vector<GF>& table, std::vector<GF>& sorted_inputs, std::vector<GF>& permuted_table) {  // lookup.rs:60-111
  size_t n = inputs.size();
  sorted_inputs = inputs;
  std::sort(sorted_inputs.begin(), sorted_inputs.end(), [](GF a, GF b) { return a.v < b.v; });
  std::vector<GF> sorted_table = table;
  std::sort(sorted_table.begin(), sorted_table.end(), [](GF a, GF b) { return a.v < b.v; });
  std::vector<size_t> unused_inds; std::vector<GF> unused_vals;
  permuted_table.assign(n, GF());
  size_t i = 0, j = 0;
  while (j < n && i < n) {
    u64 iv = sorted_inputs[i].v, tv = sorted_table[j].v;
    if (iv > tv) { unused_vals.push_back(sorted_table[j]); j++; }
    else if (iv < tv) {
      if (!unused_vals.empty()) { permuted_table[i] = unused_vals.back(); unused_vals.pop_back(); }
      else unused_inds.push_back(i);
      i++;
    } else { permuted_table[i] = sorted_table[j]; i++; j++; }
  }
  for (size_t k = j; k < n; k++) unused_vals.push_back(sorted_table[k]);
  for (size_t k = i; k < n; k++) unused_inds.push_back(k);
  assert(unused_inds.size() == unused_vals.size());
  for (size_t k = 0; k < unused_inds.size(); k++) permuted_table[unused_inds[k]] = unused_vals[k];
}
static inline std::vector<GF> range_table(size_t rows, u64 range_max) {
  std::vector<GF> t(rows);
  for (size_t i = 0; i < rows; i++) t[i] = GF(i < range_max ? (u64)i : range_max - 1);
  return t;
}
static inline void generate_u16_range_check(size_t t0, size_t t1, std::vector<std::vector<GF>>& cols) {  // range_check.rs:20-47
  size_t rows = cols[0].size();
  assert(rows >= 65536);  // :26
  std::vector<GF> table = range_table(rows, 65536);
  cols.push_back(table);
  size_t base = cols.size();
  cols.resize(base + 2 * (t1 - t0));
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t k = t0; k < t1; k++) {
    for (auto& x : cols[k]) { assert(x.v < 65536); (void)x; }
    permuted_cols(cols[k], table, cols[base + 2 * (k - t0)], cols[base + 2 * (k - t0) + 1]);
  }
}
template <class P>
static inline void eval_u16_range_check(Consumer<P>& yc, const P* lv, const P* nv, int start, size_t ntargets) {  // :49-68
  for (int i = start + 1; i < start + 1 + 2 * (int)ntargets; i += 2) eval_lookups(yc, lv, nv, i, i + 1);
  P cur = lv[start], next = nv[start];
  yc.constraint_first_row(cur);
  P incr = next - cur;
  yc.constraint_transition(incr * incr - incr);
  yc.constraint_last_row(cur - cst<P>(65535));
}
static inline void generate_split_u16_range_check(size_t t0, size_t t1, std::vector<std::vector<GF>>& cols) {  // :116-160
  size_t rows = cols[0].size();
  assert(rows >= 256);
  std::vector<GF> table = range_table(rows, 256);
  cols.push_back(table);
  size_t base = cols.size();
  cols.resize(base + 6 * (t1 - t0));
#pragma omp parallel for schedule(dynamic, 4)
  for (size_t k = t0; k < t1; k++) {
    std::vector<GF> lo(rows), hi(rows);
    for (size_t i = 0; i < rows; i++) { u64 v = cols[k][i].v; assert(v < 65536); lo[i] = GF(v & 0xff); hi[i] = GF(v >> 8); }
    size_t o = base + 6 * (k - t0);
    permuted_cols(lo, table, cols[o + 1], cols[o + 2]);
    permuted_cols(hi, table, cols[o + 4], cols[o + 5]);
    cols[o].swap(lo); cols[o + 3].swap(hi);
  }
}
template <class P>
static inline void eval_split_u16_range_check(Consumer<P>& yc, const P* lv, const P* nv, int main_col, size_t t0, size_t t1) {  // :162-192
  for (size_t i = 0; i < t1 - t0; i++) {
    P lo = lv[main_col + 1 + 6 * i], hi = lv[main_col + 4 + 6 * i];
    yc.constraint(lv[t0 + i] - (lo + hi * cst<P>(256)));
  }
  for (int i = main_col + 1; i < main_col + 1 + 6 * (int)(t1 - t0); i += 6) {
    eval_lookups(yc, lv, nv, i + 1, i + 2);
    eval_lookups(yc, lv, nv, i + 4, i + 5);
  }
  P cur = lv[main_col], next = nv[main_col];
  yc.constraint_first_row(cur);
  P incr = next - cur;
  yc.constraint_transition(incr * incr - incr);
  yc.constraint_last_row(cur - cst<P>(255));
}


// ---- src/fields/fq2.rs, src/curves/g2/muladd.rs -------------------------------------------------------------
template <class T> using Fq2Limbs = Arr<Arr<T, 16>, 2>;
template <class T> using Fq2Wide = Arr<Arr<T, 31>, 2>;
// pol_mul_fq2 fq2.rs:41-58
template <class T> static inline Fq2Wide<T> pol_mul_fq2(const Fq2Limbs<T>& x, const Fq2Limbs<T>& y) {
  Fq2Wide<T> z;
  z[0] = pol_sub_normal(pol_mul_wide(x[0], y[0]), pol_mul_wide(x[1], y[1]));
  Arr<T, 31> a = pol_mul_wide(x[0], y[1]), b = pol_mul_wide(x[1], y[0]);
  for (int i = 0; i < 31; i++) z[1][i] = a[i] + b[i];
  return z;
}
template <class T> static inline Fq2Wide<T> to_wide_fq2(const Fq2Limbs<T>& x) { Fq2Wide<T> z; z[0] = widen(x[0]); z[1] = widen(x[1]); return z; }
template <class T, size_t N> static inline Arr<Arr<T, N>, 2> pol_sub_fq2(const Arr<Arr<T, N>, 2>& x, const Arr<Arr<T, N>, 2>& y) {
  Arr<Arr<T, N>, 2> z; z[0] = pol_sub_normal(x[0], y[0]); z[1] = pol_sub_normal(x[1], y[1]); return z;
}
template <class T, size_t N> static inline Arr<Arr<T, N>, 2> pol_add_fq2(const Arr<Arr<T, N>, 2>& x, const Arr<Arr<T, N>, 2>& y) {
  Arr<Arr<T, N>, 2> z; for (int c = 0; c < 2; c++) for (size_t i = 0; i < N; i++) z[c][i] = x[c][i] + y[c][i]; return z;
}
template <class T, size_t N> static inline Arr<Arr<T, N>, 2> pol_mul_scalar_fq2(const Arr<Arr<T, N>, 2>& x, T c) {
  Arr<Arr<T, N>, 2> z; z[0] = pol_mul_scalar(x[0], c); z[1] = pol_mul_scalar(x[1], c); return z;
}
template <class T> struct G2Output {  // muladd.rs:32-40 ; 640 columns (:56-80)
  Fq2Limbs<T> lambda, new_x, new_y;
  ModulusAuxZero<T> aux_zeros[2]; ModulusAux<T> auxs[4];
  T quot_sign_zeros[2], quot_signs[4];
};
static const int G2_OUTPUT_COLS = 40 * N_LIMBS;
template <class T> static inline Fq2Limbs<T> read_fq2(const T* lv, int& cur) { Fq2Limbs<T> r; r[0] = read16(lv, cur); r[1] = read16(lv, cur); return r; }
template <class T> static inline G2Output<T> read_g2_output(const T* lv, int& cur) {  // :82-116
  G2Output<T> o;
  o.lambda = read_fq2(lv, cur); o.new_x = read_fq2(lv, cur); o.new_y = read_fq2(lv, cur);
  for (int i = 0; i < 2; i++) { o.aux_zeros[i].quot_abs = readn<T, 17>(lv, cur); o.aux_zeros[i].aux_input_lo = readn<T, 31>(lv, cur); o.aux_zeros[i].aux_input_hi = readn<T, 31>(lv, cur); }
  for (int i = 0; i < 4; i++) { o.auxs[i].out_aux_red = read16(lv, cur); o.auxs[i].quot_abs = readn<T, 17>(lv, cur); o.auxs[i].aux_input_lo = readn<T, 31>(lv, cur); o.auxs[i].aux_input_hi = readn<T, 31>(lv, cur); }
  for (int i = 0; i < 2; i++) o.quot_sign_zeros[i] = lv[cur++];
  for (int i = 0; i < 4; i++) o.quot_signs[i] = lv[cur++];
  return o;
}
template <class P>
static inline void eval_g2_tail(Consumer<P>& yc, P filter, const Fq2Wide<P>& zero_pol, const Fq2Wide<P>& new_x_input, const Fq2Limbs<P>& x1, const Fq2Limbs<P>& y1, const G2Output<P>& o) {
  Arr<P, 16> modulus = bn254_modulus_p<P>();
  for (int i = 0; i < 2; i++) eval_modular_zero(yc, filter, modulus, zero_pol[i], o.quot_sign_zeros[i], o.aux_zeros[i]);
  for (int i = 0; i < 2; i++) eval_modular_op(yc, filter, modulus, new_x_input[i], o.new_x[i], o.quot_signs[i], o.auxs[i]);
  Fq2Wide<P> new_y_input = pol_sub_fq2(pol_mul_fq2(o.lambda, pol_sub_fq2(x1, o.new_x)), to_wide_fq2(y1));
  for (int i = 0; i < 2; i++) eval_modular_op(yc, filter, modulus, new_y_input[i], o.new_y[i], o.quot_signs[i + 2], o.auxs[i + 2]);
}
// eval_g2_add :416-472
template <class P>
static inline void eval_g2_add(Consumer<P>& yc, P filter, const Fq2Limbs<P>& a_x, const Fq2Limbs<P>& a_y, const Fq2Limbs<P>& b_x, const Fq2Limbs<P>& b_y, const G2Output<P>& o) {
  Fq2Wide<P> zero_pol = pol_sub_fq2(pol_mul_fq2(o.lambda, pol_sub_fq2(b_x, a_x)), to_wide_fq2(pol_sub_fq2(b_y, a_y)));
  Fq2Wide<P> new_x_input = pol_sub_fq2(pol_mul_fq2(o.lambda, o.lambda), to_wide_fq2(pol_add_fq2(a_x, b_x)));
  eval_g2_tail(yc, filter, zero_pol, new_x_input, a_x, a_y, o);
}
// eval_g2_double :203-261
template <class P>
static inline void eval_g2_double(Consumer<P>& yc, P filter, const Fq2Limbs<P>& x, const Fq2Limbs<P>& y, const G2Output<P>& o) {
  Fq2Wide<P> zero_pol = pol_sub_fq2(pol_mul_scalar_fq2(pol_mul_fq2(o.lambda, y), cst<P>(2)), pol_mul_scalar_fq2(pol_mul_fq2(x, x), cst<P>(3)));
  Fq2Wide<P> new_x_input = pol_sub_fq2(pol_mul_fq2(o.lambda, o.lambda), to_wide_fq2(pol_mul_scalar_fq2(x, cst<P>(2))));
  eval_g2_tail(yc, filter, zero_pol, new_x_input, x, y, o);
}

// Fq2 = Fq[i]/(i^2+1) witness arithmetic (arkworks Fq2 in the reference, muladd.rs:118-126, :330-341)
struct Fq2 { Fq c0, c1; };
static inline Fq2 fq2_add(const Fq2& a, const Fq2& b) { return {fq_add(a.c0, b.c0), fq_add(a.c1, b.c1)}; }
static inline Fq2 fq2_sub(const Fq2& a, const Fq2& b) { return {fq_sub(a.c0, b.c0), fq_sub(a.c1, b.c1)}; }
static inline Fq2 fq2_mul(const Fq2& a, const Fq2& b) {
  return {fq_sub(fq_mul(a.c0, b.c0), fq_mul(a.c1, b.c1)), fq_add(fq_mul(a.c0, b.c1), fq_mul(a.c1, b.c0))};
}
static inline Fq2 fq2_inv(const Fq2& a) {
  Fq n = fq_inv(fq_add(fq_mul(a.c0, a.c0), fq_mul(a.c1, a.c1)));
  return {fq_mul(a.c0, n), fq_sub(fq_from_u64(0), fq_mul(a.c1, n))};
}
struct U256x2 { U256 c0, c1; };
struct G2Affine { U256x2 x, y; };
static inline Fq2 fq2_from(const U256x2& v) { return {fq_from_u256(v.c0), fq_from_u256(v.c1)}; }

// generate_g2_add (:330-414) / generate_g2_double (:118-201): writes lv[0..640), returns new_x, new_y.
static inline void generate_g2_op(bool is_double, const U256x2& ax, const U256x2& ay, const U256x2& bx, const U256x2& by, GF* lv, U256x2& nx, U256x2& ny) {
  Fq2 x1 = fq2_from(ax), y1 = fq2_from(ay), lambda;
  if (is_double) {
    Fq2 x2 = fq2_mul(x1, x1);
    Fq2 num = fq2_add(fq2_add(x2, x2), x2), den = fq2_add(y1, y1);
    assert(!(den.c0.m.is_zero() && den.c1.m.is_zero()));
    lambda = fq2_mul(num, fq2_inv(den));
  } else {
    Fq2 x2 = fq2_from(bx), y2 = fq2_from(by), den = fq2_sub(x2, x1);
    assert(!(den.c0.m.is_zero() && den.c1.m.is_zero()));
    lambda = fq2_mul(fq2_sub(y2, y1), fq2_inv(den));
  }
  auto limbs2 = [](const U256x2& v) { Fq2Limbs<int64_t> r; r[0] = u256_to_limbs16(v.c0); r[1] = u256_to_limbs16(v.c1); return r; };
  Fq2Limbs<int64_t> l; l[0] = u256_to_limbs16(fq_to_u256(lambda.c0)); l[1] = u256_to_limbs16(fq_to_u256(lambda.c1));
  Fq2Limbs<int64_t> axl = limbs2(ax), ayl = limbs2(ay), bxl = is_double ? axl : limbs2(bx), byl = is_double ? ayl : limbs2(by);
  Fq2Wide<int64_t> zero_pol;
  if (is_double) zero_pol = pol_sub_fq2(pol_mul_scalar_fq2(pol_mul_fq2(l, ayl), (int64_t)2), pol_mul_scalar_fq2(pol_mul_fq2(axl, axl), (int64_t)3));
  else zero_pol = pol_sub_fq2(pol_mul_fq2(l, pol_sub_fq2(bxl, axl)), to_wide_fq2(pol_sub_fq2(byl, ayl)));
  ModWitness wz[2], wx[2], wy[2];
  for (int i = 0; i < 2; i++) wz[i] = generate_modular_witness(zero_pol[i], true);
  Fq2Wide<int64_t> new_x_input = pol_sub_fq2(pol_mul_fq2(l, l), to_wide_fq2(pol_add_fq2(axl, bxl)));
  for (int i = 0; i < 2; i++) wx[i] = generate_modular_witness(new_x_input[i], false);
  Fq2Limbs<int64_t> nxl; nxl[0] = wx[0].output; nxl[1] = wx[1].output;
  Fq2Wide<int64_t> new_y_input = pol_sub_fq2(pol_mul_fq2(l, pol_sub_fq2(axl, nxl)), to_wide_fq2(ayl));
  for (int i = 0; i < 2; i++) wy[i] = generate_modular_witness(new_y_input[i], false);
  int cur = 0;
  for (int c = 0; c < 2; c++) for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)l[c][i]);
  for (int c = 0; c < 2; c++) for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)wx[c].output[i]);
  for (int c = 0; c < 2; c++) for (int i = 0; i < 16; i++) lv[cur++] = GF((u64)wy[c].output[i]);
  for (int i = 0; i < 2; i++) write_mod_aux(lv, cur, wz[i], false);
  for (int i = 0; i < 2; i++) write_mod_aux(lv, cur, wx[i], true);
  for (int i = 0; i < 2; i++) write_mod_aux(lv, cur, wy[i], true);
  for (int i = 0; i < 2; i++) lv[cur++] = GF::from_i64(wz[i].quot_sign);
  for (int i = 0; i < 2; i++) lv[cur++] = GF::from_i64(wx[i].quot_sign);
  for (int i = 0; i < 2; i++) lv[cur++] = GF::from_i64(wy[i].quot_sign);
  assert(cur == G2_OUTPUT_COLS);
  nx.c0 = limbs16_to_u256(wx[0].output.data()); nx.c1 = limbs16_to_u256(wx[1].output.data());
  ny.c0 = limbs16_to_u256(wy[0].output.data()); ny.c1 = limbs16_to_u256(wy[1].output.data());
}
static inline void write_g2_output_default(GF* lv) {  // G2Output::default :42-54
  for (int i = 0; i < G2_OUTPUT_COLS - 6; i++) lv[i] = GF();
  for (int i = G2_OUTPUT_COLS - 6; i < G2_OUTPUT_COLS; i++) lv[i] = GF::one();
}

// ---- tables ---------------------------------------------------------------------------------------------------------
template <class Derived>
struct AirBase : Air {
  void eval(const GF* lv, const GF* nv, const GF* pi, Consumer<GF>& c) const override { static_cast<const Derived*>(this)->template eval_t<GF>(lv, nv, pi, c); }
  void eval(const Ext* lv, const Ext* nv, const Ext* pi, Consumer<Ext>& c) const override { static_cast<const Derived*>(this)->template eval_t<Ext>(lv, nv, pi, c); }
};

// G1Stark: src/curves/g1/muladd.rs:462-624 (single add rows, split range check).
struct G1OpAir : AirBase<G1OpAir> {
  static constexpr int MAIN_COLS = 24 * N_LIMBS + 2;
  static constexpr int START_RC = 4 * N_LIMBS, NUM_RC = 20 * N_LIMBS - 4, END_RC = START_RC + NUM_RC;
  size_t num_columns() const override { return MAIN_COLS + 1 + 6 * NUM_RC; }
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // range_check.rs:228-246
    std::vector<std::pair<size_t, size_t>> p;
    for (size_t i = MAIN_COLS + 1; i < (size_t)MAIN_COLS + 1 + 6 * NUM_RC; i += 6) {
      p.push_back({MAIN_COLS, i + 2}); p.push_back({MAIN_COLS, i + 5});
      p.push_back({i, i + 1}); p.push_back({i + 3, i + 4});
    }
    return p;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const {  // :550-582
    eval_split_u16_range_check(yc, lv, nv, MAIN_COLS, START_RC, END_RC);
    int cur = 0;
    auto a_x = read16(lv, cur), a_y = read16(lv, cur), b_x = read16(lv, cur), b_y = read16(lv, cur);
    G1Output<P> o = read_g1_output(lv, cur);
    P is_add = lv[cur++], is_double = lv[cur++];
    eval_g1_add(yc, is_add, a_x, a_y, b_x, b_y, o);
    eval_g1_double(yc, is_double, a_x, a_y, o);
  }
  // generate_trace :481-546 with caller-provided point pairs (the reference draws them at random).
  std::vector<std::vector<GF>> generate_trace(const std::vector<G1Affine>& a, const std::vector<G1Affine>& b) const {
    size_t rows = a.size();
    std::vector<std::vector<GF>> cols(MAIN_COLS, std::vector<GF>(rows));
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) {
      GF lv[MAIN_COLS];
      auto put = [&](int off, const U256& v) { auto l = u256_to_limbs16(v); for (int i = 0; i < 16; i++) lv[off + i] = GF((u64)l[i]); };
      put(0, a[r].x); put(16, a[r].y); put(32, b[r].x); put(48, b[r].y);
      U256 nx, ny;
      generate_g1_op(false, a[r].x, a[r].y, b[r].x, b[r].y, lv + 64, nx, ny);
      lv[MAIN_COLS - 2] = GF::one(); lv[MAIN_COLS - 1] = GF();
      for (int c = 0; c < MAIN_COLS; c++) cols[c][r] = lv[c];
    }
    generate_split_u16_range_check(START_RC, END_RC, cols);
    return cols;
  }
};

// G1ExpStark: src/curves/g1/exp.rs.
struct G1ExpIONative { G1Affine x, offset; uint32_t exp_val[NUM_INPUT_LIMBS]; G1Affine output; };  // :109-114
struct G1ExpAir : AirBase<G1ExpAir> {
  size_t num_io;
  // constants(num_io) :6-34
  int start_flags_col, num_main_cols, start_periodic_pulse_col, start_io_pulses_col, start_lookups_col, num_range_check_cols;
  size_t ncols, npi;
  std::vector<size_t> pulse_positions;  // get_pulse_positions :153-163
  explicit G1ExpAir(size_t n) : num_io(n) {
    start_flags_col = 24 * N_LIMBS;
    num_main_cols = start_flags_col + NUM_FLAGS_COLS;
    start_periodic_pulse_col = num_main_cols;
    start_io_pulses_col = start_periodic_pulse_col + 2;
    start_lookups_col = start_io_pulses_col + 1 + 4 * (int)num_io;
    num_range_check_cols = 24 * N_LIMBS - 3;
    ncols = start_lookups_col + 1 + 2 * num_range_check_cols;
    npi = 7 * NUM_INPUT_LIMBS * num_io;
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    for (size_t i = 0; i < num_io; i++) { pulse_positions.push_back(i * rpb); pulse_positions.push_back(i * rpb + rpb - 1); }
  }
  size_t num_columns() const override { return ncols; }
  size_t num_public_inputs() const override { return npi; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // range_check.rs:96-113
    std::vector<std::pair<size_t, size_t>> p;
    for (int i = 0; i < num_range_check_cols; i++) {
      p.push_back({(size_t)start_lookups_col, (size_t)start_lookups_col + 1 + 2 * i + 1});
      p.push_back({(size_t)i, (size_t)start_lookups_col + 1 + 2 * i});
    }
    return p;
  }
  template <class P> static Arr<P, 8> u16_to_u32(const Arr<P, 16>& x) {  // utils.rs:56-64
    Arr<P, 8> r; P base = cst<P>(1ULL << 16);
    for (int i = 0; i < 8; i++) r[i] = x[2 * i] + base * x[2 * i + 1];
    return r;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P* pi, Consumer<P>& yc) const {  // :331-495
    int is_final_col = start_flags_col, is_double_col = start_flags_col + 2, is_add_col = start_flags_col + 4, start_limbs_col = start_flags_col + 6;
    P one = cst<P>(1);
    int cur = 0;
    auto a_x = read16(lv, cur), a_y = read16(lv, cur), b_x = read16(lv, cur), b_y = read16(lv, cur);
    G1Output<P> output = read_g1_output(lv, cur);
    P is_add = lv[is_add_col], is_double = lv[is_double_col], is_final = lv[is_final_col];
    P is_not_final = one - is_final;
    // is_final == sum of output pulses  :359-365
    P sum_is_output = P();
    for (size_t i = 1; i < 2 * num_io; i += 2) sum_is_output = sum_is_output + lv[get_pulse_col(start_io_pulses_col, (int)i)];
    yc.constraint(is_final - sum_is_output);
    // public inputs :368-392
    {
      Arr<P, 8> x_x = u16_to_u32(a_x), x_y = u16_to_u32(a_y), bx32 = u16_to_u32(b_x), by32 = u16_to_u32(b_y);
      Arr<P, 8> limbs; for (int k = 0; k < 8; k++) limbs[k] = lv[start_limbs_col + k];
      limbs[0] = limbs[0] * cst<P>(2) + is_add;
      size_t pc = 0;
      for (size_t i = 0; i < 2 * num_io; i += 2) {
        const P* io = pi + pc; pc += 56;  // x.x x.y off.x off.y exp_val out.x out.y  (read_g1_exp_io :137-151)
        P is_in = lv[get_pulse_col(start_io_pulses_col, (int)i)], is_out = lv[get_pulse_col(start_io_pulses_col, (int)i + 1)];
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[k] - x_x[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[8 + k] - x_y[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[16 + k] - bx32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[24 + k] - by32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_out * (io[40 + k] - bx32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_out * (io[48 + k] - by32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[32 + k] - limbs[k]));
      }
    }
    // transitions :395-461
    cur = 0;
    auto n_a_x = read16(nv, cur), n_a_y = read16(nv, cur), n_b_x = read16(nv, cur), n_b_y = read16(nv, cur);
    auto eqt = [&](P filter, const Arr<P, 16>& x, const Arr<P, 16>& y) { for (int i = 0; i < 16; i++) yc.constraint_transition(filter * (x[i] - y[i])); };
    P fd = is_not_final * is_double;
    eqt(fd, n_a_x, output.new_x); eqt(fd, n_a_y, output.new_y); eqt(fd, n_b_x, b_x); eqt(fd, n_b_y, b_y);
    P fa = is_not_final * is_add;
    eqt(fa, n_a_x, a_x); eqt(fa, n_a_y, a_y); eqt(fa, n_b_x, output.new_x); eqt(fa, n_b_y, output.new_y);
    P fn = is_not_final * (one - is_double - is_add);
    eqt(fn, n_a_x, a_x); eqt(fn, n_a_y, a_y); eqt(fn, n_b_x, b_x); eqt(fn, n_b_y, b_y);
    eval_flags(yc, lv, nv, start_flags_col);                       // :462
    eval_g1_add(yc, is_add, a_x, a_y, b_x, b_y, output);           // :463
    eval_g1_double(yc, is_double, a_x, a_y, output);               // :464
    eval_flags(yc, lv, nv, start_flags_col);                       // :467 (duplicate, kept)
    eval_periodic_pulse(yc, lv, nv, start_flags_col + 1, start_periodic_pulse_col, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    eval_pulse(yc, lv, nv, start_io_pulses_col, pulse_positions);
    eval_u16_range_check(yc, lv, nv, start_lookups_col, (size_t)num_range_check_cols);
  }
  // generate_trace_for_one_block :255-288 -> writes 512 rows of num_main_cols into `rows`
  void generate_block(const G1ExpIONative& in, std::vector<std::vector<GF>>& cols, size_t row0, G1Affine& out) const {
    size_t num_rows = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    std::vector<GF> lv(num_main_cols, GF()), nvv(num_main_cols, GF());
    int sf = start_flags_col;
    auto put = [&](GF* r, int off, const U256& v) { auto l = u256_to_limbs16(v); for (int i = 0; i < 16; i++) r[off + i] = GF((u64)l[i]); };
    generate_flags_first_row(lv.data(), sf, in.exp_val);
    U256 ax = in.x.x, ay = in.x.y, bx = in.offset.x, by = in.offset.y, nx, ny;
    // generate_g1_exp_first_row :165-191
    put(lv.data(), 0, ax); put(lv.data(), 16, ay); put(lv.data(), 32, bx); put(lv.data(), 48, by);
    if (lv[sf + 4] == GF::one()) generate_g1_op(false, ax, ay, bx, by, lv.data() + 64, nx, ny); else write_g1_output_default(lv.data() + 64);
    for (int c = 0; c < num_main_cols; c++) cols[c][row0] = lv[c];
    for (size_t i = 0; i + 1 < num_rows; i++) {
      std::fill(nvv.begin(), nvv.end(), GF());
      generate_flags_next_row(lv.data(), nvv.data(), i, sf);
      // generate_g1_exp_next_row :193-230
      if (lv[sf + 2] == GF::one()) { ax = nx; ay = ny; }
      else if (lv[sf + 4] == GF::one()) { bx = nx; by = ny; }
      put(nvv.data(), 0, ax); put(nvv.data(), 16, ay); put(nvv.data(), 32, bx); put(nvv.data(), 48, by);
      if (nvv[sf + 2] == GF::one()) generate_g1_op(true, ax, ay, ax, ay, nvv.data() + 64, nx, ny);
      else if (nvv[sf + 4] == GF::one()) generate_g1_op(false, ax, ay, bx, by, nvv.data() + 64, nx, ny);
      else write_g1_output_default(nvv.data() + 64);
      for (int c = 0; c < num_main_cols; c++) cols[c][row0 + i + 1] = nvv[c];
      lv.swap(nvv);
    }
    out.x = bx; out.y = by;  // b at the last row (:273-281)
  }
  // generate_trace :290-318 ; fills `outputs` with x*s+offset per instance.
  std::vector<std::vector<GF>> generate_trace(std::vector<G1ExpIONative>& inputs) const {
    assert(inputs.size() == num_io);
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS, rows = rpb * num_io;
    std::vector<std::vector<GF>> cols(num_main_cols, std::vector<GF>(rows));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < num_io; k++) generate_block(inputs[k], cols, k * rpb, inputs[k].output);
    generate_periodic_pulse_witness(cols, start_flags_col + 1, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    generate_pulse(cols, pulse_positions);
    generate_u16_range_check(0, (size_t)num_range_check_cols, cols);
    assert(cols.size() == ncols);
    return cols;
  }
  // generate_public_inputs :320-327 ; g1_exp_io_to_columns :124-135 ; fq_to_u32_columns utils.rs:24
  std::vector<GF> generate_public_inputs(const std::vector<G1ExpIONative>& inputs) const {
    std::vector<GF> pi;
    auto put = [&](const U256& v) { for (int i = 0; i < 8; i++) pi.push_back(GF((v.l[i / 2] >> (32 * (i % 2))) & 0xffffffffULL)); };
    for (auto& in : inputs) {
      put(in.x.x); put(in.x.y); put(in.offset.x); put(in.offset.y);
      for (int i = 0; i < 8; i++) pi.push_back(GF(in.exp_val[i]));
      put(in.output.x); put(in.output.y);
    }
    return pi;
  }
};

// G2ExpStark: src/curves/g2/exp.rs (same machine as G1ExpStark over Fq2 coordinates).
struct G2ExpIONative { G2Affine x, offset; uint32_t exp_val[NUM_INPUT_LIMBS]; G2Affine output; };  // :90-95
struct G2ExpAir : AirBase<G2ExpAir> {
  size_t num_io;
  int start_flags_col, num_main_cols, start_periodic_pulse_col, start_io_pulses_col, start_lookups_col, num_range_check_cols;  // constants :6-34
  size_t ncols, npi;
  std::vector<size_t> pulse_positions;
  explicit G2ExpAir(size_t n) : num_io(n) {
    start_flags_col = 48 * N_LIMBS;
    num_main_cols = start_flags_col + NUM_FLAGS_COLS;
    start_periodic_pulse_col = num_main_cols;
    start_io_pulses_col = start_periodic_pulse_col + 2;
    start_lookups_col = start_io_pulses_col + 1 + 4 * (int)num_io;
    num_range_check_cols = 48 * N_LIMBS - 6;
    ncols = start_lookups_col + 1 + 2 * num_range_check_cols;
    npi = 13 * NUM_INPUT_LIMBS * num_io;
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    for (size_t i = 0; i < num_io; i++) { pulse_positions.push_back(i * rpb); pulse_positions.push_back(i * rpb + rpb - 1); }
  }
  size_t num_columns() const override { return ncols; }
  size_t num_public_inputs() const override { return npi; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // :800-806
    std::vector<std::pair<size_t, size_t>> p;
    for (int i = 0; i < num_range_check_cols; i++) {
      p.push_back({(size_t)start_lookups_col, (size_t)start_lookups_col + 1 + 2 * i + 1});
      p.push_back({(size_t)i, (size_t)start_lookups_col + 1 + 2 * i});
    }
    return p;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P* pi, Consumer<P>& yc) const {  // :345-507
    int is_final_col = start_flags_col, is_double_col = start_flags_col + 2, is_add_col = start_flags_col + 4, start_limbs_col = start_flags_col + 6;
    P one = cst<P>(1);
    int cur = 0;
    Fq2Limbs<P> a_x = read_fq2(lv, cur), a_y = read_fq2(lv, cur), b_x = read_fq2(lv, cur), b_y = read_fq2(lv, cur);
    G2Output<P> output = read_g2_output(lv, cur);
    P is_add = lv[is_add_col], is_double = lv[is_double_col], is_final = lv[is_final_col];
    P is_not_final = one - is_final;
    P sum_is_output = P();
    for (size_t i = 1; i < 2 * num_io; i += 2) sum_is_output = sum_is_output + lv[get_pulse_col(start_io_pulses_col, (int)i)];
    yc.constraint(is_final - sum_is_output);  // :374-379
    {  // public inputs :382-414
      Arr<P, 8> a32[4] = {G1ExpAir::u16_to_u32(a_x[0]), G1ExpAir::u16_to_u32(a_x[1]), G1ExpAir::u16_to_u32(a_y[0]), G1ExpAir::u16_to_u32(a_y[1])};
      Arr<P, 8> b32[4] = {G1ExpAir::u16_to_u32(b_x[0]), G1ExpAir::u16_to_u32(b_x[1]), G1ExpAir::u16_to_u32(b_y[0]), G1ExpAir::u16_to_u32(b_y[1])};
      Arr<P, 8> limbs; for (int k = 0; k < 8; k++) limbs[k] = lv[start_limbs_col + k];
      limbs[0] = limbs[0] * cst<P>(2) + is_add;
      size_t pc = 0;
      for (size_t i = 0; i < 2 * num_io; i += 2) {
        const P* io = pi + pc; pc += 104;  // x[4] offset[4] exp_val output[4]   (read_g2_exp_io :158-178)
        P is_in = lv[get_pulse_col(start_io_pulses_col, (int)i)], is_out = lv[get_pulse_col(start_io_pulses_col, (int)i + 1)];
        for (int q = 0; q < 4; q++) for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[8 * q + k] - a32[q][k]));
        for (int q = 0; q < 4; q++) for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[32 + 8 * q + k] - b32[q][k]));
        for (int q = 0; q < 4; q++) for (int k = 0; k < 8; k++) yc.constraint(is_out * (io[72 + 8 * q + k] - b32[q][k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[64 + k] - limbs[k]));
      }
    }
    cur = 0;
    Fq2Limbs<P> n_a_x = read_fq2(nv, cur), n_a_y = read_fq2(nv, cur), n_b_x = read_fq2(nv, cur), n_b_y = read_fq2(nv, cur);
    auto eqt = [&](P filter, const Fq2Limbs<P>& x, const Fq2Limbs<P>& y) { for (int c = 0; c < 2; c++) for (int i = 0; i < 16; i++) yc.constraint_transition(filter * (x[c][i] - y[c][i])); };
    P fd = is_not_final * is_double;
    eqt(fd, n_a_x, output.new_x); eqt(fd, n_a_y, output.new_y); eqt(fd, n_b_x, b_x); eqt(fd, n_b_y, b_y);
    P fa = is_not_final * is_add;
    eqt(fa, n_a_x, a_x); eqt(fa, n_a_y, a_y); eqt(fa, n_b_x, output.new_x); eqt(fa, n_b_y, output.new_y);
    P fn = is_not_final * (one - is_double - is_add);
    eqt(fn, n_a_x, a_x); eqt(fn, n_a_y, a_y); eqt(fn, n_b_x, b_x); eqt(fn, n_b_y, b_y);
    eval_flags(yc, lv, nv, start_flags_col);                      // :474
    eval_g2_add(yc, is_add, a_x, a_y, b_x, b_y, output);          // :475
    eval_g2_double(yc, is_double, a_x, a_y, output);              // :476
    eval_flags(yc, lv, nv, start_flags_col);                      // :479 (duplicate, kept)
    eval_periodic_pulse(yc, lv, nv, start_flags_col + 1, start_periodic_pulse_col, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    eval_pulse(yc, lv, nv, start_io_pulses_col, pulse_positions);
    eval_u16_range_check(yc, lv, nv, start_lookups_col, (size_t)num_range_check_cols);
  }
  void generate_block(const G2ExpIONative& in, std::vector<std::vector<GF>>& cols, size_t row0, G2Affine& out) const {  // :271-305
    size_t num_rows = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    std::vector<GF> lv(num_main_cols, GF()), nvv(num_main_cols, GF());
    int sf = start_flags_col;
    auto put2 = [&](GF* r, int off, const U256x2& v) { auto l0 = u256_to_limbs16(v.c0), l1 = u256_to_limbs16(v.c1); for (int i = 0; i < 16; i++) { r[off + i] = GF((u64)l0[i]); r[off + 16 + i] = GF((u64)l1[i]); } };
    generate_flags_first_row(lv.data(), sf, in.exp_val);
    U256x2 ax = in.x.x, ay = in.x.y, bx = in.offset.x, by = in.offset.y, nx, ny;
    put2(lv.data(), 0, ax); put2(lv.data(), 32, ay); put2(lv.data(), 64, bx); put2(lv.data(), 96, by);
    if (lv[sf + 4] == GF::one()) generate_g2_op(false, ax, ay, bx, by, lv.data() + 128, nx, ny); else write_g2_output_default(lv.data() + 128);
    for (int c = 0; c < num_main_cols; c++) cols[c][row0] = lv[c];
    for (size_t i = 0; i + 1 < num_rows; i++) {
      std::fill(nvv.begin(), nvv.end(), GF());
      generate_flags_next_row(lv.data(), nvv.data(), i, sf);
      if (lv[sf + 2] == GF::one()) { ax = nx; ay = ny; } else if (lv[sf + 4] == GF::one()) { bx = nx; by = ny; }
      put2(nvv.data(), 0, ax); put2(nvv.data(), 32, ay); put2(nvv.data(), 64, bx); put2(nvv.data(), 96, by);
      if (nvv[sf + 2] == GF::one()) generate_g2_op(true, ax, ay, ax, ay, nvv.data() + 128, nx, ny);
      else if (nvv[sf + 4] == GF::one()) generate_g2_op(false, ax, ay, bx, by, nvv.data() + 128, nx, ny);
      else write_g2_output_default(nvv.data() + 128);
      for (int c = 0; c < num_main_cols; c++) cols[c][row0 + i + 1] = nvv[c];
      lv.swap(nvv);
    }
    out.x = bx; out.y = by;
  }
  std::vector<std::vector<GF>> generate_trace(std::vector<G2ExpIONative>& inputs) const {  // :307-335
    assert(inputs.size() == num_io);
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS, rows = rpb * num_io;
    std::vector<std::vector<GF>> cols(num_main_cols, std::vector<GF>(rows));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < num_io; k++) generate_block(inputs[k], cols, k * rpb, inputs[k].output);
    generate_periodic_pulse_witness(cols, start_flags_col + 1, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    generate_pulse(cols, pulse_positions);
    generate_u16_range_check(0, (size_t)num_range_check_cols, cols);
    assert(cols.size() == ncols);
    return cols;
  }
  std::vector<GF> generate_public_inputs(const std::vector<G2ExpIONative>& inputs) const {  // :139-156, :337-342
    std::vector<GF> pi;
    auto put = [&](const U256& v) { for (int i = 0; i < 8; i++) pi.push_back(GF((v.l[i / 2] >> (32 * (i % 2))) & 0xffffffffULL)); };
    auto put2 = [&](const U256x2& v) { put(v.c0); put(v.c1); };
    for (auto& in : inputs) {
      put2(in.x.x); put2(in.x.y); put2(in.offset.x); put2(in.offset.y);
      for (int i = 0; i < 8; i++) pi.push_back(GF(in.exp_val[i]));
      put2(in.output.x); put2(in.output.y);
    }
    return pi;
  }
};

// ---- src/fields/fq12/mul.rs --------------------------------------------------------------------------------------
// Fq12 in the reference's FLAT basis: sum_k (r_k + s_k i) w^k, k < 6, w^6 = xi = 9 + i; coefficients
// [0..6) = r_k, [6..12) = s_k (the order `MyFq12.coeffs` has, plonky2-bn254; utils.rs:174-183).
template <class T> using Fq12Limbs = Arr<Arr<T, 16>, 12>;
template <class T> using Fq12Wide = Arr<Arr<T, 31>, 12>;
// pol_mul_fq12 :24-87
template <class T> static inline Fq12Wide<T> pol_mul_fq12(const Fq12Limbs<T>& a, const Fq12Limbs<T>& b, T xi) {
  Arr<Arr<T, 31>, 11> d, s;  // a0b0 - a1b1, a0b1 + a1b0 by power of w
  for (auto& x : d) for (auto& y : x) y = T();
  for (auto& x : s) for (auto& y : x) y = T();
  for (int i = 0; i < 6; i++)
    for (int j = 0; j < 6; j++) {
      Arr<T, 31> c00 = pol_mul_wide(a[i], b[j]), c01 = pol_mul_wide(a[i], b[j + 6]), c10 = pol_mul_wide(a[i + 6], b[j]), c11 = pol_mul_wide(a[i + 6], b[j + 6]);
      for (int k = 0; k < 31; k++) { d[i + j][k] = d[i + j][k] + c00[k] - c11[k]; s[i + j][k] = s[i + j][k] + c01[k] + c10[k]; }
    }
  Fq12Wide<T> out;
  for (int i = 0; i < 6; i++)
    for (int k = 0; k < 31; k++) {
      if (i < 5) { out[i][k] = d[i][k] + xi * d[i + 6][k] - s[i + 6][k]; out[i + 6][k] = s[i][k] + d[i + 6][k] + xi * s[i + 6][k]; }
      else { out[i][k] = d[i][k]; out[i + 6][k] = s[i][k]; }
    }
  return out;
}
template <class T> struct Fq12Output {  // :173-177 ; 84*N_LIMBS columns (:217-231)
  Fq12Limbs<T> output; ModulusAux<T> auxs[12]; T quot_signs[12];
};
static const int FQ12_OUTPUT_COLS = 84 * N_LIMBS;
template <class T> static inline Fq12Limbs<T> read_fq12(const T* lv, int& cur) { Fq12Limbs<T> r; for (int i = 0; i < 12; i++) r[i] = read16(lv, cur); return r; }
template <class T> static inline Fq12Output<T> read_fq12_output(const T* lv, int& cur) {  // :233-252
  Fq12Output<T> o;
  o.output = read_fq12(lv, cur);
  for (int i = 0; i < 12; i++) { o.auxs[i].out_aux_red = read16(lv, cur); o.auxs[i].quot_abs = readn<T, 17>(lv, cur); o.auxs[i].aux_input_lo = readn<T, 31>(lv, cur); o.auxs[i].aux_input_hi = readn<T, 31>(lv, cur); }
  for (int i = 0; i < 12; i++) o.quot_signs[i] = lv[cur++];
  return o;
}
// eval_fq12_mul :254-275
template <class P>
static inline void eval_fq12_mul(Consumer<P>& yc, P filter, const Fq12Limbs<P>& x, const Fq12Limbs<P>& y, const Fq12Output<P>& o) {
  Fq12Wide<P> input = pol_mul_fq12(x, y, cst<P>(9));
  Arr<P, 16> modulus = bn254_modulus_p<P>();
  for (int i = 0; i < 12; i++) eval_modular_op(yc, filter, modulus, input[i], o.output[i], o.quot_signs[i], o.auxs[i]);
}
// generate_fq12_mul :192-215 : writes lv[0..1344), returns the product limbs
static inline void generate_fq12_mul(const Fq12Limbs<int64_t>& x, const Fq12Limbs<int64_t>& y, GF* lv, Fq12Limbs<int64_t>& out) {
  Fq12Wide<int64_t> pol_input = pol_mul_fq12<int64_t>(x, y, 9);
  ModWitness w[12];
  for (int i = 0; i < 12; i++) { w[i] = generate_modular_witness(pol_input[i], false); out[i] = w[i].output; }
  int cur = 0;
  for (int i = 0; i < 12; i++) for (int k = 0; k < 16; k++) lv[cur++] = GF((u64)w[i].output[k]);
  for (int i = 0; i < 12; i++) write_mod_aux(lv, cur, w[i], true);
  for (int i = 0; i < 12; i++) lv[cur++] = GF::from_i64(w[i].quot_sign);
  assert(cur == FQ12_OUTPUT_COLS);
}
static inline void write_fq12_output_default(GF* lv) {  // Fq12Output::default :179-187
  for (int i = 0; i < FQ12_OUTPUT_COLS - 12; i++) lv[i] = GF();
  for (int i = FQ12_OUTPUT_COLS - 12; i < FQ12_OUTPUT_COLS; i++) lv[i] = GF::one();
}

// Fq12ExpStark: src/fields/fq12/exp.rs  (offset * x^e by square-and-multiply; split range check).
struct Fq12ExpIONative { U256 x[12], offset[12]; uint32_t exp_val[NUM_INPUT_LIMBS]; U256 output[12]; };  // :86-91
struct Fq12ExpAir : AirBase<Fq12ExpAir> {
  size_t num_io;
  int start_flags_col, num_main_cols, start_periodic_pulse_col, start_io_pulses_col, start_lookups_col, start_range_check_col, num_range_check_cols;  // :6-34
  size_t ncols, npi;
  std::vector<size_t> pulse_positions;
  static constexpr int IO_LEN = 36 * N_LIMBS + NUM_INPUT_LIMBS;  // :93
  explicit Fq12ExpAir(size_t n) : num_io(n) {
    start_flags_col = 108 * N_LIMBS;
    num_main_cols = start_flags_col + NUM_FLAGS_COLS;
    start_periodic_pulse_col = num_main_cols;
    start_io_pulses_col = start_periodic_pulse_col + 2;
    start_lookups_col = start_io_pulses_col + 1 + 4 * (int)num_io;
    start_range_check_col = 24 * N_LIMBS;
    num_range_check_cols = 84 * N_LIMBS - 12;
    ncols = start_lookups_col + 1 + 6 * num_range_check_cols;
    npi = IO_LEN * num_io;
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    for (size_t i = 0; i < num_io; i++) { pulse_positions.push_back(i * rpb); pulse_positions.push_back(i * rpb + rpb - 1); }
  }
  size_t num_columns() const override { return ncols; }
  size_t num_public_inputs() const override { return npi; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // :598-604, range_check.rs:230-246
    std::vector<std::pair<size_t, size_t>> p;
    size_t mc = start_lookups_col;
    for (size_t i = mc + 1; i < mc + 1 + 6 * (size_t)num_range_check_cols; i += 6) {
      p.push_back({mc, i + 2}); p.push_back({mc, i + 5}); p.push_back({i, i + 1}); p.push_back({i + 3, i + 4});
    }
    return p;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P* pi, Consumer<P>& yc) const {  // :323-428
    int is_final_col = start_flags_col, is_sq_col = start_flags_col + 2, is_mul_col = start_flags_col + 4, start_limbs_col = start_flags_col + 6;
    P one = cst<P>(1);
    int cur = 0;
    Fq12Limbs<P> a = read_fq12(lv, cur), b = read_fq12(lv, cur);
    Fq12Output<P> output = read_fq12_output(lv, cur);
    P is_mul = lv[is_mul_col], is_sq = lv[is_sq_col], is_final = lv[is_final_col];
    P is_not_final = one - is_final;
    P sum_is_output = P();
    for (size_t i = 1; i < 2 * num_io; i += 2) sum_is_output = sum_is_output + lv[get_pulse_col(start_io_pulses_col, (int)i)];
    yc.constraint(is_final - sum_is_output);
    {  // public inputs :358-373 ; layout x[12][16] offset[12][16] exp_val[8] output[12][16] (read_fq12_exp_io :119-131)
      Arr<P, 8> limbs; for (int k = 0; k < 8; k++) limbs[k] = lv[start_limbs_col + k];
      limbs[0] = limbs[0] * cst<P>(2) + is_mul;
      size_t pc = 0;
      for (size_t i = 0; i < 2 * num_io; i += 2) {
        const P* io = pi + pc; pc += IO_LEN;
        P is_in = lv[get_pulse_col(start_io_pulses_col, (int)i)], is_out = lv[get_pulse_col(start_io_pulses_col, (int)i + 1)];
        for (int c = 0; c < 12; c++) {
          for (int k = 0; k < 16; k++) yc.constraint(is_in * (io[16 * c + k] - a[c][k]));
          for (int k = 0; k < 16; k++) yc.constraint(is_in * (io[192 + 16 * c + k] - b[c][k]));
          for (int k = 0; k < 16; k++) yc.constraint(is_out * (io[392 + 16 * c + k] - b[c][k]));
        }
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[384 + k] - limbs[k]));
      }
    }
    cur = 0;
    Fq12Limbs<P> next_a = read_fq12(nv, cur), next_b = read_fq12(nv, cur);
    auto eqt = [&](P filter, const Fq12Limbs<P>& x, const Fq12Limbs<P>& y) { for (int c = 0; c < 12; c++) for (int k = 0; k < 16; k++) yc.constraint_transition(filter * (x[c][k] - y[c][k])); };
    P fs = is_not_final * is_sq;
    eqt(fs, next_a, output.output); eqt(fs, next_b, b);
    P fm = is_not_final * is_mul;
    eqt(fm, next_a, a); eqt(fm, next_b, output.output);
    P fn = is_not_final * (one - is_sq - is_mul);
    eqt(fn, next_a, a); eqt(fn, next_b, b);
    eval_flags(yc, lv, nv, start_flags_col);           // :395
    eval_fq12_mul(yc, is_sq, a, a, output);            // :396
    eval_fq12_mul(yc, is_mul, a, b, output);           // :397
    eval_flags(yc, lv, nv, start_flags_col);           // :400 (duplicate, kept)
    eval_periodic_pulse(yc, lv, nv, start_flags_col + 1, start_periodic_pulse_col, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    eval_pulse(yc, lv, nv, start_io_pulses_col, pulse_positions);
    eval_split_u16_range_check(yc, lv, nv, start_lookups_col, (size_t)start_range_check_col, (size_t)(start_range_check_col + num_range_check_cols));
  }
  void generate_block(const Fq12ExpIONative& in, std::vector<std::vector<GF>>& cols, size_t row0, U256* out) const {  // :229-268
    size_t num_rows = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    std::vector<GF> lv(num_main_cols, GF()), nvv(num_main_cols, GF());
    int sf = start_flags_col;
    Fq12Limbs<int64_t> a, b, prod;
    for (int c = 0; c < 12; c++) { a[c] = u256_to_limbs16(in.x[c]); b[c] = u256_to_limbs16(in.offset[c]); }
    auto put = [&](GF* r, int off, const Fq12Limbs<int64_t>& v) { for (int c = 0; c < 12; c++) for (int k = 0; k < 16; k++) r[off + 16 * c + k] = GF((u64)v[c][k]); };
    generate_flags_first_row(lv.data(), sf, in.exp_val);
    put(lv.data(), 0, a); put(lv.data(), 192, b);
    if (lv[sf + 4] == GF::one()) generate_fq12_mul(a, b, lv.data() + 384, prod); else write_fq12_output_default(lv.data() + 384);
    for (int c = 0; c < num_main_cols; c++) cols[c][row0] = lv[c];
    for (size_t i = 0; i + 1 < num_rows; i++) {
      std::fill(nvv.begin(), nvv.end(), GF());
      generate_flags_next_row(lv.data(), nvv.data(), i, sf);
      if (lv[sf + 2] == GF::one()) a = prod; else if (lv[sf + 4] == GF::one()) b = prod;
      put(nvv.data(), 0, a); put(nvv.data(), 192, b);
      if (nvv[sf + 2] == GF::one()) generate_fq12_mul(a, a, nvv.data() + 384, prod);
      else if (nvv[sf + 4] == GF::one()) generate_fq12_mul(a, b, nvv.data() + 384, prod);
      else write_fq12_output_default(nvv.data() + 384);
      for (int c = 0; c < num_main_cols; c++) cols[c][row0 + i + 1] = nvv[c];
      lv.swap(nvv);
    }
    for (int c = 0; c < 12; c++) out[c] = limbs16_to_u256(b[c].data());
  }
  std::vector<std::vector<GF>> generate_trace(std::vector<Fq12ExpIONative>& inputs) const {  // :283-312
    assert(inputs.size() == num_io);
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS, rows = rpb * num_io;
    std::vector<std::vector<GF>> cols(num_main_cols, std::vector<GF>(rows));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < num_io; k++) generate_block(inputs[k], cols, k * rpb, inputs[k].output);
    generate_periodic_pulse_witness(cols, start_flags_col + 1, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    generate_pulse(cols, pulse_positions);
    generate_split_u16_range_check((size_t)start_range_check_col, (size_t)(start_range_check_col + num_range_check_cols), cols);
    assert(cols.size() == ncols);
    return cols;
  }
  std::vector<GF> generate_public_inputs(const std::vector<Fq12ExpIONative>& inputs) const {  // :95-117, :314-319
    std::vector<GF> pi;
    auto put = [&](const U256& v) { auto l = u256_to_limbs16(v); for (int k = 0; k < 16; k++) pi.push_back(GF((u64)l[k])); };
    for (auto& in : inputs) {
      for (int c = 0; c < 12; c++) put(in.x[c]);
      for (int c = 0; c < 12; c++) put(in.offset[c]);
      for (int i = 0; i < 8; i++) pi.push_back(GF(in.exp_val[i]));
      for (int c = 0; c < 12; c++) put(in.output[c]);
    }
    return pi;
  }
};

// ---- Fq12ExpU64Stark: src/fields/fq12_u64/{flags_u64,exp_u64}.rs (offset * x^e in Fq12, e a u64: 128 rows per instance) ----
static const int NUM_FLAGS_U64_COLS = 6;  // flags_u64.rs:21 : is_final, a, b, filtered_bit, bit, val
static inline void generate_flags_u64_first_row(GF* lv, int s, uint64_t exp_val) {  // :34-55
  uint64_t first_bit = exp_val & 1, rest = exp_val >> 1;
  lv[s] = GF(); lv[s + 1] = GF(); lv[s + 2] = GF::one(); lv[s + 3] = GF(first_bit); lv[s + 4] = GF(first_bit); lv[s + 5] = GF(rest);
}
static inline void generate_flags_u64_next_row(const GF* lv, GF* nv, size_t cur_row, int s) {  // :57-94
  nv[s + 1] = GF::one() - lv[s + 1];
  nv[s + 2] = GF::one() - lv[s + 2];
  nv[s] = cur_row == 2 * 64 - 2 ? GF::one() : GF();
  if (lv[s + 1] == GF::one()) { uint64_t v = lv[s + 5].v; nv[s + 4] = GF(v & 1); nv[s + 5] = GF(v >> 1); }
  else { nv[s + 4] = lv[s + 4]; nv[s + 5] = lv[s + 5]; }
  nv[s + 3] = nv[s + 4] * nv[s + 2];
}
template <class P>
static inline void eval_flags_u64(Consumer<P>& yc, const P* lv, const P* nv, int s) {  // :96-139
  P one = cst<P>(1), two = cst<P>(2);
  yc.constraint_first_row(lv[s + 1]);
  yc.constraint_first_row(lv[s + 2] - one);
  P bit = lv[s + 4];
  yc.constraint(bit * bit - bit);
  yc.constraint(bit * lv[s + 2] - lv[s + 3]);
  yc.constraint_transition(lv[s + 1] + nv[s + 1] - one);
  yc.constraint_transition(lv[s + 2] + nv[s + 2] - one);
  P first_limb = lv[s + 5], next_first_limb = nv[s + 5], next_bit = nv[s + 4], is_split = lv[s + 1];
  P is_not_final = one - lv[s], is_not_split = one - is_split;
  yc.constraint_transition(is_not_final * is_split * (first_limb - two * next_first_limb - next_bit));
  yc.constraint_transition(is_not_split * (next_bit - bit));
  yc.constraint_transition(is_not_final * is_not_split * (first_limb - next_first_limb));
}
struct Fq12ExpU64IONative { U256 x[12], offset[12]; uint64_t exp_val; U256 output[12]; };  // exp_u64.rs:91-96
struct Fq12ExpU64Air : AirBase<Fq12ExpU64Air> {
  size_t num_io;
  int start_flags_col, num_main_cols, start_io_pulses_col, start_lookups_col, start_range_check_col, num_range_check_cols;  // exp_u64.rs:19-45
  size_t ncols, npi;
  std::vector<size_t> pulse_positions;
  static constexpr int IO_LEN = 36 * N_LIMBS + 1;  // :98
  static constexpr size_t RPB = 2 * 64;
  explicit Fq12ExpU64Air(size_t n) : num_io(n) {
    start_flags_col = 108 * N_LIMBS;
    num_main_cols = start_flags_col + NUM_FLAGS_U64_COLS;
    start_io_pulses_col = num_main_cols;
    start_lookups_col = start_io_pulses_col + 1 + 4 * (int)num_io;
    start_range_check_col = 24 * N_LIMBS;
    num_range_check_cols = 84 * N_LIMBS - 12;
    ncols = start_lookups_col + 1 + 6 * num_range_check_cols;
    npi = IO_LEN * num_io;
    for (size_t i = 0; i < num_io; i++) { pulse_positions.push_back(i * RPB); pulse_positions.push_back(i * RPB + RPB - 1); }  // :230-241
  }
  size_t num_columns() const override { return ncols; }
  size_t num_public_inputs() const override { return npi; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // :564-570, range_check.rs:230-246
    std::vector<std::pair<size_t, size_t>> p;
    size_t mc = start_lookups_col;
    for (size_t i = mc + 1; i < mc + 1 + 6 * (size_t)num_range_check_cols; i += 6) {
      p.push_back({mc, i + 2}); p.push_back({mc, i + 5}); p.push_back({i, i + 1}); p.push_back({i + 3, i + 4});
    }
    return p;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P* pi, Consumer<P>& yc) const {  // :321-413
    int is_final_col = start_flags_col, is_sq_col = start_flags_col + 1, is_mul_col = start_flags_col + 3, exp_val_col = start_flags_col + 5;
    P one = cst<P>(1);
    int cur = 0;
    Fq12Limbs<P> a = read_fq12(lv, cur), b = read_fq12(lv, cur);
    Fq12Output<P> output = read_fq12_output(lv, cur);
    P is_mul = lv[is_mul_col], is_sq = lv[is_sq_col], is_final = lv[is_final_col];
    P is_not_final = one - is_final;
    P sum_is_output = P();
    for (size_t i = 1; i < 2 * num_io; i += 2) sum_is_output = sum_is_output + lv[get_pulse_col(start_io_pulses_col, (int)i)];
    yc.constraint(is_final - sum_is_output);
    {  // public inputs :351-367 ; layout x[12][16] offset[12][16] exp_val output[12][16] (read_fq12_exp_u64_io :130-145)
      P recovered = lv[exp_val_col] * cst<P>(2) + is_mul;
      size_t pc = 0;
      for (size_t i = 0; i < 2 * num_io; i += 2) {
        const P* io = pi + pc; pc += IO_LEN;
        P is_in = lv[get_pulse_col(start_io_pulses_col, (int)i)], is_out = lv[get_pulse_col(start_io_pulses_col, (int)i + 1)];
        for (int c = 0; c < 12; c++) {
          for (int k = 0; k < 16; k++) yc.constraint(is_in * (io[16 * c + k] - a[c][k]));
          for (int k = 0; k < 16; k++) yc.constraint(is_in * (io[192 + 16 * c + k] - b[c][k]));
          for (int k = 0; k < 16; k++) yc.constraint(is_out * (io[385 + 16 * c + k] - b[c][k]));
        }
        yc.constraint(is_in * (io[384] - recovered));
      }
    }
    cur = 0;
    Fq12Limbs<P> next_a = read_fq12(nv, cur), next_b = read_fq12(nv, cur);
    auto eqt = [&](P filter, const Fq12Limbs<P>& x, const Fq12Limbs<P>& y) { for (int c = 0; c < 12; c++) for (int k = 0; k < 16; k++) yc.constraint_transition(filter * (x[c][k] - y[c][k])); };
    P fs = is_not_final * is_sq;
    eqt(fs, next_a, output.output); eqt(fs, next_b, b);
    P fm = is_not_final * is_mul;
    eqt(fm, next_a, a); eqt(fm, next_b, output.output);
    P fn = is_not_final * (one - is_sq - is_mul);
    eqt(fn, next_a, a); eqt(fn, next_b, b);
    eval_flags_u64(yc, lv, nv, start_flags_col);     // :388
    eval_fq12_mul(yc, is_sq, a, a, output);          // :389
    eval_fq12_mul(yc, is_mul, a, b, output);         // :390
    eval_flags_u64(yc, lv, nv, start_flags_col);     // :393 (duplicate, kept)
    eval_pulse(yc, lv, nv, start_io_pulses_col, pulse_positions);
    eval_split_u16_range_check(yc, lv, nv, start_lookups_col, (size_t)start_range_check_col, (size_t)(start_range_check_col + num_range_check_cols));
  }
  void generate_block(const Fq12ExpU64IONative& in, std::vector<std::vector<GF>>& cols, size_t row0, U256* out) const {  // :147-228, :256-281
    std::vector<GF> lv(num_main_cols, GF()), nvv(num_main_cols, GF());
    int sf = start_flags_col;
    Fq12Limbs<int64_t> a, b, prod;
    for (int c = 0; c < 12; c++) { a[c] = u256_to_limbs16(in.x[c]); b[c] = u256_to_limbs16(in.offset[c]); }
    auto put = [&](GF* r, int off, const Fq12Limbs<int64_t>& v) { for (int c = 0; c < 12; c++) for (int k = 0; k < 16; k++) r[off + 16 * c + k] = GF((u64)v[c][k]); };
    generate_flags_u64_first_row(lv.data(), sf, in.exp_val);
    put(lv.data(), 0, a); put(lv.data(), 192, b);
    if (lv[sf + 3] == GF::one()) generate_fq12_mul(a, b, lv.data() + 384, prod); else write_fq12_output_default(lv.data() + 384);
    for (int c = 0; c < num_main_cols; c++) cols[c][row0] = lv[c];
    for (size_t i = 0; i + 1 < RPB; i++) {
      std::fill(nvv.begin(), nvv.end(), GF());
      generate_flags_u64_next_row(lv.data(), nvv.data(), i, sf);
      if (lv[sf + 1] == GF::one()) a = prod; else if (lv[sf + 3] == GF::one()) b = prod;
      put(nvv.data(), 0, a); put(nvv.data(), 192, b);
      if (nvv[sf + 1] == GF::one()) generate_fq12_mul(a, a, nvv.data() + 384, prod);
      else if (nvv[sf + 3] == GF::one()) generate_fq12_mul(a, b, nvv.data() + 384, prod);
      else write_fq12_output_default(nvv.data() + 384);
      for (int c = 0; c < num_main_cols; c++) cols[c][row0 + i + 1] = nvv[c];
      lv.swap(nvv);
    }
    for (int c = 0; c < 12; c++) out[c] = limbs16_to_u256(b[c].data());
  }
  std::vector<std::vector<GF>> generate_trace(std::vector<Fq12ExpU64IONative>& inputs) const {  // :283-306
    assert(inputs.size() == num_io);
    size_t rows = RPB * num_io;
    std::vector<std::vector<GF>> cols(num_main_cols, std::vector<GF>(rows));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < num_io; k++) generate_block(inputs[k], cols, k * RPB, inputs[k].output);
    generate_pulse(cols, pulse_positions);
    generate_split_u16_range_check((size_t)start_range_check_col, (size_t)(start_range_check_col + num_range_check_cols), cols);
    assert(cols.size() == ncols);
    return cols;
  }
  std::vector<GF> generate_public_inputs(const std::vector<Fq12ExpU64IONative>& inputs) const {  // :100-128, :308-313
    std::vector<GF> pi;
    auto put = [&](const U256& v) { auto l = u256_to_limbs16(v); for (int k = 0; k < 16; k++) pi.push_back(GF((u64)l[k])); };
    for (auto& in : inputs) {
      for (int c = 0; c < 12; c++) put(in.x[c]);
      for (int c = 0; c < 12; c++) put(in.offset[c]);
      pi.push_back(GF(in.exp_val));
      for (int c = 0; c < 12; c++) put(in.output[c]);
    }
    return pi;
  }
};

// ---- FqExpStark: src/fields/fq/exp.rs (offset * x^e in Fq by square-and-multiply; u16 range check) --------------------
// FqOutput (src/fields/fq/mul.rs:18-22, column layout :49-54): output[16], ModulusAux (95), quot_sign = 7*N_LIMBS columns.
template <class T> struct FqOutput { Arr<T, 16> output; ModulusAux<T> aux; T quot_sign; };
static const int FQ_OUTPUT_COLS = 7 * N_LIMBS;
template <class T> static inline FqOutput<T> read_fq_output(const T* lv, int& cur) {  // mul.rs:57-68
  FqOutput<T> o;
  o.output = read16(lv, cur);
  o.aux.out_aux_red = read16(lv, cur); o.aux.quot_abs = readn<T, 17>(lv, cur); o.aux.aux_input_lo = readn<T, 31>(lv, cur); o.aux.aux_input_hi = readn<T, 31>(lv, cur);
  o.quot_sign = lv[cur++];
  return o;
}
template <class P>
static inline void eval_fq_mul(Consumer<P>& yc, P filter, const Arr<P, 16>& x, const Arr<P, 16>& y, const FqOutput<P>& o) {  // mul.rs:70-88
  eval_modular_op(yc, filter, bn254_modulus_p<P>(), pol_mul_wide(x, y), o.output, o.quot_sign, o.aux);
}
// generate_fq_mul (mul.rs:35-46) + write_fq_output: writes lv[0..112), returns the product limbs
static inline void generate_fq_mul(const Arr<int64_t, 16>& x, const Arr<int64_t, 16>& y, GF* lv, Arr<int64_t, 16>& out) {
  ModWitness w = generate_modular_witness(pol_mul_wide<int64_t>(x, y), false);
  out = w.output;
  int cur = 0;
  for (int k = 0; k < 16; k++) lv[cur++] = GF((u64)w.output[k]);
  write_mod_aux(lv, cur, w, true);
  lv[cur++] = GF::from_i64(w.quot_sign);
  assert(cur == FQ_OUTPUT_COLS);
}
static inline void write_fq_output_default(GF* lv) {  // FqOutput::default mul.rs:24-32
  for (int i = 0; i < FQ_OUTPUT_COLS - 1; i++) lv[i] = GF();
  lv[FQ_OUTPUT_COLS - 1] = GF::one();
}
struct FqExpIONative { U256 x, offset; uint32_t exp_val[NUM_INPUT_LIMBS]; U256 output; };  // exp.rs:89-94
struct FqExpAir : AirBase<FqExpAir> {
  size_t num_io;
  int start_flags_col, num_main_cols, start_periodic_pulse_col, start_io_pulses_col, start_lookups_col, num_range_check_cols;  // exp.rs:6-34
  size_t ncols, npi;
  std::vector<size_t> pulse_positions;
  static constexpr int IO_LEN = 4 * NUM_INPUT_LIMBS;  // exp.rs:96
  explicit FqExpAir(size_t n) : num_io(n) {
    start_flags_col = 9 * N_LIMBS;
    num_main_cols = start_flags_col + NUM_FLAGS_COLS;
    start_periodic_pulse_col = num_main_cols;
    start_io_pulses_col = start_periodic_pulse_col + 2;
    start_lookups_col = start_io_pulses_col + 1 + 4 * (int)num_io;
    num_range_check_cols = 9 * N_LIMBS - 1;
    ncols = start_lookups_col + 1 + 2 * num_range_check_cols;
    npi = IO_LEN * num_io;
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    for (size_t i = 0; i < num_io; i++) { pulse_positions.push_back(i * rpb); pulse_positions.push_back(i * rpb + rpb - 1); }
  }
  size_t num_columns() const override { return ncols; }
  size_t num_public_inputs() const override { return npi; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override {  // exp.rs:574-580, range_check.rs:96-113
    std::vector<std::pair<size_t, size_t>> p;
    for (int i = 0; i < num_range_check_cols; i++) {
      p.push_back({(size_t)start_lookups_col, (size_t)start_lookups_col + 1 + 2 * i + 1});
      p.push_back({(size_t)i, (size_t)start_lookups_col + 1 + 2 * i});
    }
    return p;
  }
  template <class P> void eval_t(const P* lv, const P* nv, const P* pi, Consumer<P>& yc) const {  // exp.rs:288-394
    int is_final_col = start_flags_col, is_sq_col = start_flags_col + 2, is_mul_col = start_flags_col + 4, start_limbs_col = start_flags_col + 6;
    P one = cst<P>(1);
    int cur = 0;
    Arr<P, 16> a = read16(lv, cur), b = read16(lv, cur);
    FqOutput<P> output = read_fq_output(lv, cur);
    P is_mul = lv[is_mul_col], is_sq = lv[is_sq_col], is_final = lv[is_final_col];
    P is_not_final = one - is_final;
    P sum_is_output = P();
    for (size_t i = 1; i < 2 * num_io; i += 2) sum_is_output = sum_is_output + lv[get_pulse_col(start_io_pulses_col, (int)i)];
    yc.constraint(is_final - sum_is_output);
    {  // public inputs :320-337 ; x offset exp_val output as 8 u32 limbs each (read_fq_exp_io :110-126)
      Arr<P, 8> a32 = G1ExpAir::u16_to_u32(a), b32 = G1ExpAir::u16_to_u32(b);
      Arr<P, 8> limbs; for (int k = 0; k < 8; k++) limbs[k] = lv[start_limbs_col + k];
      limbs[0] = limbs[0] * cst<P>(2) + is_mul;
      size_t pc = 0;
      for (size_t i = 0; i < 2 * num_io; i += 2) {
        const P* io = pi + pc; pc += IO_LEN;
        P is_in = lv[get_pulse_col(start_io_pulses_col, (int)i)], is_out = lv[get_pulse_col(start_io_pulses_col, (int)i + 1)];
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[k] - a32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[8 + k] - b32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_out * (io[24 + k] - b32[k]));
        for (int k = 0; k < 8; k++) yc.constraint(is_in * (io[16 + k] - limbs[k]));
      }
    }
    cur = 0;
    Arr<P, 16> next_a = read16(nv, cur), next_b = read16(nv, cur);
    auto eqt = [&](P filter, const Arr<P, 16>& x, const Arr<P, 16>& y) { for (int k = 0; k < 16; k++) yc.constraint_transition(filter * (x[k] - y[k])); };
    P fs = is_not_final * is_sq;
    eqt(fs, next_a, output.output); eqt(fs, next_b, b);
    P fm = is_not_final * is_mul;
    eqt(fm, next_a, a); eqt(fm, next_b, output.output);
    P fn = is_not_final * (one - is_sq - is_mul);
    eqt(fn, next_a, a); eqt(fn, next_b, b);
    eval_flags(yc, lv, nv, start_flags_col);     // :361
    eval_fq_mul(yc, is_sq, a, a, output);        // :362
    eval_fq_mul(yc, is_mul, a, b, output);       // :363
    eval_flags(yc, lv, nv, start_flags_col);     // :366 (duplicate, kept)
    eval_periodic_pulse(yc, lv, nv, start_flags_col + 1, start_periodic_pulse_col, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    eval_pulse(yc, lv, nv, start_io_pulses_col, pulse_positions);
    eval_u16_range_check(yc, lv, nv, start_lookups_col, (size_t)num_range_check_cols);
  }
  void generate_block(const FqExpIONative& in, std::vector<std::vector<GF>>& cols, size_t row0, U256& out) const {  // :128-176, :214-246
    size_t num_rows = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    std::vector<GF> lv(num_main_cols, GF()), nvv(num_main_cols, GF());
    int sf = start_flags_col;
    Arr<int64_t, 16> a = u256_to_limbs16(in.x), b = u256_to_limbs16(in.offset), prod;
    auto put = [&](GF* r, int off, const Arr<int64_t, 16>& v) { for (int k = 0; k < 16; k++) r[off + k] = GF((u64)v[k]); };
    generate_flags_first_row(lv.data(), sf, in.exp_val);
    put(lv.data(), 0, a); put(lv.data(), 16, b);
    if (lv[sf + 4] == GF::one()) generate_fq_mul(a, b, lv.data() + 32, prod); else write_fq_output_default(lv.data() + 32);
    for (int c = 0; c < num_main_cols; c++) cols[c][row0] = lv[c];
    for (size_t i = 0; i + 1 < num_rows; i++) {
      std::fill(nvv.begin(), nvv.end(), GF());
      generate_flags_next_row(lv.data(), nvv.data(), i, sf);
      if (lv[sf + 2] == GF::one()) a = prod; else if (lv[sf + 4] == GF::one()) b = prod;
      put(nvv.data(), 0, a); put(nvv.data(), 16, b);
      if (nvv[sf + 2] == GF::one()) generate_fq_mul(a, a, nvv.data() + 32, prod);
      else if (nvv[sf + 4] == GF::one()) generate_fq_mul(a, b, nvv.data() + 32, prod);
      else write_fq_output_default(nvv.data() + 32);
      for (int c = 0; c < num_main_cols; c++) cols[c][row0 + i + 1] = nvv[c];
      lv.swap(nvv);
    }
    out = limbs16_to_u256(b.data());
  }
  std::vector<std::vector<GF>> generate_trace(std::vector<FqExpIONative>& inputs) const {  // :248-277
    assert(inputs.size() == num_io);
    size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS, rows = rpb * num_io;
    std::vector<std::vector<GF>> cols(num_main_cols, std::vector<GF>(rows));
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t k = 0; k < num_io; k++) generate_block(inputs[k], cols, k * rpb, inputs[k].output);
    generate_periodic_pulse_witness(cols, start_flags_col + 1, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    generate_pulse(cols, pulse_positions);
    generate_u16_range_check(0, (size_t)num_range_check_cols, cols);
    assert(cols.size() == ncols);
    return cols;
  }
  std::vector<GF> generate_public_inputs(const std::vector<FqExpIONative>& inputs) const {  // :98-108, :279-284
    std::vector<GF> pi;
    auto put32 = [&](const U256& v) { for (int i = 0; i < 8; i++) pi.push_back(GF((v.l[i >> 1] >> (32 * (i & 1))) & 0xffffffffULL)); };
    for (auto& in : inputs) {
      put32(in.x); put32(in.offset);
      for (int i = 0; i < 8; i++) pi.push_back(GF(in.exp_val[i]));
      put32(in.output);
    }
    return pi;
  }
};

// ---- the reference's own single-operation test tables (512 rows in its tests; split range check like G1Stark) ----------
// split_u16_range_check_pairs (range_check.rs:228-246) for targets [main_col + 1 + 6 i ...)
static inline std::vector<std::pair<size_t, size_t>> split_range_check_pairs(size_t main_cols, size_t num_rc) {
  std::vector<std::pair<size_t, size_t>> p;
  for (size_t i = main_cols + 1; i < main_cols + 1 + 6 * num_rc; i += 6) {
    p.push_back({main_cols, i + 2}); p.push_back({main_cols, i + 5});
    p.push_back({i, i + 1}); p.push_back({i + 3, i + 4});
  }
  return p;
}
// ModularStark: src/modular/modular.rs:361-537.  Row: input0[16] input1[16] output[16] ModulusAux(95) quot_sign filter;
// output and aux are range-checked; constraints: the split range check, then eval_modular_op on input0 * input1.
struct ModularAir : AirBase<ModularAir> {
  static constexpr int MAIN_COLS = 9 * N_LIMBS + 1;                                                 // :361
  static constexpr int START_RC = 2 * N_LIMBS, NUM_RC = 7 * N_LIMBS - 1, END_RC = START_RC + NUM_RC;  // :364-366
  size_t num_columns() const override { return MAIN_COLS + 1 + 6 * NUM_RC; }                    // :368
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override { return split_range_check_pairs(MAIN_COLS, NUM_RC); }  // :533-535
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const {  // :441-477
    eval_split_u16_range_check(yc, lv, nv, MAIN_COLS, START_RC, END_RC);
    int cur = 0;
    auto input0 = read16(lv, cur), input1 = read16(lv, cur);
    FqOutput<P> o = read_fq_output(lv, cur);   // output, aux, quot_sign in this order (read_u256, read_modulus_aux, lv[cur])
    P filter = lv[cur++];
    assert(cur == MAIN_COLS);
    eval_modular_op(yc, filter, bn254_modulus_p<P>(), pol_mul_wide(input0, input1), o.output, o.quot_sign, o.aux);
  }
  // generate_trace :383-437 with caller-provided operands (the reference draws them at random)
  std::vector<std::vector<GF>> generate_trace(const std::vector<U256>& a, const std::vector<U256>& b) const {
    size_t rows = a.size();
    std::vector<std::vector<GF>> cols(MAIN_COLS, std::vector<GF>(rows));
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) {
      GF lv[MAIN_COLS];
      auto x = u256_to_limbs16(a[r]), y = u256_to_limbs16(b[r]);
      for (int i = 0; i < 16; i++) { lv[i] = GF((u64)x[i]); lv[16 + i] = GF((u64)y[i]); }
      Arr<int64_t, 16> out;
      generate_fq_mul(x, y, lv + 32, out);
      lv[MAIN_COLS - 1] = GF::one();
      for (int c = 0; c < MAIN_COLS; c++) cols[c][r] = lv[c];
    }
    generate_split_u16_range_check(START_RC, END_RC, cols);
    return cols;
  }
};
// Fq12Stark: src/fields/fq12/mul.rs:355-517.  Row: x[12][16] y[12][16] Fq12Output(1344) filter.
struct Fq12MulAir : AirBase<Fq12MulAir> {
  static constexpr int MAIN_COLS = 108 * N_LIMBS + 1;                                                   // :355
  static constexpr int START_RC = 24 * N_LIMBS, NUM_RC = 84 * N_LIMBS - 12, END_RC = START_RC + NUM_RC;  // :356-358
  size_t num_columns() const override { return MAIN_COLS + 1 + 6 * NUM_RC; }
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override { return split_range_check_pairs(MAIN_COLS, NUM_RC); }
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const {  // :423-448
    eval_split_u16_range_check(yc, lv, nv, MAIN_COLS, START_RC, END_RC);
    int cur = 0;
    auto x = read_fq12(lv, cur), y = read_fq12(lv, cur);
    Fq12Output<P> o = read_fq12_output(lv, cur);
    P filter = lv[cur++];
    assert(cur == MAIN_COLS);
    eval_fq12_mul(yc, filter, x, y, o);
  }
  // generate_trace :375-419 with caller-provided operands: x[r], y[r] = 12 coefficients each
  std::vector<std::vector<GF>> generate_trace(const std::vector<std::array<U256, 12>>& xs, const std::vector<std::array<U256, 12>>& ys) const {
    size_t rows = xs.size();
    std::vector<std::vector<GF>> cols(MAIN_COLS, std::vector<GF>(rows));
#pragma omp parallel for schedule(static)
    for (size_t r = 0; r < rows; r++) {
      std::vector<GF> lv(MAIN_COLS);
      Fq12Limbs<int64_t> x, y, out;
      for (int i = 0; i < 12; i++) {
        x[i] = u256_to_limbs16(xs[r][i]); y[i] = u256_to_limbs16(ys[r][i]);
        for (int k = 0; k < 16; k++) { lv[16 * i + k] = GF((u64)x[i][k]); lv[192 + 16 * i + k] = GF((u64)y[i][k]); }
      }
      generate_fq12_mul(x, y, lv.data() + 384, out);
      lv[MAIN_COLS - 1] = GF::one();
      for (int c = 0; c < MAIN_COLS; c++) cols[c][r] = lv[c];
    }
    generate_split_u16_range_check(START_RC, END_RC, cols);
    return cols;
  }
};

// MyStark, the lookup unit-test table: src/utils/lookup.rs:136-213.  Four columns: inputs, table, permuted inputs, permuted
// table; constraints = eval_lookups on columns 2 / 3; pairs (0, 2), (1, 3).  The reference's test (lookup.rs:215-229) proves it
// on the crate's ONLY fixed input, inputs [6, 3, 1, 1, 0, 0, 0, 0] against table 0..7 (lookup.rs:154-161): 8 rows -- the one
// workload whose trace a Rust run reproduces without any seeded generator, hence the first entry of the parity kit.  Oracle
// only: 8 rows are far below anything the device kernels are built for (the product returns SBN_ERR_UNSUPPORTED).
struct LookupAir : AirBase<LookupAir> {
  size_t num_columns() const override { return 4; }
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override { return {{0, 2}, {1, 3}}; }   // :199-204
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const { eval_lookups(yc, lv, nv, 2, 3); }   // :168-180
  std::vector<std::vector<GF>> generate_trace(const std::vector<GF>& inputs, const std::vector<GF>& table) const {   // :151-166
    std::vector<std::vector<GF>> cols(4);
    cols[0] = inputs; cols[1] = table;
    permuted_cols(inputs, table, cols[2], cols[3]);
    return cols;
  }
};

// FlagStark, the flags unit-test table: src/utils/flags.rs:379-547.  MAIN_COLS = 6 + 8 flag columns, then the rotation pulse
// (counter, witness), the io-pulse counter and (witness, pulse) for the first and last row of every 512-row block (:372-377);
// constraints :449-492: is_final - sum(output pulses), eval_flags, eval_periodic_pulse, eval_pulse; NO permutation pairs.
// The reference's test proves NUM_INPUTS = 16 random inputs (:549-565); here num_io is a parameter.
struct FlagAir : AirBase<FlagAir> {
  size_t num_io;
  std::vector<size_t> pulse_positions;
  static constexpr int MAIN_COLS = 6 + NUM_INPUT_LIMBS;
  explicit FlagAir(size_t n) : num_io(n) {
    const size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS;
    for (size_t i = 0; i < n; i++) { pulse_positions.push_back(i * rpb); pulse_positions.push_back(i * rpb + rpb - 1); }   // :411-417
  }
  size_t num_columns() const override { return MAIN_COLS + 2 + 1 + 4 * num_io; }   // :376
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override { return {}; }
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const {   // :449-492
    P output = cst<P>(0);
    for (size_t i = 1; i < 2 * num_io; i += 2) output = output + lv[get_pulse_col(MAIN_COLS + 2, (int)i)];
    yc.constraint(lv[0] - output);
    eval_flags(yc, lv, nv, 0);
    eval_periodic_pulse(yc, lv, nv, 1, MAIN_COLS, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    eval_pulse(yc, lv, nv, MAIN_COLS + 2, pulse_positions);
  }
  // generate_trace :392-440; limbs[k] = the 8 u32 limbs of input k
  std::vector<std::vector<GF>> generate_trace(const std::vector<std::array<uint32_t, NUM_INPUT_LIMBS>>& inputs) const {
    assert(inputs.size() == num_io);
    const size_t rpb = 2 * INPUT_LIMB_BITS * NUM_INPUT_LIMBS, rows = rpb * num_io;
    std::vector<std::vector<GF>> cols(MAIN_COLS, std::vector<GF>(rows));
    for (size_t k = 0; k < num_io; k++) {
      std::vector<GF> lv(MAIN_COLS, GF()), nvv(MAIN_COLS, GF());
      generate_flags_first_row(lv.data(), 0, inputs[k].data());
      for (int c = 0; c < MAIN_COLS; c++) cols[c][k * rpb] = lv[c];
      for (size_t i = 0; i + 1 < rpb; i++) {
        std::fill(nvv.begin(), nvv.end(), GF());
        generate_flags_next_row(lv.data(), nvv.data(), i, 0);
        for (int c = 0; c < MAIN_COLS; c++) cols[c][k * rpb + i + 1] = nvv[c];
        lv.swap(nvv);
      }
    }
    generate_periodic_pulse_witness(cols, 1, 2 * INPUT_LIMB_BITS, 2 * INPUT_LIMB_BITS - 2);
    generate_pulse(cols, pulse_positions);
    assert(cols.size() == num_columns());
    return cols;
  }
};

// The FlagStark of src/fields/fq12_u64/flags_u64.rs:282-420, the flags_u64 unit-test table: 6 flag columns, the io-pulse
// counter and (witness, pulse) per block boundary; 128 rows per input; constraints :338-375.
struct FlagU64Air : AirBase<FlagU64Air> {
  size_t num_io;
  std::vector<size_t> pulse_positions;
  static constexpr int MAIN_COLS = 6;
  explicit FlagU64Air(size_t n) : num_io(n) { for (size_t i = 0; i < n; i++) { pulse_positions.push_back(i * 128); pulse_positions.push_back(i * 128 + 127); } }
  size_t num_columns() const override { return MAIN_COLS + 1 + 4 * num_io; }   // :285
  size_t num_public_inputs() const override { return 0; }
  std::vector<std::pair<size_t, size_t>> permutation_pairs() const override { return {}; }
  template <class P> void eval_t(const P* lv, const P* nv, const P*, Consumer<P>& yc) const {
    P output = cst<P>(0);
    for (size_t i = 1; i < 2 * num_io; i += 2) output = output + lv[get_pulse_col(MAIN_COLS, (int)i)];
    yc.constraint(lv[0] - output);
    eval_flags_u64(yc, lv, nv, 0);
    eval_pulse(yc, lv, nv, MAIN_COLS, pulse_positions);
  }
  std::vector<std::vector<GF>> generate_trace(const std::vector<uint64_t>& inputs) const {   // :316-337
    assert(inputs.size() == num_io);
    const size_t rows = 128 * num_io;
    std::vector<std::vector<GF>> cols(MAIN_COLS, std::vector<GF>(rows));
    for (size_t k = 0; k < num_io; k++) {
      std::vector<GF> lv(MAIN_COLS, GF()), nvv(MAIN_COLS, GF());
      generate_flags_u64_first_row(lv.data(), 0, inputs[k]);
      for (int c = 0; c < MAIN_COLS; c++) cols[c][k * 128] = lv[c];
      for (size_t i = 0; i + 1 < 128; i++) {
        std::fill(nvv.begin(), nvv.end(), GF());
        generate_flags_u64_next_row(lv.data(), nvv.data(), i, 0);
        for (int c = 0; c < MAIN_COLS; c++) cols[c][k * 128 + i + 1] = nvv[c];
        lv.swap(nvv);
      }
    }
    generate_pulse(cols, pulse_positions);
    assert(cols.size() == num_columns());
    return cols;
  }
};

}  // namespace orc
