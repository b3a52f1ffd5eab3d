// Constraint evaluators of the reference's G1 tables, templated over the element type P so that
// the SAME code runs on the device over the base field (quotient evaluation, one thread per LDE
// point) and on the host over the quadratic extension (verifier at zeta) -- mirroring how the
// reference reuses `eval_packed_generic` for both (src/curves/g1/exp.rs:331-495).
//
// The reference folds constraints one by one into two Horner accumulators (acc = acc*alpha + c,
// starky ConstraintConsumer).  That is polynomial evaluation in alpha, so any exact regrouping gives
// the same field element.  Here every gadget computes a LOCAL Horner sum h over its own constraints
// with the filter factored out, and is merged as acc = acc*alpha^count + filter*h (alpha^k from a
// per-proof table).  This removes one multiply per constraint, shares the sub-expressions that
// eval_g1_add and eval_g1_double have in common (3 of 7 limb convolutions, the whole new_y block),
// evaluates eval_flags once for its two emissions (exp.rs:462,467), and replaces the
// 56*num_io public-input constraints (exp.rs:368-392) by 4 multiplies per instance.
//
// Row values are fetched through an accessor (`row.l(c)` local, `row.n(c)` next): on the device it
// reads the column-major LDE matrix directly (lanes = consecutive LDE points => coalesced), so the
// limb convolutions need no per-thread arrays -- operands stream from L1/L2.
#pragma once
#include "gl.cuh"

static constexpr int SBN_NCH = 2;              // StarkConfig.num_challenges (standard_fast_config)
// alpha^k tables are sized per table at run time: one entry per constraint of the whole stream (AIR + permutation checks:
// a constraint's weight is alpha^(number of constraints after it)), at least 1,025.
GL_HD size_t apow_len(size_t nconstraints, size_t nzs) { size_t m = nconstraints + 2 * nzs; return (m > 1024 ? m : 1024) + 1; }
static constexpr int G1EXP_MAX_IO = 512;        // largest num_io of any Exp table (Fq12ExpStark(512) = 2^18 rows)

// BN254 base-field modulus in 16-bit limbs (src/modular/modular.rs:298-309).
GL_HD u64 bn254_modulus_limb(int j) {
  constexpr u64 M[16] = {0xfd47, 0xd87c, 0x8c16, 0x3c20, 0xca8d, 0x6871, 0x6a91, 0x9781,
                         0x585d, 0x8181, 0x45b6, 0xb850, 0xa029, 0xe131, 0x4e72, 0x3064};
  return M[j];
}

// sum of products x * w.  Generic form: one field multiply and add per term.  On the device over the base field the
// products are accumulated UNREDUCED: x = x1 2^32 + x0, w = w1 2^32 + w0, three 64-bit column sums a0 += x0 w0,
// a1 += x0 w1 + x1 w0, a2 += x1 w1 (v_mad_u64_u32, each with a carry counter), one reduction mod p in value().
// 8 instructions per term instead of ~28 (16 for the multiply's product + reduction, 8 + hazard padding for the
// canonical add); the four carries live in four SGPR pairs and every carry is read three instructions after it is
// written, so the block needs no wait-state padding.  The folds of this file (constraints times powers of alpha) are
// all of this shape, and they are ~55 % of the quotient kernel's multiplies.
template <class P>
struct Acc {
  P v;
  GL_HD void clear() { v = lift<P>(0); }
  GL_HD void mac(P x, P w) { v = v + x * w; }
  GL_HD void macv(P x, P w) { v = v + x * w; }
  GL_HD void add(P x) { v = v + x; }
  GL_HD P value() const { return v; }
};
#if defined(__HIP_DEVICE_COMPILE__)
template <>
struct Acc<F> {
  u64 a0, a1, a2; u32 c0, c1, c2;
  __device__ __forceinline__ void clear() { a0 = a1 = a2 = 0; c0 = c1 = c2 = 0; }
  __device__ __forceinline__ void mac(F x, F w) {
    // w is uniform over the wave wherever this file calls mac (alpha-power tables, per-proof constants): scalar operands
    const u32 wl = (u32)__builtin_amdgcn_readfirstlane((int)(u32)w.v), wh = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(w.v >> 32));
    u64 k0, k1, k2, k3;
    asm("s_nop 1\n\t"   // the scalar operands may come straight from v_readfirstlane: two wait states before a VALU reads them
        "v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %9, %11, %13, %2\n\t"
        "v_mad_u64_u32 %1, %8, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, vcc, 0, %5, %9\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %8\n\t"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(c0), "+v"(c1), "+v"(c2), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
        : "v"((u32)x.v), "v"((u32)(x.v >> 32)), "s"(wl), "s"(wh)
        : "vcc");
  }
  __device__ __forceinline__ void macv(F x, F w) {   // the same with a weight that differs from lane to lane (vector operands, no wait states)
    u64 k0, k1, k2, k3;
    asm("v_mad_u64_u32 %0, %6, %10, %12, %0\n\t"
        "v_mad_u64_u32 %1, %7, %10, %13, %1\n\t"
        "v_mad_u64_u32 %2, %9, %11, %13, %2\n\t"
        "v_mad_u64_u32 %1, %8, %11, %12, %1\n\t"
        "v_addc_co_u32_e64 %3, vcc, 0, %3, %6\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %7\n\t"
        "v_addc_co_u32_e64 %5, vcc, 0, %5, %9\n\t"
        "v_addc_co_u32_e64 %4, vcc, 0, %4, %8\n\t"
        : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(c0), "+v"(c1), "+v"(c2), "=&s"(k0), "=&s"(k1), "=&s"(k2), "=&s"(k3)
        : "v"((u32)x.v), "v"((u32)(x.v >> 32)), "v"((u32)w.v), "v"((u32)(w.v >> 32))
        : "vcc");
  }
  __device__ __forceinline__ void add(F x) { const u64 t = a0 + x.v; c0 += t < a0 ? 1u : 0u; a0 = t; }
  __device__ __forceinline__ F value() const {
    // a0 + c0 2^64 + (a1 + c1 2^64) 2^32 + (a2 + c2 2^64) 2^64, with 2^64 = 2^32 - 1, 2^96 = -1, 2^128 = -2^32 (mod p)
    const F t32((u64)1 << 32), eps(GLEPS);
    return F::from_u64(a0) + F::from_u64(a1) * t32 + (F::from_u64(a2) + F((u64)c0)) * eps - F((u64)c1) - F((u64)c2) * t32;
  }
};
#endif

// The consumer (starky ConstraintConsumer): acc_j = sum_t c_t alpha_j^(n-1-t) over the n constraints of the evaluation, in
// emission order.  `rem` counts the constraints still to come, so the constraint being emitted has exponent rem - 1 and
// the sum is one running dot product with the alpha-power table (no Horner chain).  The caller sets the exact count with
// start(n); rem must be back at 0 afterwards (checked on the host).
template <class P>
struct Cons {
  P alpha[SBN_NCH];
  Acc<P> a[SBN_NCH];
  int rem;
  const P* apow[SBN_NCH];  // apow[j][k] = alpha_j^k, k < apow_len(...)
  P z_last, l_first, l_last;
  GL_HD void start(int n) {
    rem = n;
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) a[j].clear();
  }
  GL_HD P result(int j) const { return a[j].value(); }
  GL_HD void c(P x) {
    --rem;
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) a[j].mac(x, apow[j][rem]);
  }
  GL_HD void ct(P x) { c(x * z_last); }
  GL_HD void cf(P x) { c(x * l_first); }
  GL_HD void cl(P x) { c(x * l_last); }
  // `count` constraints whose local sums (first of them weighted alpha^(count-1)) are h[j], all times `filter`
  GL_HD void merge(const P* h, P filter, int count) {
    rem -= count;
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) a[j].mac(filter * h[j], apow[j][rem]);
  }
};

template <class P>
struct Horner2 {  // local two-challenge sum over `count` constraints, the first weighted alpha^(count-1)
  Acc<P> a[SBN_NCH];
  int rem;
  GL_HD explicit Horner2(int count) : rem(count) {
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) a[j].clear();
  }
  GL_HD void push(const Cons<P>& cs, P x) {
    --rem;
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) a[j].mac(x, cs.apow[j][rem]);
  }
  GL_HD void value(P* out) const {
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) out[j] = a[j].value();
  }
};

// ---- G1 add/double gadget columns (src/curves/g1/muladd.rs:79-94) -----------
namespace g1c {
static constexpr int AX = 0, AY = 16, BX = 32, BY = 48, GB = 64;
static constexpr int LAM = GB + 0, NX = GB + 16, NY = GB + 32;
static constexpr int Z_QA = GB + 48, Z_LO = GB + 65, Z_HI = GB + 96;
static constexpr int X_OAR = GB + 127, X_QA = GB + 143, X_LO = GB + 160, X_HI = GB + 191;
static constexpr int Y_OAR = GB + 222, Y_QA = GB + 238, Y_LO = GB + 255, Y_HI = GB + 286;
static constexpr int SGN_Z = GB + 317, SGN_X = GB + 318, SGN_Y = GB + 319;
static constexpr int END = GB + 320;  // 384
}  // namespace g1c

// ---- The modular gadgets in FACTORED form ----------------------------------------------------
// A modular gadget (modular.rs:102-153, modular_zero.rs:82-120) ends with 32 coefficient constraints
//     coef_k = sign * (quot * modulus)_k + ((x - beta) * aux)_k + [k < 16] lin_k - conv_k,        k = 0..31,
// where conv = a limb convolution such as lambda * (x2 - x1) (pol_mul_wide, 16 x 16 limbs, 31 coefficients).  The
// consumer folds them as sum_k alpha^(31-k) coef_k, and a Horner-weighted sum of CONVOLUTION coefficients factors:
//     sum_{k<=30} alpha^(30-k) sum_{i+j=k} a_i b_j = (sum_i a_i alpha^(15-i)) * (sum_j b_j alpha^(15-j)) = rev(a) * rev(b),
// rev(v) being the Horner fold of the limbs of v in column order.  Likewise the quotient part is rev17(quot) * rev16(m)
// and the (x - beta) * aux part is (1 - 2^16 alpha) * rev31(aux).  So the 32 constraints cost a few dot products with
// powers of alpha instead of 16 x 16 limb products per convolution -- the SAME field element (exact arithmetic, only
// distributivity is used; no inverse of alpha, so alpha = 0 is fine too), 7 convolutions x 256 multiplies -> ~100 for the
// G1 gadget, 2 x 144 x 256 -> ~600 for Fq12, and every limb column is read once instead of once per
// coefficient it contributes to (the quotient kernel's first segment moved 3.0 GB for 0.45 GB of columns before).
// tests/test_product_host.py::test_regrouped_constraints_equal_the_oracle_on_random_rows compares with the oracle's
// constraint-by-constraint fold on random rows.
template <class P, class Row>
GL_HD P rev_limbs(const Cons<P>& cs, const Row& row, int j, int col, int n) {   // sum_i row[col + i] alpha_j^(n-1-i)
  // a dot product with the alpha-power table (uniform index: scalar loads), not a Horner chain: the products are independent
  const P* ap = cs.apow[j];
  Acc<P> h; h.clear();
  h.add(row.l(col + n - 1));
#pragma unroll 4   // (the loads of four limbs in flight: the evaluators are bound by load latency, not by issue)
  for (int i = 0; i < n - 1; i++) h.mac(row.l(col + i), ap[n - 1 - i]);
  const P r = h.value();
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);   // keep the folds of different limb vectors apart: interleaving them all costs the G1 kernel 190 spilled registers
#endif
  return r;
}
template <class P>
GL_HD P rev_modulus(const Cons<P>& cs, int j) {   // sum_i m_i alpha_j^(15-i), m = the BN254 base-field modulus in 16-bit limbs
  P h = lift<P>(bn254_modulus_limb(0));
  for (int i = 1; i < 16; i++) h = h * cs.alpha[j] + lift<P>(bn254_modulus_limb(i));
  return h;
}
// Both challenges from ONE pass over the limbs: the evaluators wait for their column loads, so every limb is loaded once
// and feeds two accumulators (the head kernel of G1ExpStark(128): 1.30 -> 0.69 ms against one pass per challenge).
template <class P, class Row>
GL_HD void rev_limbs2(const Cons<P>& cs, const Row& row, int col, int n, P* out) {
  Acc<P> h[SBN_NCH];
  const P last = row.l(col + n - 1);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) { h[j].clear(); h[j].add(last); }
#pragma unroll 16
  for (int i = 0; i < n - 1; i++) {
    const P x = row.l(col + i);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) h[j].mac(x, cs.apow[j][n - 1 - i]);
  }
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) out[j] = h[j].value();
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);   // keep the folds of different limb vectors apart: interleaved, their loads in flight spill registers
#endif
}
// Weighted sums over the 32 coefficient constraints of  sign * (quot * modulus)_k + ((x - beta) * aux)_k :
// quot_abs at qa_col (17 limbs), aux_lo / aux_hi (31 limbs each; aux = lo - 2^29 + 2^16 hi, modular.rs:118-121);
// rev31(aux) = rev31(lo) + 2^16 rev31(hi) - 2^29 (1 + alpha + ... + alpha^30).
template <class P, class Row>
GL_HD void modop_tail2(const Cons<P>& cs, const Row& row, int qa_col, int lo_col, int hi_col, P sign, const P* mrev, P* out) {
  const P base = lift<P>(65536), off = lift<P>(1ULL << 29), one = lift<P>(1);
  P q[SBN_NCH];
  rev_limbs2(cs, row, qa_col, 17, q);
  Acc<P> ulo[SBN_NCH], uhi[SBN_NCH];
  P geo[SBN_NCH];
  const P l30 = row.l(lo_col + 30), h30 = row.l(hi_col + 30);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) { ulo[j].clear(); uhi[j].clear(); ulo[j].add(l30); uhi[j].add(h30); geo[j] = one; }
#pragma unroll 10
  for (int k = 0; k < 30; k++) {
    const P xl = row.l(lo_col + k), xh = row.l(hi_col + k);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) { const P w = cs.apow[j][30 - k]; ulo[j].mac(xl, w); uhi[j].mac(xh, w); geo[j] = geo[j] + w; }
  }
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) {
    const P u = ulo[j].value() + base * uhi[j].value() - off * geo[j];
    out[j] = sign * q[j] * mrev[j] + (one - base * cs.alpha[j]) * u;
  }
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);   // keep the folds of different limb vectors apart: interleaved, their loads in flight spill registers
#endif
}
// Local sums of the 34 leading constraints of eval_modular_op: eval_packed_generic_addcy
// (src/modular/addcy.rs:16-58) for `modulus + out_aux_red = output + 2^256` (16 carry constraints, exponents 33..18;
// given_cy[0]*(given_cy[0]-1) = 0 with given_cy[0] = 1; cy == given_cy[0], exponent 16; given_cy[1..16] = 0), then
// quot_sign^2 = 1 (modular.rs:123), exponent 0.
template <class P, class Row>
GL_HD void modop_prefix2(const Cons<P>& cs, const Row& row, int oar_col, int out_col, P sign, P* out) {
  const P overflow = lift<P>(65536), overflow_inv = lift<P>(18446462594437939201ULL), one = lift<P>(1);
  Acc<P> h[SBN_NCH];
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) h[j].clear();
  P cy = lift<P>(0);
#pragma unroll 8
  for (int i = 0; i < 16; i++) {
    const P t = cy + lift<P>(bn254_modulus_limb(i)) + row.l(oar_col + i) - row.l(out_col + i);
    const P c = t * (overflow - t);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) h[j].mac(c, cs.apow[j][33 - i]);
    cy = t * overflow_inv;
  }
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) { h[j].mac(cy - one, cs.apow[j][16]); h[j].add(sign * sign - one); out[j] = h[j].value(); }
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);   // keep the folds of different limb vectors apart: interleaved, their loads in flight spill registers
#endif
}

// Both eval_g1_add (muladd.rs:179-230) and eval_g1_double (:291-342) on the same row.
// h_add / h_dbl receive the 165-constraint local sums (filters NOT applied).
template <class P, class Row>
GL_HD void g1_gadget(const Cons<P>& cs, const Row& row, P* h_add, P* h_dbl) {
  using namespace g1c;
  const P one = lift<P>(1), two = lift<P>(2), three = lift<P>(3);
  const P sz = row.l(SGN_Z), sx = row.l(SGN_X), sy = row.l(SGN_Y);
  // The gadget emits zero (33 constraints), x (66), y (66): local sums weighted alpha^(132, 66, 0).  Everything that enters
  // them LINEARLY (modular-reduction sides, outputs, operands outside products) goes straight into `lin` (+ the add-only and
  // double-only parts), so that only five evaluations stay live for the three products.
  // zero_pol = lambda*dx - dy (add) / 2*lambda*y - 3*x^2 (double); x: lambda^2 - (x1 + x2); y: lambda*(x1 - new_x) - y1
  P mrev[SBN_NCH], lin[SBN_NCH], t[SBN_NCH], u[SBN_NCH], L[SBN_NCH], ax[SBN_NCH], ay[SBN_NCH], bx[SBN_NCH], nx[SBN_NCH];
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) mrev[j] = rev_modulus(cs, j);
  modop_tail2(cs, row, Z_QA, Z_LO, Z_HI, sz, mrev, t);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = ((sz * sz - one) * cs.apow[j][32] + t[j]) * cs.apow[j][132];   // quot_sign_zero^2 - 1 (modular_zero.rs:91) first
  modop_tail2(cs, row, X_QA, X_LO, X_HI, sx, mrev, t);
  modop_prefix2(cs, row, X_OAR, NX, sx, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = lin[j] + (u[j] * cs.apow[j][32] + t[j]) * cs.apow[j][66];
  modop_tail2(cs, row, Y_QA, Y_LO, Y_HI, sy, mrev, t);
  modop_prefix2(cs, row, Y_OAR, NY, sy, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = lin[j] + u[j] * cs.apow[j][32] + t[j];
  rev_limbs2(cs, row, LAM, 16, L); rev_limbs2(cs, row, AX, 16, ax); rev_limbs2(cs, row, AY, 16, ay); rev_limbs2(cs, row, BX, 16, bx);
  rev_limbs2(cs, row, NX, 16, nx); rev_limbs2(cs, row, BY, 16, t); rev_limbs2(cs, row, NY, 16, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) {
    const P al = cs.alpha[j], a16 = cs.apow[j][16], a66 = cs.apow[j][66], a132 = cs.apow[j][132];
    const P common = lin[j] + a16 * ((nx[j] + ax[j]) * a66 + u[j] + ay[j]) - al * ((L[j] * L[j]) * a66 + L[j] * (ax[j] - nx[j]));
    h_add[j] = common + a16 * ((t[j] - ay[j]) * a132 + bx[j] * a66) - al * ((L[j] * (bx[j] - ax[j])) * a132);
    h_dbl[j] = common + a16 * (ax[j] * a66) - al * ((two * (L[j] * ay[j]) - three * (ax[j] * ax[j])) * a132);
  }
}
static constexpr int G1_GADGET_CONSTRAINTS = 165;

// eval_flags (src/utils/flags.rs:136-195): 26 constraints, emitted into a local sum.
template <class P, class Row>
GL_HD void flags_block(const Cons<P>& cs, const Row& row, int s, Horner2<P>& h) {
  const P one = lift<P>(1);
  const int fin = s, rot = s + 1, a = s + 2, b = s + 3, fb = s + 4, bitc = s + 5, sl = s + 6, el = sl + 8;
  P la = row.l(a), lb = row.l(b), bit = row.l(bitc), lrot = row.l(rot), lfin = row.l(fin);
  h.push(cs, la * cs.l_first);
  h.push(cs, (lb - one) * cs.l_first);
  h.push(cs, bit * bit - bit);
  h.push(cs, bit * lb - row.l(fb));
  h.push(cs, lrot * la);
  h.push(cs, lfin * lrot);
  h.push(cs, (la + row.n(a) - one) * cs.z_last);
  h.push(cs, (lb + row.n(b) - one) * cs.z_last);
  P first_limb = row.l(sl), next_first_limb = row.n(sl), next_bit = row.n(bitc);
  P is_not_final = one - lfin, is_not_split = one - la;
  P nrf = one - lrot - lfin;
  h.push(cs, is_not_final * la * (first_limb - (next_first_limb + next_first_limb) - next_bit) * cs.z_last);
  h.push(cs, is_not_split * (next_bit - bit) * cs.z_last);
  h.push(cs, nrf * is_not_split * (first_limb - next_first_limb) * cs.z_last);
  for (int c = sl + 1; c < el; c++) h.push(cs, lrot * (row.n(c - 1) - row.l(c)) * cs.z_last);
  h.push(cs, lrot * row.n(el - 1) * cs.z_last);
  for (int c = sl + 1; c < el; c++) h.push(cs, nrf * (row.n(c) - row.l(c)) * cs.z_last);
}
static constexpr int FLAGS_CONSTRAINTS = 26;
// eval_flags_u64 (src/fields/fq12_u64/flags_u64.rs:96-139): 9 constraints; columns is_final, a, b, filtered_bit, bit, val.
template <class P, class Row>
GL_HD void flags_u64_block(const Cons<P>& cs, const Row& row, int s, Horner2<P>& h) {
  const P one = lift<P>(1);
  const int fin = s, a = s + 1, b = s + 2, fb = s + 3, bitc = s + 4, val = s + 5;
  P la = row.l(a), lb = row.l(b), bit = row.l(bitc), lfin = row.l(fin);
  h.push(cs, la * cs.l_first);
  h.push(cs, (lb - one) * cs.l_first);
  h.push(cs, bit * bit - bit);
  h.push(cs, bit * lb - row.l(fb));
  h.push(cs, (la + row.n(a) - one) * cs.z_last);
  h.push(cs, (lb + row.n(b) - one) * cs.z_last);
  P first_limb = row.l(val), next_first_limb = row.n(val), next_bit = row.n(bitc);
  P is_not_final = one - lfin, is_not_split = one - la;
  h.push(cs, is_not_final * la * (first_limb - (next_first_limb + next_first_limb) - next_bit) * cs.z_last);
  h.push(cs, is_not_split * (next_bit - bit) * cs.z_last);
  h.push(cs, is_not_final * is_not_split * (first_limb - next_first_limb) * cs.z_last);
}
static constexpr int FLAGS_U64_CONSTRAINTS = 9;

// eval_lookups (src/utils/lookup.rs:13-34).
template <class P, class Row>
GL_HD void lookup_pair(Cons<P>& cs, const Row& row, int col_in, int col_tab) {
  P nin = row.n(col_in);
  P d_table = nin - row.n(col_tab);
  cs.c((nin - row.l(col_in)) * d_table);
  cs.cl(d_table);
}
template <class P, class Row>
GL_HD void range_table_block(Cons<P>& cs, const Row& row, int table_col, u64 range_max_minus_1) {
  P cur = row.l(table_col), incr = row.n(table_col) - cur;
  cs.cf(cur);
  cs.ct(incr * incr - incr);
  cs.cl(cur - lift<P>(range_max_minus_1));
}

// ---- G1Stark (src/curves/g1/muladd.rs:462-624) --------------------------------------------------
struct G1OpShape {
  static constexpr int MAIN_COLS = 24 * 16 + 2;                 // 386
  static constexpr int START_RC = 64, NUM_RC = 20 * 16 - 4;     // 316 targets
  static constexpr int NUM_COLS = MAIN_COLS + 1 + 6 * NUM_RC;   // 2283
  static constexpr int NUM_PAIRS = 4 * NUM_RC;                  // 1264
  static constexpr int NUM_CONSTRAINTS = NUM_RC + 4 * NUM_RC + 3 + 2 * 165;
  // split_u16_range_check_pairs (range_check.rs:228-246)
  GL_HD static void pair(int z, int& lhs, int& rhs) {
    int t = z >> 2, w = z & 3, i = MAIN_COLS + 1 + 6 * t;
    if (w == 0) { lhs = MAIN_COLS; rhs = i + 2; }
    else if (w == 1) { lhs = MAIN_COLS; rhs = i + 5; }
    else if (w == 2) { lhs = i; rhs = i + 1; }
    else { lhs = i + 3; rhs = i + 4; }
  }
};
template <class P, class Row>
GL_HD void g1op_eval(Cons<P>& cs, const Row& row) {
  typedef G1OpShape S;
  // eval_split_u16_range_check (range_check.rs:162-192)
  const P c256 = lift<P>(256);
  for (int i = 0; i < S::NUM_RC; i++) {
    P lo = row.l(S::MAIN_COLS + 1 + 6 * i), hi = row.l(S::MAIN_COLS + 4 + 6 * i);
    cs.c(row.l(S::START_RC + i) - (lo + hi * c256));
  }
  for (int i = S::MAIN_COLS + 1; i < S::MAIN_COLS + 1 + 6 * S::NUM_RC; i += 6) {
    lookup_pair(cs, row, i + 1, i + 2);
    lookup_pair(cs, row, i + 4, i + 5);
  }
  range_table_block(cs, row, S::MAIN_COLS, 255);
  P h_add[SBN_NCH], h_dbl[SBN_NCH];
  g1_gadget(cs, row, h_add, h_dbl);
  cs.merge(h_add, row.l(S::MAIN_COLS - 2), G1_GADGET_CONSTRAINTS);  // is_add
  cs.merge(h_dbl, row.l(S::MAIN_COLS - 1), G1_GADGET_CONSTRAINTS);  // is_double
}

// ---- G2 add/double gadget (src/curves/g2/muladd.rs:56-80 column order; 640 columns) --------------------
namespace g2c {
static constexpr int AX = 0, AY = 32, BX = 64, BY = 96, GB = 128;  // each Fq2 = c0[16] c1[16]
static constexpr int LAM = GB, NX = GB + 32, NY = GB + 64;
GL_HD int z_qa(int c) { return GB + 96 + 79 * c; }    // aux_zeros[c]: quot_abs(17) lo(31) hi(31)
GL_HD int x_base(int c) { return GB + 254 + 95 * c; } // auxs[c]   : out_aux_red(16) quot_abs(17) lo(31) hi(31)
GL_HD int y_base(int c) { return GB + 444 + 95 * c; } // auxs[2+c]
static constexpr int SGN_Z = GB + 634, SGN_X = GB + 636, SGN_Y = GB + 638;
}  // namespace g2c

// eval_g2_add (muladd.rs:416-472) and eval_g2_double (:203-261) on one row: 330-constraint local sums.
// Fq2 limb products (fq2.rs:41-58): (X*Y).c0 = X0*Y0 - X1*Y1, (X*Y).c1 = X0*Y1 + X1*Y0.
template <class P>
struct G2Comp {   // the reversed evaluations of one Fq2 component (c0 or c1) that enter products, for both challenges
  P L[SBN_NCH], ax[SBN_NCH], ay[SBN_NCH], bx[SBN_NCH], nx[SBN_NCH];
};
// Everything of component c that enters the 330 constraints LINEARLY (modular-reduction sides, outputs, the operands
// outside products) goes straight into three running sums per challenge -- common to add and double, add only, double only --
// so that only five evaluations per component stay live for the Fq2 products.  Weights: the gadget emits zero[c0], zero[c1]
// (33 each), x[c0], x[c1], y[c0], y[c1] (66 each), so their local sums carry alpha^(297, 264, 198, 132, 66, 0).
template <class P, class Row>
GL_HD void g2_component(const Cons<P>& cs, const Row& row, int c, const P* mrev, G2Comp<P>& o, P* lin, P* lin_add, P* lin_dbl) {
  using namespace g2c;
  const P one = lift<P>(1);
  const P sz = row.l(SGN_Z + c), sx = row.l(SGN_X + c), sy = row.l(SGN_Y + c);
  const int wz = c ? 264 : 297, wx = c ? 132 : 198, wy = c ? 0 : 66;
  P t[SBN_NCH], u[SBN_NCH];
  modop_tail2(cs, row, z_qa(c), z_qa(c) + 17, z_qa(c) + 48, sz, mrev, t);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = lin[j] + ((sz * sz - one) * cs.apow[j][32] + t[j]) * cs.apow[j][wz];   // quot_sign_zero^2 - 1 (modular_zero.rs:91) first
  modop_tail2(cs, row, x_base(c) + 16, x_base(c) + 33, x_base(c) + 64, sx, mrev, t);
  modop_prefix2(cs, row, x_base(c), NX + 16 * c, sx, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = lin[j] + (u[j] * cs.apow[j][32] + t[j]) * cs.apow[j][wx];
  modop_tail2(cs, row, y_base(c) + 16, y_base(c) + 33, y_base(c) + 64, sy, mrev, t);
  modop_prefix2(cs, row, y_base(c), NY + 16 * c, sy, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) lin[j] = lin[j] + (u[j] * cs.apow[j][32] + t[j]) * cs.apow[j][wy];
  rev_limbs2(cs, row, LAM + 16 * c, 16, o.L); rev_limbs2(cs, row, AX + 16 * c, 16, o.ax); rev_limbs2(cs, row, AY + 16 * c, 16, o.ay);
  rev_limbs2(cs, row, BX + 16 * c, 16, o.bx); rev_limbs2(cs, row, NX + 16 * c, 16, o.nx);
  rev_limbs2(cs, row, BY + 16 * c, 16, t); rev_limbs2(cs, row, NY + 16 * c, 16, u);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) {
    const P a16 = cs.apow[j][16];
    // zero (add): + a16 (y2 - y1);  x: + a16 (new_x + x1 + [x2 | x1]);  y: + a16 (new_y + y1)
    lin[j] = lin[j] + a16 * ((o.nx[j] + o.ax[j]) * cs.apow[j][wx] + (u[j] + o.ay[j]) * cs.apow[j][wy]);
    lin_add[j] = lin_add[j] + a16 * ((t[j] - o.ay[j]) * cs.apow[j][wz] + o.bx[j] * cs.apow[j][wx]);
    lin_dbl[j] = lin_dbl[j] + a16 * (o.ax[j] * cs.apow[j][wx]);
  }
}
template <class P, class Row>
GL_HD void g2_gadget(const Cons<P>& cs, const Row& row, P* h_add, P* h_dbl) {
  const P two = lift<P>(2), three = lift<P>(3), zero = lift<P>(0);
  P mrev[SBN_NCH], lin[SBN_NCH], lin_add[SBN_NCH], lin_dbl[SBN_NCH];
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) { mrev[j] = rev_modulus(cs, j); lin[j] = zero; lin_add[j] = zero; lin_dbl[j] = zero; }
  G2Comp<P> p0, p1;   // (two named objects, not an array: a component index that is not a constant puts them in scratch memory)
  g2_component(cs, row, 0, mrev, p0, lin, lin_add, lin_dbl);
  g2_component(cs, row, 1, mrev, p1, lin, lin_add, lin_dbl);
#pragma unroll   // (j must be a compile-time index: a run-time one puts the consumer in scratch memory and its tables behind flat loads)
  for (int j = 0; j < SBN_NCH; j++) {
    // Fq2 limb products (fq2.rs:41-58) at the level of the reversed evaluations
    const P L0 = p0.L[j], L1 = p1.L[j], x0 = p0.ax[j], x1 = p1.ax[j], y0 = p0.ay[j], y1 = p1.ay[j];
    const P d0 = p0.bx[j] - x0, d1 = p1.bx[j] - x1, e0 = x0 - p0.nx[j], e1 = x1 - p1.nx[j];
    const P x01 = x0 * x1, l01 = L0 * L1;
    const P c1a = L0 * d0 - L1 * d1, c1b = L0 * d1 + L1 * d0;        // lambda * delta_x
    const P c2a = L0 * y0 - L1 * y1, c2b = L0 * y1 + L1 * y0;        // lambda * y1
    const P c3a = x0 * x0 - x1 * x1, c3b = x01 + x01;                // x1^2
    const P c4a = L0 * L0 - L1 * L1, c4b = l01 + l01;                // lambda^2
    const P c5a = L0 * e0 - L1 * e1, c5b = L0 * e1 + L1 * e0;        // lambda * (x1 - new_x)
    const P* ap = cs.apow[j];
    const P common = lin[j] - cs.alpha[j] * (c4a * ap[198] + c4b * ap[132] + c5a * ap[66] + c5b);
    h_add[j] = common + lin_add[j] - cs.alpha[j] * (c1a * ap[297] + c1b * ap[264]);
    h_dbl[j] = common + lin_dbl[j] - cs.alpha[j] * ((two * c2a - three * c3a) * ap[297] + (two * c2b - three * c3b) * ap[264]);
  }
}

// ---- Fq12 multiplication gadget (src/fields/fq12/mul.rs) ------------------------------------------------------
// Columns relative to the row: a[12][16] at 0, b[12][16] at 192, Fq12Output at 384 = output[12][16],
// auxs[12] (out_aux_red 16, quot_abs 17, lo 31, hi 31), quot_signs[12]   (mul.rs:217-231).
namespace f12c {
static constexpr int A = 0, B = 192, OUT = 384, AUX = 576, SGN = 576 + 12 * 95;
}
// eval_fq12_mul(is_sq, a, a) and eval_fq12_mul(is_mul, a, b) on one row (fq12/exp.rs:396-397): two
// 792-constraint local sums.  pol_mul_fq12 (mul.rs:24-87): with D[m] = sum_{i+j=m} (a_i b_j - a_{i+6} b_{j+6}),
// S[m] = sum_{i+j=m} (a_i b_{j+6} + a_{i+6} b_j):  real[m] = D[m] + 9 D[m+6] - S[m+6],
// imag[m] = S[m] + D[m+6] + 9 S[m+6]  (m < 5; the m = 5 terms have no wrap-around part).
// Coefficients m and m+6 are processed together so that every limb product is formed exactly once.
// The 2 x 144 coefficient products of the square and of the product for challenge J.  J and the coefficient index M are
// template parameters: the body is too large for the compiler to unroll loops over them ("unrolled size is too large"), a
// run-time j puts the consumer in scratch memory behind flat loads and a run-time m does the same to ar / br.
template <int J, int M, class P>
GL_HD void fq12_coeff(const Cons<P>& cs, const P (&ar)[12], const P (&br)[12], P& hs, P& hm) {
  // pol_mul_fq12 (mul.rs:24-87) on the evaluations: D[t] = sum_{i+j=t} (x_i y_j - x_{i+6} y_{j+6}), S[t] = sum (x_i y_{j+6} + x_{i+6} y_j)
  const P zero = lift<P>(0), nine = lift<P>(9);
  P ds = zero, ss = zero, ds6 = zero, ss6 = zero, dm = zero, sm = zero, dm6 = zero, sm6 = zero;
#pragma unroll
  for (int i = 0; i < 6; i++) {
    const bool wrap = i > M;
    if (!(wrap && M == 5)) {                   // i + j = 11 has no term with j < 6
      const int t = wrap ? M + 6 - i : M - i;  // i + t = M  or  i + t = M + 6
      const P d_s = ar[i] * ar[t] - ar[i + 6] * ar[t + 6], s_s = ar[i] * ar[t + 6] + ar[i + 6] * ar[t];
      const P d_m = ar[i] * br[t] - ar[i + 6] * br[t + 6], s_m = ar[i] * br[t + 6] + ar[i + 6] * br[t];
      if (wrap) { ds6 += d_s; ss6 += s_s; dm6 += d_m; sm6 += s_m; }
      else { ds += d_s; ss += s_s; dm += d_m; sm += s_m; }
    }
  }
  const P w0 = cs.apow[J][66 * (11 - M)], w1 = cs.apow[J][66 * (5 - M)];
  hs = hs + (ds + nine * ds6 - ss6) * w0 + (ss + ds6 + nine * ss6) * w1;
  hm = hm + (dm + nine * dm6 - sm6) * w0 + (sm + dm6 + nine * sm6) * w1;
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);   // one coefficient at a time
#endif
}
template <int J, class P, class Row>
GL_HD void fq12_products(const Cons<P>& cs, const Row& row, P& hs, P& hm) {
  using namespace f12c;
  P ar[12], br[12];   // reversed evaluations of the 12 + 12 coefficient limb vectors
#pragma unroll   // (the compiler does not unroll this one fully -- "unrolled size is too large" -- so ar / br live in scratch memory:
                 // 24 stores, and the straight-line alternative spills 700 registers)
  for (int i = 0; i < 12; i++) { ar[i] = rev_limbs(cs, row, J, A + 16 * i, 16); br[i] = rev_limbs(cs, row, J, B + 16 * i, 16); }
  hs = lift<P>(0); hm = lift<P>(0);
  fq12_coeff<J, 0>(cs, ar, br, hs, hm); fq12_coeff<J, 1>(cs, ar, br, hs, hm); fq12_coeff<J, 2>(cs, ar, br, hs, hm);
  fq12_coeff<J, 3>(cs, ar, br, hs, hm); fq12_coeff<J, 4>(cs, ar, br, hs, hm); fq12_coeff<J, 5>(cs, ar, br, hs, hm);
}
template <class P, class Row>
GL_HD void fq12_gadget(const Cons<P>& cs, const Row& row, P* h_sq, P* h_mul) {
  using namespace f12c;
  static_assert(SBN_NCH == 2, "fq12_products is instantiated for two challenges");
  const P zero = lift<P>(0);
  // The modular-reduction side (out_aux_red, output, quot, aux: 111 of the 143 limbs per coefficient) is the same for the
  // square and the product and enters both sums with the weight alpha^(66 (11 - c)); every one of its limbs is loaded once
  // and feeds both challenges.
  P mrev[SBN_NCH], hpre[SBN_NCH];
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) { mrev[j] = rev_modulus(cs, j); hpre[j] = zero; }
  for (int c = 0; c < 12; c++) {
    const int ab = AUX + 95 * c;
    const P sgn = row.l(SGN + c);
    P pp[SBN_NCH], tt[SBN_NCH], oo[SBN_NCH];
    modop_prefix2(cs, row, ab, OUT + 16 * c, sgn, pp);
    modop_tail2(cs, row, ab + 16, ab + 33, ab + 64, sgn, mrev, tt);
    rev_limbs2(cs, row, OUT + 16 * c, 16, oo);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) hpre[j] = hpre[j] + (pp[j] * cs.apow[j][32] + tt[j] + cs.apow[j][16] * oo[j]) * cs.apow[j][66 * (11 - c)];
  }
  P hs0, hm0, hs1, hm1;
  fq12_products<0>(cs, row, hs0, hm0);
  fq12_products<1>(cs, row, hs1, hm1);
  h_sq[0] = hpre[0] - cs.alpha[0] * hs0; h_mul[0] = hpre[0] - cs.alpha[0] * hm0;
  h_sq[1] = hpre[1] - cs.alpha[1] * hs1; h_mul[1] = hpre[1] - cs.alpha[1] * hm1;
}

// ---- Fq multiplication gadget (src/fields/fq/mul.rs) -------------------------------------------------------------
// Columns relative to the row: a[16] at 0, b[16] at 16, FqOutput at 32 = output[16], aux (out_aux_red 16, quot_abs 17,
// lo 31, hi 31), quot_sign (mul.rs:49-54).  eval_fq_mul(is_sq, a, a) and eval_fq_mul(is_mul, a, b) on one row
// (fq/exp.rs:362-363): two 66-constraint local sums sharing the modular-reduction side.
template <class P, class Row>
GL_HD void fq_gadget(const Cons<P>& cs, const Row& row, P* h_sq, P* h_mul) {
  constexpr int A = 0, B = 16, OUT = 32, AUX = 48, SGN = 143;
  const P sgn = row.l(SGN);
  P mrev[SBN_NCH], a[SBN_NCH], b[SBN_NCH], pp[SBN_NCH], tt[SBN_NCH], oo[SBN_NCH];
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) mrev[j] = rev_modulus(cs, j);
  rev_limbs2(cs, row, A, 16, a); rev_limbs2(cs, row, B, 16, b); rev_limbs2(cs, row, OUT, 16, oo);
  modop_prefix2(cs, row, AUX, OUT, sgn, pp);
  modop_tail2(cs, row, AUX + 16, AUX + 33, AUX + 64, sgn, mrev, tt);
#pragma unroll   // (j must be a compile-time index: a run-time one puts the consumer in scratch memory and its tables behind flat loads)
  for (int j = 0; j < SBN_NCH; j++) {
    const P pre = pp[j] * cs.apow[j][32] + tt[j] + cs.apow[j][16] * oo[j];
    h_sq[j] = pre - cs.alpha[j] * (a[j] * a[j]);
    h_mul[j] = pre - cs.alpha[j] * (a[j] * b[j]);
  }
}

// ---- ModularStark (src/modular/modular.rs:361-537) and Fq12Stark (src/fields/fq12/mul.rs:355-517) ----------------------------
// The reference's own test tables for the modular gadget and the Fq12 product: one operation per row behind a filter column
// (the last main column), split range check over output + aux columns, constraints = the range check, then the gadget --
// the shape of G1Stark above.  Their rows are laid out as fq_gadget / fq12_gadget expect (a, b, output, aux.., signs).
struct OpShape {
  int kind;        // 7 = ModularStark (a * b mod p), 8 = Fq12Stark (x * y in Fq12): the sbn_air_kind values
  int main_cols, start_rc, num_rc, gadget_cons;
  GL_HD explicit OpShape(int k) : kind(k) {
    if (k == 8) { main_cols = 108 * 16 + 1; start_rc = 24 * 16; num_rc = 84 * 16 - 12; gadget_cons = 12 * 66; }   // mul.rs:355-358
    else { main_cols = 9 * 16 + 1; start_rc = 2 * 16; num_rc = 7 * 16 - 1; gadget_cons = 66; }                     // modular.rs:361-366
  }
  GL_HD int num_cols() const { return main_cols + 1 + 6 * num_rc; }
  GL_HD int num_pairs() const { return 4 * num_rc; }
  GL_HD int num_constraints() const { return num_rc + 4 * num_rc + 3 + gadget_cons; }
  // split_u16_range_check_pairs (range_check.rs:228-246)
  GL_HD void pair(int z, int& lhs, int& rhs) const {
    int t = z >> 2, w = z & 3, i = main_cols + 1 + 6 * t;
    if (w == 0) { lhs = main_cols; rhs = i + 2; }
    else if (w == 1) { lhs = main_cols; rhs = i + 5; }
    else if (w == 2) { lhs = i; rhs = i + 1; }
    else { lhs = i + 3; rhs = i + 4; }
  }
};
template <int KIND, class P, class Row>
GL_HD void op_eval(Cons<P>& cs, const Row& row, const OpShape& S) {
  // eval_split_u16_range_check (range_check.rs:162-192)
  const P c256 = lift<P>(256);
#pragma unroll 4
  for (int i = 0; i < S.num_rc; i++) {
    P lo = row.l(S.main_cols + 1 + 6 * i), hi = row.l(S.main_cols + 4 + 6 * i);
    cs.c(row.l(S.start_rc + i) - (lo + hi * c256));
  }
#pragma unroll 2
  for (int i = S.main_cols + 1; i < S.main_cols + 1 + 6 * S.num_rc; i += 6) {
    lookup_pair(cs, row, i + 1, i + 2);
    lookup_pair(cs, row, i + 4, i + 5);
  }
  range_table_block(cs, row, S.main_cols, 255);
  // eval_modular_op(filter, input0 * input1, ..) (modular.rs:465-476) / eval_fq12_mul(filter, x, y, output) (mul.rs:447): the
  // product sums of the gadgets; their square sums are not needed here
  P h_sq[SBN_NCH], h_mul[SBN_NCH];
  if (KIND == 8) fq12_gadget(cs, row, h_sq, h_mul); else fq_gadget(cs, row, h_sq, h_mul);
  cs.merge(h_mul, row.l(S.main_cols - 1), S.gadget_cons);
}

// ---- MyStark (src/utils/lookup.rs:136-213) and FlagStark (src/utils/flags.rs:379-547): the reference's unit-test tables -------
// MyStark: columns inputs, table, permuted inputs, permuted table; constraints = eval_lookups(2, 3); pairs (0, 2), (1, 3).
// The smallest table of the reference (4 columns: Merkle leaves are copied rows, hash_or_noop) -- here at any power-of-two
// height >= 512 (the reference's own 8-row instance lives in the oracle only, tests/golden/parity_kit).
struct LookupShape {
  static constexpr int NUM_COLS = 4, NUM_PAIRS = 2, NUM_CONSTRAINTS = 2;
  GL_HD void pair(int z, int& lhs, int& rhs) const { lhs = z; rhs = z + 2; }   // lookup.rs:199-204
};
template <class P, class Row>
GL_HD void lookup_eval(Cons<P>& cs, const Row& row) { lookup_pair(cs, row, 2, 3); }   // lookup.rs:168-180
// FlagStark(num_io): the bit-peeling flags of the Exp tables on their own -- 14 flag columns, the rotation pulse (counter,
// witness), the io-pulse counter and (witness, pulse) per block boundary; 512 rows per input; NO permutation pairs, so the
// proof has no permutation-Z commitment at all (starky: uses_permutation_args() == false).
struct FlagShape {
  int num_io;
  GL_HD explicit FlagShape(int n) : num_io(n) {}
  static constexpr int MAIN_COLS = 14, START_PERIODIC = 14, START_IO_PULSES = 16;   // flags.rs:372-377: 6 + NUM_INPUT_LIMBS, + 2, + 1 + 4 inputs
  GL_HD int num_cols() const { return MAIN_COLS + 2 + 1 + 4 * num_io; }
  GL_HD int num_constraints() const { return 1 + FLAGS_CONSTRAINTS + 5 + 2 + 4 * num_io; }
  GL_HD int witness_col(int i) const { return START_IO_PULSES + 1 + 2 * i; }   // pulse.rs:14
  GL_HD int pulse_col(int i) const { return START_IO_PULSES + 2 + 2 * i; }     // pulse.rs:10
};
// The u64 variant (src/fields/fq12_u64/flags_u64.rs:282-377): 6 flag columns, 128 rows per input, no rotation pulse.
struct FlagU64Shape {
  int num_io;
  GL_HD explicit FlagU64Shape(int n) : num_io(n) {}
  static constexpr int MAIN_COLS = 6;
  GL_HD int num_cols() const { return MAIN_COLS + 1 + 4 * num_io; }                              // flags_u64.rs:285
  GL_HD int num_constraints() const { return 1 + FLAGS_U64_CONSTRAINTS + 2 + 4 * num_io; }
  GL_HD int witness_col(int i) const { return MAIN_COLS + 1 + 2 * i; }
  GL_HD int pulse_col(int i) const { return MAIN_COLS + 2 + 2 * i; }
};
template <class P, class Row>
GL_HD void flag_u64_eval(Cons<P>& cs, const Row& row, const FlagU64Shape& sh) {   // flags_u64.rs:338-375
  const P one = lift<P>(1);
  P sum_out = lift<P>(0);
  for (int i = 0; i < sh.num_io; i++) sum_out = sum_out + row.l(sh.pulse_col(2 * i + 1));
  cs.c(row.l(0) - sum_out);
  Horner2<P> hf(FLAGS_U64_CONSTRAINTS);
  flags_u64_block(cs, row, 0, hf);
  P hfv[SBN_NCH];
  hf.value(hfv);
  cs.merge(hfv, one, FLAGS_U64_CONSTRAINTS);
  const int st = FlagU64Shape::MAIN_COLS;
  P counter = row.l(st);
  cs.cf(counter);
  cs.ct(row.n(st) - counter - one);
  for (int i = 0; i < 2 * sh.num_io; i++) {
    u64 pos = (u64)(i >> 1) * 128 + ((i & 1) ? 127 : 0);
    P cmp = counter - lift<P>(pos);
    P pulse = row.l(sh.pulse_col(i));
    cs.c(cmp * row.l(sh.witness_col(i)) + pulse - one);
    cs.c(cmp * pulse);
  }
}
template <class P, class Row>
GL_HD void flag_eval(Cons<P>& cs, const Row& row, const FlagShape& sh) {   // flags.rs:449-492, in its emission order
  const P one = lift<P>(1);
  P sum_out = lift<P>(0);
  for (int i = 0; i < sh.num_io; i++) sum_out = sum_out + row.l(sh.pulse_col(2 * i + 1));
  cs.c(row.l(0) - sum_out);                                  // is_final - sum of the output pulses
  Horner2<P> hf(FLAGS_CONSTRAINTS);
  flags_block(cs, row, 0, hf);
  P hfv[SBN_NCH];
  hf.value(hfv);
  cs.merge(hfv, one, FLAGS_CONSTRAINTS);
  {   // eval_periodic_pulse(pulse_col = is_rotate, period 64, first_pulse 62)   pulse.rs:146-170
    const int st = FlagShape::START_PERIODIC;
    P counter = row.l(st), witness = row.l(st + 1), is_reset = row.l(1), next_counter = row.n(st);
    cs.cf(counter - lift<P>(1));
    cs.ct((one - is_reset) * (next_counter - counter - one));
    cs.ct(is_reset * next_counter);
    P delta = counter - lift<P>(63);
    cs.c(delta * witness + is_reset - one);
    cs.c(delta * is_reset);
  }
  {   // eval_pulse over the first and last row of every block   pulse.rs:45-63
    const int st = FlagShape::START_IO_PULSES;
    P counter = row.l(st);
    cs.cf(counter);
    cs.ct(row.n(st) - counter - one);
    for (int i = 0; i < 2 * sh.num_io; i++) {
      u64 pos = (u64)(i >> 1) * 512 + ((i & 1) ? 511 : 0);
      P cmp = counter - lift<P>(pos);
      P pulse = row.l(sh.pulse_col(i));
      cs.c(cmp * row.l(sh.witness_col(i)) + pulse - one);
      cs.c(cmp * pulse);
    }
  }
}

// ---- G1ExpStark / G2ExpStark (src/curves/g1/exp.rs, src/curves/g2/exp.rs) --------------------------------
// Both tables are the same double-and-add machine; E = 1 (Fq coordinates) or 2 (Fq2 coordinates) scales
// the point columns (32E per point), the gadget (320E columns, 165E constraints) and the public inputs.
struct ExpShape {  // constants(num_io): g1/exp.rs:6-34, g2/exp.rs:6-34, fq12/exp.rs:6-34
  // E = 1: G1ExpStark, 2: G2ExpStark (curve tables), 12: Fq12ExpStark (offset * x^e, square-and-multiply),
  // 0: FqExpStark (the same square-and-multiply machine over one Fq coefficient; u32 public inputs, u16 range check;
  //    constants(num_io): fq/exp.rs:6-34),
  // 13: Fq12ExpU64Stark (Fq12ExpStark with a u64 exponent: 128 rows per instance, the 6-column flags of
  //    fq12_u64/flags_u64.rs, no rotation pulse, one public input for the exponent; constants: fq12_u64/exp_u64.rs:19-45)
  int E, num_io, W, L, pi_per_io, b_col, gadget_col, nx_col, gadget_cons, start_flags, num_main, start_periodic, start_io_pulses, start_lookups;
  int rc_start, num_rc, split_rc, num_cols, num_pi, rpb, n_exp_slots, flag_sq, flag_mul;
  GL_HD ExpShape(int e, int n) {
    E = e; num_io = n; rpb = 512; n_exp_slots = 8; flag_sq = 2; flag_mul = 4;
    if (e == 12 || e == 13) {
      W = 192; L = 192; gadget_col = 384; nx_col = 384; gadget_cons = 792; start_flags = 108 * 16;
      rc_start = 384; num_rc = 84 * 16 - 12; split_rc = 1;
      if (e == 13) { rpb = 128; n_exp_slots = 1; flag_sq = 1; flag_mul = 3; }
    } else if (e == 0) {
      W = 16; L = 8; gadget_col = 32; nx_col = 32; gadget_cons = 66; start_flags = 9 * 16;
      rc_start = 0; num_rc = 9 * 16 - 1; split_rc = 0;
    } else {
      W = 32 * e; L = 16 * e; gadget_col = 64 * e; nx_col = gadget_col + 16 * e; gadget_cons = 165 * e; start_flags = 384 * e;
      rc_start = 0; num_rc = 381 * e; split_rc = 0;
    }
    pi_per_io = 3 * L + n_exp_slots; b_col = W;
    if (e == 13) { num_main = start_flags + 6; start_periodic = -1; start_io_pulses = num_main; }
    else { num_main = start_flags + 14; start_periodic = num_main; start_io_pulses = start_periodic + 2; }
    start_lookups = start_io_pulses + 1 + 4 * n;
    num_cols = start_lookups + 1 + (split_rc ? 6 : 2) * num_rc; num_pi = pi_per_io * n;
  }
  GL_HD int num_pairs() const { return (split_rc ? 4 : 2) * num_rc; }
  // constraints of sections [9] and [10] of exp_eval (io pulses, range check): the tail when the evaluation is split
  GL_HD int num_tail_constraints() const { return 2 + 4 * num_io + (split_rc ? 5 * num_rc + 3 : 2 * num_rc + 3); }
  GL_HD int num_constraints() const {
    const int fc = E == 13 ? 9 : 26, pc = E == 13 ? 0 : 5;  // flags constraints, rotation-pulse constraints
    return 1 + num_pi + 3 * 2 * W + fc + 2 * gadget_cons + fc + pc + 2 + 4 * num_io + (split_rc ? 5 * num_rc + 3 : 2 * num_rc + 3);
  }
  // u16_range_check_pairs (range_check.rs:96-113) / split_u16_range_check_pairs (:230-246)
  GL_HD void pair(int z, int& lhs, int& rhs) const {
    if (split_rc) {
      int t = z >> 2, w = z & 3, i = start_lookups + 1 + 6 * t;
      if (w == 0) { lhs = start_lookups; rhs = i + 2; }
      else if (w == 1) { lhs = start_lookups; rhs = i + 5; }
      else if (w == 2) { lhs = i; rhs = i + 1; }
      else { lhs = i + 3; rhs = i + 4; }
    } else if (z & 1) { lhs = z >> 1; rhs = start_lookups + z; }
    else { lhs = start_lookups; rhs = start_lookups + 2 + z; }
  }
  GL_HD int witness_col(int i) const { return start_io_pulses + 1 + 2 * i; }  // pulse.rs:14
  GL_HD int pulse_col(int i) const { return start_io_pulses + 2 + 2 * i; }    // pulse.rs:10
  // Emission slot m (0..pi_per_io) of one instance's vec_equal calls.
  //  curves (g1/exp.rs:381-391, g2/exp.rs:391-414): x (L u32 limbs, input pulse), offset (L, input), output (L, output
  //    pulse), exp_val (8, input); a u32 limb is formed from two u16 columns.
  //  fq12 (fq12/exp.rs:362-372): per coefficient c: x[c] (16 u16 limbs, input), offset[c] (16, input), output[c] (16,
  //    output); then exp_val (8, input).
  // Public inputs are stored x, offset, exp_val, output in every table.
  GL_HD int pi_index(int m) const {
    if (E == 12 || E == 13) {
      if (m >= 576) return 384 + (m - 576);
      int c = m / 48, w = m % 48;
      return w < 16 ? 16 * c + w : (w < 32 ? 192 + 16 * c + (w - 16) : 384 + n_exp_slots + 16 * c + (w - 32));
    }
    return m < 2 * L ? m : (m < 3 * L ? 2 * L + 8 + (m - 2 * L) : 2 * L + (m - 3 * L));
  }
  GL_HD bool slot_is_out(int m) const {
    if (E == 12 || E == 13) return m < 576 && (m % 48) >= 32;
    return m >= 2 * L && m < 3 * L;
  }
  // first column of the compared value (a u16 pair for curves, one u16 column for fq12); -1 for the exponent limbs
  GL_HD int slot_col(int m) const {
    if (E == 12 || E == 13) {
      if (m >= 576) return -1;
      int c = m / 48, w = m % 48;
      return w < 16 ? 16 * c + w : (w < 32 ? 192 + 16 * c + (w - 16) : 192 + 16 * c + (w - 32));
    }
    return m < L ? 2 * m : (m < 2 * L ? b_col + 2 * (m - L) : (m < 3 * L ? b_col + 2 * (m - 2 * L) : -1));
  }
};

// Per-proof constants of the regrouped public-input block: for challenge j and instance i,
//   W = alpha_j^(pi_per_io*(num_io-1-i)), WA = W * sum_{m in IN} alpha_j^(pi_per_io-1-m) pi[i][..], WO likewise for OUT.
template <class P>
struct ExpPiConsts {
  P W[SBN_NCH][G1EXP_MAX_IO], WA[SBN_NCH][G1EXP_MAX_IO], WO[SBN_NCH][G1EXP_MAX_IO];
};
template <class P>
static inline void exp_pi_consts(const ExpShape& sh, const P* const apow[SBN_NCH], const P* pi, ExpPiConsts<P>& out) {
  const int S = sh.pi_per_io;
  for (int j = 0; j < SBN_NCH; j++)
    for (int i = 0; i < sh.num_io; i++) {
      P a = lift<P>(0), o = lift<P>(0);
      for (int m = 0; m < S; m++) {
        P t = apow[j][S - 1 - m] * pi[S * i + sh.pi_index(m)];
        if (sh.slot_is_out(m)) o = o + t; else a = a + t;
      }
      P w = apow[j][S * (sh.num_io - 1 - i)];
      out.W[j][i] = w; out.WA[j][i] = w * a; out.WO[j][i] = w * o;
    }
}

// `part`: 0 = all constraints, 1 = sections [1]-[8] only, 2 = sections [9]-[10] only (sh.num_tail_constraints() of them);
// 3 = part 2 WITHOUT the 2 num_rc lookup constraints of the u16 range check (their exponents are skipped: lookups_beside_permutation
// below emits them from the workgroups of the permutation checks, which load the same sorted / permuted columns anyway).
// The stream is a Horner sum in alpha, so the quotient kernel evaluates the parts in different workgroups and joins
// them as head * alpha^(tail count) + tail.
template <int E, class P, class Row>
GL_HD void exp_eval(Cons<P>& cs, const Row& row, const ExpShape& sh, const ExpPiConsts<P>* pic, int part = 0) {
  const bool lookups_elsewhere = part == 3;
  if (part == 3) part = 2;
  const P one = lift<P>(1), base = lift<P>(65536);
  const int sf = sh.start_flags, S = sh.pi_per_io;
  constexpr bool F12 = E == 12 || E == 13;  // Fq12 operands (u16 public-input limbs, split range check)
  P is_final = row.l(sf), is_double = row.l(sf + sh.flag_sq), is_add = row.l(sf + sh.flag_mul);
  P is_not_final = one - is_final;
  if (part != 2) {
  // [1] is_final - sum(output pulses)                                         g1/exp.rs:359-365
  // [2] public-input binding, regrouped                                       g1/exp.rs:368-392, g2/exp.rs:382-414
  {
    Acc<P> vin[SBN_NCH], vout[SBN_NCH];
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) { vin[j].clear(); vout[j].clear(); }
    for (int m = 0; m < S; m++) {
      P v;
      int col = sh.slot_col(m);
      if (col >= 0) {
        v = row.l(col);
        if (!F12) v = v + base * row.l(col + 1);  // u16_columns_to_u32_columns, utils.rs:56 (curve tables)
      } else {
        int k = m - 3 * sh.L;
        v = row.l(sf + (E == 13 ? 5 : 6 + k));   // u64 table: the single running value (exp_u64.rs:364-366)
        if (k == 0) v = v + v + is_add;  // limbs[0]*2 + bit  (g1/exp.rs:389)
      }
      bool is_out = sh.slot_is_out(m);
#pragma unroll
      for (int j = 0; j < SBN_NCH; j++) { if (is_out) vout[j].mac(v, cs.apow[j][S - 1 - m]); else vin[j].mac(v, cs.apow[j][S - 1 - m]); }
    }
    P sum_out = lift<P>(0);
    Acc<P> s_in_wa[SBN_NCH], s_in_w[SBN_NCH], s_out_wo[SBN_NCH], s_out_w[SBN_NCH];
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) { s_in_wa[j].clear(); s_in_w[j].clear(); s_out_wo[j].clear(); s_out_w[j].clear(); }
#pragma unroll 2
    for (int i = 0; i < sh.num_io; i++) {
      P pin = row.l(sh.pulse_col(2 * i)), pout = row.l(sh.pulse_col(2 * i + 1));
      sum_out = sum_out + pout;
#pragma unroll
      for (int j = 0; j < SBN_NCH; j++) {
        s_in_wa[j].mac(pin, pic->WA[j][i]);
        s_in_w[j].mac(pin, pic->W[j][i]);
        s_out_wo[j].mac(pout, pic->WO[j][i]);
        s_out_w[j].mac(pout, pic->W[j][i]);
      }
    }
    cs.c(is_final - sum_out);
    P b[SBN_NCH];
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) b[j] = s_in_wa[j].value() - vin[j].value() * s_in_w[j].value() + s_out_wo[j].value() - vout[j].value() * s_out_w[j].value();
    cs.merge(b, one, sh.num_pi);   // the num_pi binding constraints, already folded
  }
  // [3] state transitions (fq_equal_transition / fq2_equal_transition x12)      g1/exp.rs:395-461, g2/exp.rs:416-473
  {
    constexpr int W = F12 ? 192 : (E == 0 ? 16 : 32 * E);  // columns of one operand (point / Fq12 element / Fq element)
    Horner2<P> d_na_a(W), d_nb_b(W), d_na_new(W), d_nb_new(W);  // (next_a - a), (next_b - b), (next_a - new), (next_b - new)
#pragma unroll 2
    for (int k = 0; k < W; k++) {
      P na = row.n(k), nb = row.n(W + k), nw = row.l(sh.nx_col + k);  // a, b and (new_x, new_y) are contiguous blocks
      d_na_a.push(cs, na - row.l(k));
      d_nb_b.push(cs, nb - row.l(W + k));
      d_na_new.push(cs, na - nw);
      d_nb_new.push(cs, nb - nw);
    }
    P hd[SBN_NCH], ha[SBN_NCH], hn[SBN_NCH], v_na_a[SBN_NCH], v_nb_b[SBN_NCH], v_na_new[SBN_NCH], v_nb_new[SBN_NCH];
    d_na_a.value(v_na_a); d_nb_b.value(v_nb_b); d_na_new.value(v_na_new); d_nb_new.value(v_nb_new);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) {
      P aw = cs.apow[j][W];
      hd[j] = v_na_new[j] * aw + v_nb_b[j];
      ha[j] = v_na_a[j] * aw + v_nb_new[j];
      hn[j] = v_na_a[j] * aw + v_nb_b[j];
    }
    P zl = cs.z_last * is_not_final;
    cs.merge(hd, zl * is_double, 2 * W);
    cs.merge(ha, zl * is_add, 2 * W);
    cs.merge(hn, zl * (one - is_double - is_add), 2 * W);
  }
  // [4] eval_flags, [5] eval_g*_add, [6] eval_g*_double, [7] eval_flags again   g1/exp.rs:462-472, g2/exp.rs:474-484
  constexpr int FC = E == 13 ? FLAGS_U64_CONSTRAINTS : FLAGS_CONSTRAINTS;
  Horner2<P> hf(FC);
  if (E == 13) flags_u64_block(cs, row, sf, hf); else flags_block(cs, row, sf, hf);
  P hfv[SBN_NCH];
  hf.value(hfv);
  cs.merge(hfv, one, FC);
  {
    P h_add[SBN_NCH], h_dbl[SBN_NCH];
    // curve tables: eval_g*_add (filter is_add) then eval_g*_double (is_double); fq12 / fq: eval_fq*_mul(is_sq, a, a)
    // then eval_fq*_mul(is_mul, a, b) -- is_sq shares the column of is_double, is_mul that of is_add.
    if (E == 1) g1_gadget(cs, row, h_add, h_dbl); else if (E == 2) g2_gadget(cs, row, h_add, h_dbl);
    else if (E == 0) fq_gadget(cs, row, h_dbl, h_add); else fq12_gadget(cs, row, h_dbl, h_add);
    if (F12 || E == 0) { cs.merge(h_dbl, is_double, sh.gadget_cons); cs.merge(h_add, is_add, sh.gadget_cons); }
    else { cs.merge(h_add, is_add, sh.gadget_cons); cs.merge(h_dbl, is_double, sh.gadget_cons); }
  }
  cs.merge(hfv, one, FC);
  // [8] eval_periodic_pulse(pulse_col = is_rotate, period 64, first_pulse 62)   pulse.rs:146-170 (not in the u64 table)
  if (E != 13) {
    const int st = sh.start_periodic;
    P counter = row.l(st), witness = row.l(st + 1), is_reset = row.l(sf + 1), next_counter = row.n(st);
    cs.cf(counter - lift<P>(1));  // period - first_pulse - 1 = 1
    cs.ct((one - is_reset) * (next_counter - counter - one));
    cs.ct(is_reset * next_counter);
    P delta = counter - lift<P>(63);
    cs.c(delta * witness + is_reset - one);
    cs.c(delta * is_reset);
  }
  }  // part != 2
  if (part == 1) return;
  // [9] eval_pulse over 2*num_io positions                                      pulse.rs:45-63
  {
    const int st = sh.start_io_pulses;
    P counter = row.l(st);
    cs.cf(counter);
    cs.ct(row.n(st) - counter - one);
#pragma unroll 4
    for (int i = 0; i < 2 * sh.num_io; i++) {
      u64 pos = (u64)(i >> 1) * sh.rpb + ((i & 1) ? sh.rpb - 1 : 0);  // get_pulse_positions, g1/exp.rs:153-163
      P cmp = counter - lift<P>(pos);
      P pulse = row.l(sh.pulse_col(i));
      cs.c(cmp * row.l(sh.witness_col(i)) + pulse - one);
      cs.c(cmp * pulse);
    }
  }
  // [10] eval_u16_range_check (range_check.rs:49-68) / eval_split_u16_range_check (:162-192)
  if (F12) {
    const P c256 = lift<P>(256);
    const int mc = sh.start_lookups;
#pragma unroll 4
    for (int i = 0; i < sh.num_rc; i++) cs.c(row.l(sh.rc_start + i) - (row.l(mc + 1 + 6 * i) + row.l(mc + 4 + 6 * i) * c256));
#pragma unroll 2
    for (int i = mc + 1; i < mc + 1 + 6 * sh.num_rc; i += 6) { lookup_pair(cs, row, i + 1, i + 2); lookup_pair(cs, row, i + 4, i + 5); }
    range_table_block(cs, row, mc, 255);
  } else {
    if (lookups_elsewhere) cs.rem -= 2 * sh.num_rc;
    else {
#pragma unroll 4
      for (int k = 0; k < sh.num_rc; k++) lookup_pair(cs, row, sh.start_lookups + 1 + 2 * k, sh.start_lookups + 2 + 2 * k);
    }
    range_table_block(cs, row, sh.start_lookups, 65535);
  }
}
// The lookup constraints of the u16 range check (eval_lookups, src/utils/lookup.rs:13-34) for the range-checked columns [k0, k1),
// emitted OUTSIDE the AIR's own segment with their absolute exponents: lookup k is followed by the 2 (num_rc - 1 - k) constraints of
// the later lookups, the 3 of the table block and the 2 num_zs permutation checks.  The permutation check of Z columns 2k, 2k + 1
// reads the sorted copy (start_lookups + 1 + 2k) and the permuted table (+ 2 + 2k) of column k: evaluated in the same workgroup,
// the lookups find both in cache and the tail segment no longer walks 2 num_rc columns of its own (0.8 GB of the G1 quotient
// stage's 4.4 GB).  The accumulators of a permutation segment are joined with weight alpha^0, so absolute exponents are right there.
// MEASURED SLOWER (round 4, G1: quotient stage 1.18 -> 1.25 ms, proofs identical): kept behind SBN_QUOTIENT_LOOKUPS=1 only.
template <class P, class Row>
GL_HD void lookups_beside_permutation(Cons<P>& cs, const Row& row, const ExpShape& sh, int num_zs, int k0, int k1) {
  const int after = 3 + 2 * num_zs;
#pragma unroll 2
  for (int k = k0; k < k1; k++) {
    const int col_in = sh.start_lookups + 1 + 2 * k, col_tab = col_in + 1;
    const P nin = row.n(col_in);
    const P d_table = nin - row.n(col_tab);
    const P c0 = (nin - row.l(col_in)) * d_table, c1 = d_table * cs.l_last;
    const int e = after + 2 * (sh.num_rc - 1 - k);
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) { cs.a[j].mac(c0, cs.apow[j][e + 1]); cs.a[j].mac(c1, cs.apow[j][e]); }
  }
}

// starky permutation.rs `eval_permutation_checks` for singleton pairs with batch size 2:
// Z index z <-> pair z; instance 0 uses challenge_sets[0].challenges[0].gamma, instance 1 uses
// challenge_sets[1].challenges[1].gamma.  ZRow: zl(z) / zn(z) = local / next Z values.
// The block is num_zs first-row constraints l_first (Z_z - 1), then num_zs transitions.  The consumer is a dot product with
// powers of alpha, so BOTH constraints of a Z column are emitted where its local value is loaded once: exponent
// base + 2 num_zs - 1 - z for the first-row constraint and base + num_zs - 1 - z for the transition, base = the constraints
// that follow the block (cs.rem on entry counts the block too).  [z0, z1) selects the Z columns of this call: the whole
// block is (0, num_zs); the quotient kernel splits it over two workgroups, which read disjoint halves of the Z matrix
// (the first-row constraints used to cost one of them a second pass over all of it: 0.8 GB of the 5.1 GB of the stage).
template <class P, class Row, class ZRow, class Shape>
GL_HD void permutation_checks(Cons<P>& cs, const Row& row, const ZRow& zrow, const Shape& sh, int num_zs, P gamma0, P gamma1, int z0 = 0, int z1 = -1) {
  const P one = lift<P>(1);
  if (z1 < 0) z1 = num_zs;
  const int base = cs.rem - 2 * num_zs;
#pragma unroll 4
  for (int z = z0; z < z1; z++) {
    int lc, rc;
    sh.pair(z, lc, rc);
    const P l = row.l(lc), r = row.l(rc), zl = zrow.zl(z);
    const P t = zrow.zn(z) * ((r + gamma0) * (r + gamma1)) - zl * ((l + gamma0) * (l + gamma1));
    const P f = (zl - one) * cs.l_first;   // (one multiply more than a separate first-row sum, but no second pair of accumulators: the kernel must stay at 64 registers)
#pragma unroll
    for (int j = 0; j < SBN_NCH; j++) {
      cs.a[j].mac(f, cs.apow[j][base + 2 * num_zs - 1 - z]);
      cs.a[j].mac(t, cs.apow[j][base + num_zs - 1 - z]);
    }
  }
  if (z0 == 0 && z1 == num_zs) cs.rem = base;
}
