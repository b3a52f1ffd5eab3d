// The two 256-step curve chains of the G1ExpStark / G2ExpStark witness (bn254w.cuh exp_chains; reference loop
// src/curves/g1/exp.rs:165-230, src/curves/g2/exp.rs:180-246) for EIGHT instances at a time in AVX-512 IFMA registers.
//
// exp_chains is 512 dependent point operations per instance: 2.2 ms for 128 instances on 16 host threads, the longest single
// piece of the device witness generation and the only one that depends on the host's CPU share (18 ms on two threads).  The
// instances are independent and walk the same formulas, so eight of them fill the eight 64-bit lanes of a zmm register:
//   * Fq in radix 2^52, five limbs per element, one __m512i per limb (lane = instance); Montgomery form with R' = 2^260;
//   * a product is the operand-scanning Montgomery loop on vpmadd52luq / vpmadd52huq (50 + 50 multiply-adds, the quotient digit
//     m = t0 * (-p^-1) mod 2^52 is one more), accumulators stay below 2^58, one carry sweep and one conditional subtraction at
//     the end: every value is the canonical representative in [0, p), exactly like bnw::mmul / fadd / fsub;
//   * the exponent bit differs per lane: the addition b + a is computed for all lanes and blended in under the bit mask;
//   * every stored coordinate is multiplied by 2^256 2^-260 (one more Montgomery product with the plain constant 2^256 mod p),
//     which turns x 2^260 into x 2^256 mod p: the words written to the chain arrays are those of exp_chains, bit for bit
//     (tests/test_product_host.py compares the two forms).
// One host thread does 8 instances in ~0.15 ms; 128 instances are 16 such groups.  Run-time dispatch: CPUs without AVX-512 IFMA
// take exp_chains (tracegen.hip tracegen_host_chains).
#pragma once
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define SBN_HAVE_IFMA 1
#include <immintrin.h>
#include "bn254w.cuh"

namespace ifma {

#define IFMA_FN __attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl"), always_inline)) static inline

static constexpr u64 M52 = (1ULL << 52) - 1;
// p, -p^-1 mod 2^52, 2^520 mod p, 2^260 mod p, 2^256 mod p in radix 2^52
static constexpr u64 P52[5] = {0x08c16d87cfd47ULL, 0x916871ca8d3c2ULL, 0x181585d97816aULL, 0xa029b85045b68ULL, 0x030644e72e131ULL};
static constexpr u64 N0_52 = 0x20782e4866389ULL;
static constexpr u64 R2_52[5] = {0x8a81d1966eb04ULL, 0x6195018016b86ULL, 0xb4f898c98e615ULL, 0x9969bfd531600ULL, 0x00a8469a30d3aULL};
static constexpr u64 ONE_52[5] = {0x20880f6fce4b4ULL, 0x49baa989a8455ULL, 0x18f014a498908ULL, 0x724f85a9201d8ULL, 0x01f16424e1bb7ULL};
static constexpr u64 C256_52[5] = {0xd438dc58f0d9dULL, 0x28f5c70b3dd35ULL, 0x879462c0a78ebULL, 0xdf2f666ea36f7ULL, 0x00e0a77c19a07ULL};

struct V { __m512i l[5]; };   // eight field elements, limb-sliced

IFMA_FN V vconst(const u64* c) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_set1_epi64((long long)c[i]); return r; }
IFMA_FN V vzero() { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_setzero_si512(); return r; }
// carry sweep of limbs that may be negative or above 2^52 (value in [0, 2^260)): limbs 0..3 -> [0, 2^52), limb 4 keeps the rest (signed)
IFMA_FN void vcarry(V& t) {
  const __m512i mask = _mm512_set1_epi64((long long)M52);
  for (int i = 0; i < 4; i++) {
    const __m512i c = _mm512_srai_epi64(t.l[i], 52);
    t.l[i] = _mm512_and_si512(t.l[i], mask);
    t.l[i + 1] = _mm512_add_epi64(t.l[i + 1], c);
  }
}
// lanes with t < 0 (after vcarry the sign sits in limb 4)
IFMA_FN __mmask8 vneg(const V& t) { return _mm512_movepi64_mask(t.l[4]); }
IFMA_FN V vblend(__mmask8 m, const V& a, const V& b) { V r; for (int i = 0; i < 5; i++) r.l[i] = _mm512_mask_blend_epi64(m, a.l[i], b.l[i]); return r; }   // m ? b : a
IFMA_FN __mmask8 viszero(const V& a) {
  __m512i o = _mm512_or_si512(_mm512_or_si512(a.l[0], a.l[1]), _mm512_or_si512(_mm512_or_si512(a.l[2], a.l[3]), a.l[4]));
  return _mm512_testn_epi64_mask(o, o);
}
// a + b mod p, a - b mod p on canonical operands (fadd / fsub of bn254w.cuh)
IFMA_FN V vadd(const V& a, const V& b) {
  const V p = vconst(P52);
  V s, d;
  for (int i = 0; i < 5; i++) { s.l[i] = _mm512_add_epi64(a.l[i], b.l[i]); d.l[i] = _mm512_sub_epi64(s.l[i], p.l[i]); }
  vcarry(s); vcarry(d);
  return vblend(vneg(d), d, s);   // d < 0: keep s
}
IFMA_FN V vsub(const V& a, const V& b) {
  const V p = vconst(P52);
  V s, d;
  for (int i = 0; i < 5; i++) { s.l[i] = _mm512_sub_epi64(a.l[i], b.l[i]); d.l[i] = _mm512_add_epi64(s.l[i], p.l[i]); }
  vcarry(s); vcarry(d);
  return vblend(vneg(s), s, d);   // s < 0: take s + p
}
// Montgomery product a b 2^-260 mod p, canonical result; operands canonical (limbs < 2^52)
IFMA_FN V vmul(const V& a, const V& b) {
  const V p = vconst(P52);
  const __m512i n0 = _mm512_set1_epi64((long long)N0_52), z = _mm512_setzero_si512();
  __m512i t[6] = {z, z, z, z, z, z};
  for (int i = 0; i < 5; i++) {
    for (int j = 0; j < 5; j++) {
      t[j] = _mm512_madd52lo_epu64(t[j], a.l[j], b.l[i]);
      t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], a.l[j], b.l[i]);
    }
    const __m512i m = _mm512_madd52lo_epu64(z, t[0], n0);   // (the multiply reads the low 52 bits of t0 only)
    for (int j = 0; j < 5; j++) {
      t[j] = _mm512_madd52lo_epu64(t[j], m, p.l[j]);
      t[j + 1] = _mm512_madd52hi_epu64(t[j + 1], m, p.l[j]);
    }
    // t0 is now a multiple of 2^52: divide by 2^52
    t[0] = _mm512_add_epi64(t[1], _mm512_srli_epi64(t[0], 52));
    t[1] = t[2]; t[2] = t[3]; t[3] = t[4]; t[4] = t[5]; t[5] = z;
  }
  V s, d;
  for (int i = 0; i < 5; i++) { s.l[i] = t[i]; d.l[i] = _mm512_sub_epi64(t[i], p.l[i]); }
  vcarry(s); vcarry(d);
  return vblend(vneg(d), d, s);
}
IFMA_FN V vdbl(const V& a) { return vadd(a, a); }

// coordinates: E = 1 -> Fq, E = 2 -> Fq2 = Fq[i] / (i^2 + 1), as Co<E> of bn254w.cuh
template <int E> struct VC { V c[E]; };
template <int E> IFMA_FN VC<E> cadd(const VC<E>& a, const VC<E>& b) { VC<E> r; for (int q = 0; q < E; q++) r.c[q] = vadd(a.c[q], b.c[q]); return r; }
template <int E> IFMA_FN VC<E> csub(const VC<E>& a, const VC<E>& b) { VC<E> r; for (int q = 0; q < E; q++) r.c[q] = vsub(a.c[q], b.c[q]); return r; }
IFMA_FN VC<1> cmul(const VC<1>& a, const VC<1>& b) { VC<1> r; r.c[0] = vmul(a.c[0], b.c[0]); return r; }
IFMA_FN VC<2> cmul(const VC<2>& a, const VC<2>& b) {
  VC<2> r;
  r.c[0] = vsub(vmul(a.c[0], b.c[0]), vmul(a.c[1], b.c[1]));
  r.c[1] = vadd(vmul(a.c[0], b.c[1]), vmul(a.c[1], b.c[0]));
  return r;
}
template <int E> IFMA_FN __mmask8 czero(const VC<E>& a) { __mmask8 m = 0xff; for (int q = 0; q < E; q++) m &= viszero(a.c[q]); return m; }
template <int E> IFMA_FN VC<E> cblend(__mmask8 m, const VC<E>& a, const VC<E>& b) { VC<E> r; for (int q = 0; q < E; q++) r.c[q] = vblend(m, a.c[q], b.c[q]); return r; }
template <int E> struct VJ { VC<E> X, Y, Z; };

// bnw::jac_double (dbl-2009-l)
template <int E> IFMA_FN VJ<E> jdouble(const VJ<E>& p) {
  VC<E> A = cmul(p.X, p.X), B = cmul(p.Y, p.Y), C = cmul(B, B);
  VC<E> t = cadd(p.X, B); t = cmul(t, t); t = csub(csub(t, A), C);
  VC<E> D = cadd(t, t), Ee = cadd(cadd(A, A), A), F = cmul(Ee, Ee);
  VJ<E> r;
  r.X = csub(F, cadd(D, D));
  VC<E> C8 = cadd(C, C); C8 = cadd(C8, C8); C8 = cadd(C8, C8);
  r.Y = csub(cmul(Ee, csub(D, r.X)), C8);
  VC<E> yz = cmul(p.Y, p.Z);
  r.Z = cadd(yz, yz);
  return r;
}
// bnw::jac_add (add-2007-bl); *degenerate = lanes with H = 0
template <int E> IFMA_FN VJ<E> jadd(const VJ<E>& p, const VJ<E>& q, __mmask8* degenerate) {
  VC<E> Z1Z1 = cmul(p.Z, p.Z), Z2Z2 = cmul(q.Z, q.Z);
  VC<E> U1 = cmul(p.X, Z2Z2), U2 = cmul(q.X, Z1Z1);
  VC<E> S1 = cmul(cmul(p.Y, q.Z), Z2Z2), S2 = cmul(cmul(q.Y, p.Z), Z1Z1);
  VC<E> H = csub(U2, U1);
  *degenerate = czero(H);
  VC<E> I = cadd(H, H); I = cmul(I, I);
  VC<E> J = cmul(H, I);
  VC<E> r = csub(S2, S1); r = cadd(r, r);
  VC<E> V = cmul(U1, I);
  VJ<E> o;
  o.X = csub(csub(cmul(r, r), J), cadd(V, V));
  VC<E> sj = cmul(S1, J);
  o.Y = csub(cmul(r, csub(V, o.X)), cadd(sj, sj));
  VC<E> zz = cadd(p.Z, q.Z); zz = cmul(zz, zz); zz = csub(csub(zz, Z1Z1), Z2Z2);
  o.Z = cmul(zz, H);
  return o;
}

// eight standard-form values (u32 x 8 each, at ios + lane_off[lane]) -> Montgomery form (R' = 2^260)
IFMA_FN V vload_std(const uint32_t* const* src) {
  alignas(64) u64 limb[5][8];
  for (int lane = 0; lane < 8; lane++) {
    u64 w[4]; bnw::u32x8_to_u64x4(src[lane], w);
    limb[0][lane] = w[0] & M52;
    limb[1][lane] = ((w[0] >> 52) | (w[1] << 12)) & M52;
    limb[2][lane] = ((w[1] >> 40) | (w[2] << 24)) & M52;
    limb[3][lane] = ((w[2] >> 28) | (w[3] << 36)) & M52;
    limb[4][lane] = w[3] >> 16;
  }
  V x; for (int i = 0; i < 5; i++) x.l[i] = _mm512_load_si512((const void*)limb[i]);
  return vmul(x, vconst(R2_52));
}
// x 2^260 -> x 2^256 mod p (the 4 x 64 Montgomery words of bnw::Fq), scattered to dst[lane]
IFMA_FN void vstore_m256(const V& x, u64* const* dst, int lanes) {
  const V y = vmul(x, vconst(C256_52));
  alignas(64) u64 limb[5][8];
  for (int i = 0; i < 5; i++) _mm512_store_si512((void*)limb[i], y.l[i]);
  for (int lane = 0; lane < lanes; lane++) {
    u64* o = dst[lane];
    o[0] = limb[0][lane] | (limb[1][lane] << 52);
    o[1] = (limb[1][lane] >> 12) | (limb[2][lane] << 40);
    o[2] = (limb[2][lane] >> 24) | (limb[3][lane] << 28);
    o[3] = (limb[3][lane] >> 36) | (limb[4][lane] << 16);
  }
}

// exp_chains for the instances k0 .. k0 + lanes - 1 (lanes <= 8; missing lanes repeat the last instance and store nothing).
// Returns TG_ERR_* flags.
template <int E>
__attribute__((target("avx512f,avx512ifma,avx512dq,avx512vl"))) static int exp_chains_x8(const uint32_t* ios, size_t k0, int lanes, u64* ja, u64* jb) {
  const size_t iow = 8 * (4 * (size_t)E + 1);
  const uint32_t* io[8];
  for (int l = 0; l < 8; l++) io[l] = ios + iow * (k0 + (size_t)(l < lanes ? l : lanes - 1));
  VJ<E> a, b;
  VC<E>* dst[4] = {&a.X, &a.Y, &b.X, &b.Y};
  for (int v = 0; v < 4; v++)
    for (int q = 0; q < E; q++) {
      const uint32_t* src[8];
      for (int l = 0; l < 8; l++) src[l] = io[l] + 8 * (v * E + q);
      dst[v]->c[q] = vload_std(src);
    }
  for (int q = 0; q < E; q++) { a.Z.c[q] = q == 0 ? vconst(ONE_52) : vzero(); b.Z.c[q] = a.Z.c[q]; }
  __mmask8 bad = 0;
  for (int t = 0;; t++) {
    const VC<E>* co[6] = {&a.X, &a.Y, &a.Z, &b.X, &b.Y, &b.Z};
    for (int c = 0; c < 6; c++)
      for (int q = 0; q < E; q++) {
        u64* out[8];
        u64* base = c < 3 ? ja : jb;
        for (int l = 0; l < lanes; l++) out[l] = base + bnw::jac_at<E>(k0 + (size_t)l, t, c % 3) + 4 * q;
        vstore_m256(co[c]->c[q], out, lanes);
      }
    if (t == 256) break;
    __mmask8 bits = 0;
    for (int l = 0; l < 8; l++) bits |= (__mmask8)(((io[l][32 * E + (t >> 5)] >> (t & 31)) & 1u) << l);
    if (bits) {
      __mmask8 deg;
      const VJ<E> s = jadd<E>(b, a, &deg);
      bad |= deg & bits;
      b.X = cblend<E>(bits, b.X, s.X); b.Y = cblend<E>(bits, b.Y, s.Y); b.Z = cblend<E>(bits, b.Z, s.Z);
    }
    bad |= czero<E>(a.Y);
    a = jdouble<E>(a);
  }
  return (bad & (__mmask8)((1u << lanes) - 1)) ? (int)bnw::TG_ERR_DEGENERATE : 0;
}

static inline bool available() {
  static const bool have = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512ifma") && __builtin_cpu_supports("avx512dq") &&
                           __builtin_cpu_supports("avx512vl") && !sbn::Settings::from_env_or_default().no_avx512;
  return have;
}
#undef IFMA_FN
}  // namespace ifma
#endif
