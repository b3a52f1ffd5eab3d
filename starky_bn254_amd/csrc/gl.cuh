// Goldilocks field (p = 2^64 - 2^32 + 1) and its quadratic extension F[X]/(X^2-7) for gfx950.
// Replaces plonky2_field `GoldilocksField` / `QuadraticExtension` on the prover path
// (reference dependency, Cargo.lock:591-593).  Values are kept canonical at every operation so
// that device results compare bit-for-bit with the host transcript.
//
// 64-bit modular multiply on CDNA4: there is no 64x64 VALU multiply; hipcc lowers `a*b` and
// __umul64hi to v_mul_lo_u32 / v_mul_hi_u32 / v_mad_u64_u32 sequences.  Reduction uses
// 2^64 = 2^32 - 1 and 2^96 = -1 (mod p): three adds/subs with carry fix-ups, no division.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint64_t u64;
typedef uint32_t u32;

// Forced inlining only in device code: on the host the same templates are instantiated over E2 for the
// verifier, where forced inlining of the whole constraint tree costs minutes of compile time.
#if defined(__HIP_DEVICE_COMPILE__)
#define GL_HD __host__ __device__ __forceinline__
#else
#define GL_HD __host__ __device__ inline
#endif

static constexpr u64 GLP = 0xFFFFFFFF00000001ULL;
static constexpr u64 GLEPS = 0xFFFFFFFFULL;
static constexpr u64 GL_GEN = 7;                                   // multiplicative generator = coset shift
static constexpr u64 GL_POW2_GEN = 1753635133440165772ULL;         // order 2^32

struct F {
  u64 v;
  GL_HD F() : v(0) {}
  GL_HD explicit F(u64 x) : v(x) {}  // caller guarantees x < p
  GL_HD static F from_u64(u64 x) { return F(x >= GLP ? x - GLP : x); }
  GL_HD bool operator==(const F& o) const { return v == o.v; }
  GL_HD bool operator!=(const F& o) const { return v != o.v; }
};

GL_HD F operator+(F a, F b) {
  u64 s = a.v + b.v;
  if (s < a.v || s >= GLP) s -= GLP;
  return F(s);
}
GL_HD F operator-(F a, F b) {
  u64 d = a.v - b.v;
  if (a.v < b.v) d += GLP;
  return F(d);
}
GL_HD F operator-(F a) { return F(a.v ? GLP - a.v : 0); }

GL_HD u64 gl_mulhi(u64 a, u64 b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (u64)(((unsigned __int128)a * b) >> 64);
#endif
}
// (hi:lo) mod p, result canonical.
GL_HD u64 gl_reduce128(u64 lo, u64 hi) {
  u64 hi_hi = hi >> 32, hi_lo = hi & GLEPS;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GLEPS;
  u64 t1 = (hi_lo << 32) - hi_lo;  // hi_lo * (2^32 - 1)
  u64 r = t0 + t1;
  if (r < t1) r += GLEPS;
  if (r >= GLP) r -= GLP;
  return r;
}
// (declared in both compilation passes: kernels.cuh's device functions name them, and the host pass parses those too)
// Single-instruction asm primitives with explicit carry operands (SGPR pairs): the building blocks of the weak arithmetic
// in the NTT (kernels.cuh, namespace nw) and of the sponge's constant additions (poseidon.cuh).  The compiler allocates
// every register and pads the SGPR write -> read wait states (it sees the operands of each statement).
namespace gp {
__device__ __forceinline__ u32 lo32(u64 x) { return (u32)x; }
__device__ __forceinline__ u32 hi32(u64 x) { return (u32)(x >> 32); }
__device__ __forceinline__ u64 pack(u32 lo, u32 hi) { return ((u64)hi << 32) | lo; }
__device__ __forceinline__ u32 addco(u32 a, u32 b, u64& k) { u32 r; asm("v_add_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(k) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u32 addc(u32 a, u32 b, u64 kin, u64& kout) { u32 r; asm("v_addc_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(kout) : "v"(a), "v"(b), "s"(kin)); return r; }
__device__ __forceinline__ u32 subco(u32 a, u32 b, u64& k) { u32 r; asm("v_sub_co_u32_e64 %0, %1, %2, %3" : "=v"(r), "=s"(k) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u32 subb(u32 a, u32 b, u64 kin, u64& kout) { u32 r; asm("v_subb_co_u32_e64 %0, %1, %2, %3, %4" : "=v"(r), "=s"(kout) : "v"(a), "v"(b), "s"(kin)); return r; }
__device__ __forceinline__ u32 subb0(u32 a, u64 kin, u64& kout) { u32 r; asm("v_subbrev_co_u32_e64 %0, %1, 0, %2, %3" : "=v"(r), "=s"(kout) : "v"(a), "s"(kin)); return r; }
__device__ __forceinline__ u32 sel01(u64 k) { u32 r; asm("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(r) : "s"(k)); return r; }
__device__ __forceinline__ u32 selm1(u64 k) { u32 r; asm("v_cndmask_b32_e64 %0, 0, -1, %1" : "=v"(r) : "s"(k)); return r; }
__device__ __forceinline__ u64 mad(u32 a, u32 b, u64 c, u64& k) { u64 r; asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(r), "=s"(k) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ u64 mad0(u32 a, u32 b) { u64 r, k; asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(r), "=s"(k) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ u64 madm1(u32 a, u64 c, u64& k) { u64 r; asm("v_mad_u64_u32 %0, %1, %2, -1, %3" : "=v"(r), "=s"(k) : "v"(a), "v"(c)); return r; }   // a * (2^32 - 1) + c
__device__ __forceinline__ u64 cmp_lt_s(u64 x, u64 c) { u64 k; asm("v_cmp_lt_u64_e64 %0, %1, %2" : "=s"(k) : "v"(x), "s"(c)); return k; }   // lanes with x < c (c uniform)
__device__ __forceinline__ u32 sel(u32 a, u32 b, u64 k) { u32 r; asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k)); return r; }   // k ? b : a
// weak -> canonical: w - p = w + (2^32 - 1) - 2^64, so the add carries out exactly when w >= p.  Four instructions with the
// select mask in an SGPR pair; `x >= p ? x - p : x` compiles to a compare plus a back-to-back v_cndmask_b32_e32 pair on VCC,
// which profiles/r2_valu_rates.txt prices at ~10 ns each at low occupancy.
__device__ __forceinline__ u64 canon(u64 w) {
  u64 k1, k2; u32 cl, ch;
  asm("v_add_co_u32_e64 %0, %1, %2, -1" : "=v"(cl), "=s"(k1) : "v"(lo32(w)));
  ch = addc(hi32(w), 0u, k1, k2);
  return pack(sel(lo32(w), cl, k2), sel(hi32(w), ch, k2));
}
}  // namespace gp
// gfx950 multiply from the primitives above, 13 instructions for the weak product (round 3; the derivation is in
// tools/gen_poseidon_sbox_asm.py): four chained v_mad_u64_u32 -- the third takes the whole second product as its addend, and
// its carry-out (weight 2^96 = -1) enters the reduction as the borrow-IN of lo - hi_hi; that subtraction's borrow (2^64 = 2^32 - 1
// too much) comes off the multiplier of the reducing multiply-add, hi_lo - b, whose own borrow b2 (only when hi_lo = 0; 2^32 (2^32 - 1)
// = -1) and carry c are repaid by one add-with-carry pair.  Then one conditional subtraction of p.  The compiler needs ~33
// instructions; unlike the hand-scheduled stream of the Poseidon S-boxes there are no fixed registers.
__device__ __forceinline__ u64 gl_mul_weak(u64 a, u64 b) {
  using namespace gp;
  const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  u64 k, km, k1, k2, k3, k4, k5, k6;
  const u64 p0 = mad0(a0, b0);
  const u64 p1 = mad(a0, b1, (u64)hi32(p0), k);
  const u64 p2 = mad(a1, b0, p1, km);
  const u64 p3 = mad(a1, b1, (u64)hi32(p2), k);
  const u32 l = subb(lo32(p0), hi32(p3), km, k1), h = subb0(lo32(p2), k1, k2);
  const u32 t = subb0(lo32(p3), k2, k3);
  const u64 r = madm1(t, pack(l, h), k4);
  const u32 rl = addc(lo32(r), selm1(k4), k3, k5), rh = addc(hi32(r), 0u, k5, k6);
  return pack(rl, rh);
}
__device__ __forceinline__ u64 gl_mul_dev(u64 a, u64 b) { return gp::canon(gl_mul_weak(a, b)); }
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SBN_NO_ASM_MUL)
GL_HD F operator*(F a, F b) { return F(gl_mul_dev(a.v, b.v)); }
#else
GL_HD F operator*(F a, F b) { return F(gl_reduce128(a.v * b.v, gl_mulhi(a.v, b.v))); }
#endif
GL_HD F& operator+=(F& a, F b) { a = a + b; return a; }
GL_HD F& operator-=(F& a, F b) { a = a - b; return a; }
GL_HD F& operator*=(F& a, F b) { a = a * b; return a; }

GL_HD F f_pow(F b, u64 e) {
  F r(1);
  while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
  return r;
}
GL_HD F f_inv(F a) { return f_pow(a, GLP - 2); }
GL_HD F f_exp_pow2(F a, unsigned k) { while (k--) a = a * a; return a; }
GL_HD F f_root_of_unity(unsigned n_log) { return f_exp_pow2(F(GL_POW2_GEN), 32 - n_log); }

// Quadratic extension element c0 + c1*X, X^2 = 7.
struct E2 {
  F a, b;
  GL_HD E2() {}
  GL_HD E2(F a_, F b_) : a(a_), b(b_) {}
  GL_HD explicit E2(F a_) : a(a_), b() {}
  GL_HD bool operator==(const E2& o) const { return a == o.a && b == o.b; }
  GL_HD bool operator!=(const E2& o) const { return !(*this == o); }
};
GL_HD E2 operator+(E2 x, E2 y) { return E2(x.a + y.a, x.b + y.b); }
GL_HD E2 operator-(E2 x, E2 y) { return E2(x.a - y.a, x.b - y.b); }
GL_HD E2 operator-(E2 x) { return E2(-x.a, -x.b); }
GL_HD F f_mul7(F x) {
  // 7x = 8x - x via 128-bit shift, one reduction
  u64 lo = x.v << 3, hi = x.v >> 61;
  u64 l2 = lo - x.v; if (lo < x.v) hi -= 1;
  return F(gl_reduce128(l2, hi));
}
GL_HD E2 operator*(E2 x, E2 y) { return E2(x.a * y.a + f_mul7(x.b * y.b), x.a * y.b + x.b * y.a); }
GL_HD E2 operator*(E2 x, F s) { return E2(x.a * s, x.b * s); }
GL_HD E2 operator+(E2 x, F s) { return E2(x.a + s, x.b); }
GL_HD E2 operator-(E2 x, F s) { return E2(x.a - s, x.b); }
GL_HD E2& operator+=(E2& x, E2 y) { x = x + y; return x; }
GL_HD E2& operator*=(E2& x, E2 y) { x = x * y; return x; }
GL_HD E2 e2_inv(E2 x) {
  F norm = x.a * x.a - f_mul7(x.b * x.b);
  F ni = f_inv(norm);
  return E2(x.a * ni, (-x.b) * ni);
}
GL_HD E2 e2_pow(E2 b, u64 e) {
  E2 r{F(1), F(0)};
  while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
  return r;
}
GL_HD E2 e2_exp_pow2(E2 a, unsigned k) { while (k--) a = a * a; return a; }

// Uniform "lift a small constant" for code templated over F / E2.
template <class P> GL_HD P lift(u64 v);
template <> GL_HD F lift<F>(u64 v) { return F(v); }
template <> GL_HD E2 lift<E2>(u64 v) { return E2(F(v), F(0)); }
template <class P> GL_HD P lift_f(F v);
template <> GL_HD F lift_f<F>(F v) { return v; }
template <> GL_HD E2 lift_f<E2>(F v) { return E2(v, F(0)); }

GL_HD u32 bitrev32(u32 x, unsigned bits) {
#if defined(__HIP_DEVICE_COMPILE__)
  return bits ? (__brev(x) >> (32 - bits)) : 0;
#else
  u32 r = 0;
  for (unsigned i = 0; i < bits; i++) r |= ((x >> i) & 1u) << (bits - 1 - i);
  return r;
#endif
}
