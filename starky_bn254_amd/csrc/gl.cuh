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
}  // namespace gp
#if defined(__HIP_DEVICE_COMPILE__) && !defined(SBN_NO_ASM_MUL)
// gfx950 canonical multiply from the primitives above: the 128-bit product as four chained v_mad_u64_u32 (+ one 33-bit
// column add), reduction lo - hi_hi + hi_lo (2^32 - 1) with borrow / carry repaid through SGPR-pair carries, then one
// conditional subtraction of p (r >= p <=> r + 2^32 - 1 wraps).  ~21 VALU instructions where the compiler needs ~33, and
// -- unlike the hand-scheduled stream of round 1 (gl_mul_asm.inc, fixed window v112..v127, still used by the Poseidon
// S-boxes) -- no fixed registers: kernels that only multiply through F::operator* are no longer pinned at 128 VGPRs.
__device__ __forceinline__ u64 gl_mul_dev(u64 a, u64 b) {
  using namespace gp;
  const u32 a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
  u64 k, kc, k1, k2, k3, k4, k5, k6, k7, k8;
  const u64 p0 = mad0(a0, b0);
  const u64 p1 = mad(a0, b1, (u64)hi32(p0), k);
  const u64 p2 = mad(a1, b0, (u64)lo32(p1), k);
  const u32 sl = addco(hi32(p1), hi32(p2), kc);
  const u64 p3 = mad(a1, b1, pack(sl, sel01(kc)), k);
  // (p3 : lo) mod p
  const u32 l = subco(lo32(p0), hi32(p3), k1), h = subb0(lo32(p2), k1, k2);
  const u32 l2 = subco(l, selm1(k2), k3), h2 = subb0(h, k3, k4);       // borrow: take off 2^32 - 1 more; cannot borrow again
  const u64 r = madm1(lo32(p3), pack(l2, h2), k5);
  const u64 r2 = madm1(sel01(k5), r, k6);                              // carry: add 2^32 - 1; cannot carry again
  // canonical: r2 - p = r2 + (2^32 - 1) - 2^64
  u32 cl, ch;
  asm("v_add_co_u32_e64 %0, %1, %2, -1" : "=v"(cl), "=s"(k7) : "v"(lo32(r2)));
  asm("v_addc_co_u32_e64 %0, %1, %2, 0, %3" : "=v"(ch), "=s"(k8) : "v"(hi32(r2)), "s"(k7));
  u32 ol, oh;
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(ol) : "v"(lo32(r2)), "v"(cl), "s"(k8));
  asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(oh) : "v"(hi32(r2)), "v"(ch), "s"(k8));
  return pack(ol, oh);
}
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
