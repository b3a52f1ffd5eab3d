// ORACLE (test infrastructure, NOT product code).
// CPU restatement of the Goldilocks field and its quadratic extension as used by the
// reference's prover dependency (plonky2_field 0.1.1, `goldilocks_field.rs`,
// `extension/quadratic.rs`; un-vendored: Cargo.lock:591-593 of the reference).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use this.
//
// Pinned facts (checked in tests/test_oracle_field.py):
//   p = 2^64 - 2^32 + 1, multiplicative generator / coset shift 7,
//   POWER_OF_TWO_GENERATOR = 7^(2^32-1) = 1753635133440165772 (order 2^32),
//   extension F[X]/(X^2 - 7).
#pragma once
#include <cstdint>
#include <cstddef>
#include <vector>
#include <cassert>

namespace orc {

typedef uint64_t u64;
typedef unsigned __int128 u128;

static const u64 GL_P = 0xFFFFFFFF00000001ULL;
static const u64 GL_EPS = 0xFFFFFFFFULL;  // 2^64 mod p

struct GF {
  u64 v;  // always canonical (< p)
  GF() : v(0) {}
  explicit GF(u64 x) : v(x >= GL_P ? x - GL_P : x) {}
  static GF from_i64(int64_t x) { return x >= 0 ? GF((u64)x) : GF(GL_P - (u64)(-x)); }
  static GF zero() { return GF(); }
  static GF one() { return GF(1); }
  static GF neg_one() { return GF(GL_P - 1); }
  bool operator==(const GF& o) const { return v == o.v; }
  bool operator!=(const GF& o) const { return v != o.v; }
};

// Branch-free forms (masks instead of data-dependent jumps: on random field elements every such jump is a coin toss for
// the branch predictor, which cost the first version of this file two thirds of its time).
static inline GF operator+(GF a, GF b) {
  u64 s = a.v + b.v;
  u64 wrap = (u64)(s < a.v) | (u64)(s >= GL_P);
  GF r; r.v = s - (GL_P & (0 - wrap)); return r;
}
static inline GF operator-(GF a, GF b) {
  u64 d = a.v - b.v;
  GF r; r.v = d + (GL_P & (0 - (u64)(a.v < b.v))); return r;
}
static inline GF operator-(GF a) { GF r; r.v = a.v ? GL_P - a.v : 0; return r; }

static inline u64 gl_reduce128(u128 x) {
  u64 lo = (u64)x, hi = (u64)(x >> 64);
  u64 hi_hi = hi >> 32, hi_lo = hi & GL_EPS;
  // x = lo + hi_lo*2^64 + hi_hi*2^96 ; 2^64 = eps, 2^96 = -1 (mod p)
  u64 t0 = lo - hi_hi;
  t0 -= GL_EPS & (0 - (u64)(lo < hi_hi));  // borrow: add p == subtract eps (mod 2^64)
  u64 t1 = hi_lo * GL_EPS;                 // < 2^64
  u64 r = t0 + t1;
  r += GL_EPS & (0 - (u64)(r < t1));       // carry: subtract p == add eps (mod 2^64)
  r -= GL_P & (0 - (u64)(r >= GL_P));
  return r;
}
static inline GF operator*(GF a, GF b) {
  GF r; r.v = gl_reduce128((u128)a.v * b.v); return r;
}
static inline GF& operator+=(GF& a, GF b) { a = a + b; return a; }
static inline GF& operator-=(GF& a, GF b) { a = a - b; return a; }
static inline GF& operator*=(GF& a, GF b) { a = a * b; return a; }

static inline GF gf_pow(GF b, u64 e) {
  GF r = GF::one();
  while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
  return r;
}
static inline GF gf_inv(GF a) { assert(a.v != 0); return gf_pow(a, GL_P - 2); }
static inline GF gf_exp_pow2(GF a, unsigned k) { while (k--) a = a * a; return a; }

static const u64 GL_GENERATOR = 7;
static const u64 GL_POWER_OF_TWO_GENERATOR = 1753635133440165772ULL;
static const unsigned GL_TWO_ADICITY = 32;

// plonky2_field types.rs `primitive_root_of_unity(n_log)`.
static inline GF gf_root_of_unity(unsigned n_log) {
  assert(n_log <= GL_TWO_ADICITY);
  return gf_exp_pow2(GF(GL_POWER_OF_TWO_GENERATOR), GL_TWO_ADICITY - n_log);
}

// Montgomery batch inversion (all inputs non-zero).
static inline void gf_batch_inv(const GF* in, GF* out, size_t n) {
  if (!n) return;
  std::vector<GF> pre(n);
  GF acc = GF::one();
  for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = acc * in[i]; }
  GF inv = gf_inv(acc);
  for (size_t i = n; i-- > 0;) { GF t = in[i]; out[i] = inv * pre[i]; inv = inv * t; }
}

// Quadratic extension F[X]/(X^2-7), element = [c0, c1].
struct Ext {
  GF a, b;
  Ext() {}
  Ext(GF a_, GF b_) : a(a_), b(b_) {}
  explicit Ext(GF a_) : a(a_), b() {}
  explicit Ext(u64 x) : a(x), b() {}
  static Ext zero() { return Ext(); }
  static Ext one() { return Ext(GF::one()); }
  bool operator==(const Ext& o) const { return a == o.a && b == o.b; }
  bool operator!=(const Ext& o) const { return !(*this == o); }
};
static inline Ext operator+(Ext x, Ext y) { return Ext(x.a + y.a, x.b + y.b); }
static inline Ext operator-(Ext x, Ext y) { return Ext(x.a - y.a, x.b - y.b); }
static inline Ext operator-(Ext x) { return Ext(-x.a, -x.b); }
static inline Ext operator*(Ext x, Ext y) {
  return Ext(x.a * y.a + GF(7) * (x.b * y.b), x.a * y.b + x.b * y.a);
}
static inline Ext operator*(Ext x, GF s) { return Ext(x.a * s, x.b * s); }
static inline Ext operator+(Ext x, GF s) { return Ext(x.a + s, x.b); }
static inline Ext operator-(Ext x, GF s) { return Ext(x.a - s, x.b); }
static inline Ext& operator+=(Ext& x, Ext y) { x = x + y; return x; }
static inline Ext& operator-=(Ext& x, Ext y) { x = x - y; return x; }
static inline Ext& operator*=(Ext& x, Ext y) { x = x * y; return x; }
static inline Ext ext_inv(Ext x) {
  GF norm = x.a * x.a - GF(7) * (x.b * x.b);
  GF ni = gf_inv(norm);
  return Ext(x.a * ni, (-x.b) * ni);
}
static inline Ext ext_pow(Ext b, u64 e) {
  Ext r = Ext::one();
  while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; }
  return r;
}
static inline Ext ext_exp_pow2(Ext a, unsigned k) { while (k--) a = a * a; return a; }

static inline unsigned log2_strict(size_t n) {
  unsigned k = 0;
  while (((size_t)1 << k) < n) k++;
  assert(((size_t)1 << k) == n);
  return k;
}
static inline size_t reverse_bits(size_t x, unsigned bits) {
  size_t r = 0;
  for (unsigned i = 0; i < bits; i++) { r = (r << 1) | ((x >> i) & 1); }
  return r;
}

}  // namespace orc
