// ORACLE (test infrastructure, NOT product code).
// BN254 base-field arithmetic for witness generation.  The reference uses arkworks
// (ark-bn254 0.4.0 `Fq`, Cargo.lock:35-37) and num-bigint for this
// (src/curves/g1/muladd.rs:136,415; src/modular/modular.rs:38-100); here it is restated with
// 4x64-bit Montgomery arithmetic plus a 320-bit two's-complement integer for the exact quotient.
// Checked against Python big-int arithmetic in tests/test_oracle_bn254.py.
#pragma once
#include <cstdint>
#include <cstring>
#include <cassert>
#include <array>

namespace orc {

typedef unsigned __int128 u128_t;

struct U256 {
  uint64_t l[4];
  bool operator==(const U256& o) const { return !memcmp(l, o.l, sizeof l); }
  bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
};
static inline int u256_cmp(const U256& a, const U256& b) {
  for (int i = 3; i >= 0; i--) { if (a.l[i] < b.l[i]) return -1; if (a.l[i] > b.l[i]) return 1; }
  return 0;
}
static inline uint64_t u256_add(U256& r, const U256& a, const U256& b) {
  u128_t c = 0;
  for (int i = 0; i < 4; i++) { c += (u128_t)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  return (uint64_t)c;
}
static inline uint64_t u256_sub(U256& r, const U256& a, const U256& b) {
  uint64_t br = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t t = a.l[i] - b.l[i]; uint64_t b1 = a.l[i] < b.l[i];
    uint64_t t2 = t - br; uint64_t b2 = t < br;
    r.l[i] = t2; br = b1 | b2;
  }
  return br;
}

// p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
static const U256 BN254_P = {{0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}};

struct FqCtx {
  uint64_t n0inv;  // -p^-1 mod 2^64
  U256 r1, r2;     // R mod p, R^2 mod p
  FqCtx() {
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - BN254_P.l[0] * inv;
    n0inv = (uint64_t)0 - inv;
    U256 x = {{1, 0, 0, 0}};
    for (int i = 0; i < 512; i++) {
      U256 d; uint64_t c = u256_add(d, x, x);
      if (c || u256_cmp(d, BN254_P) >= 0) { U256 t; u256_sub(t, d, BN254_P); d = t; }
      x = d;
      if (i == 255) r1 = x;
    }
    r2 = x;
  }
};
static inline const FqCtx& fq_ctx() { static const FqCtx c; return c; }

// Montgomery-form element of Fq.
struct Fq {
  U256 m;
  bool operator==(const Fq& o) const { return m == o.m; }
};
static inline Fq fq_mont_mul(const Fq& a, const Fq& b) {
  const FqCtx& cx = fq_ctx();
  uint64_t t[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < 4; i++) {
    u128_t c = 0;
    for (int j = 0; j < 4; j++) { c += (u128_t)a.m.l[j] * b.m.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
    uint64_t mm = t[0] * cx.n0inv;
    c = (u128_t)mm * BN254_P.l[0] + t[0]; c >>= 64;
    for (int j = 1; j < 4; j++) { c += (u128_t)mm * BN254_P.l[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
    c += t[4]; t[3] = (uint64_t)c; c >>= 64;
    t[4] = t[5] + (uint64_t)c; t[5] = 0;
  }
  Fq r; memcpy(r.m.l, t, 32);
  if (t[4] || u256_cmp(r.m, BN254_P) >= 0) { U256 s; u256_sub(s, r.m, BN254_P); r.m = s; }
  return r;
}
static inline Fq fq_from_u256(const U256& x) { Fq a; a.m = x; Fq r2; r2.m = fq_ctx().r2; return fq_mont_mul(a, r2); }
static inline U256 fq_to_u256(const Fq& a) { Fq one; one.m = {{1, 0, 0, 0}}; return fq_mont_mul(a, one).m; }
static inline Fq fq_from_u64(uint64_t v) { U256 x = {{v, 0, 0, 0}}; return fq_from_u256(x); }
static inline Fq fq_add(const Fq& a, const Fq& b) {
  Fq r; uint64_t c = u256_add(r.m, a.m, b.m);
  if (c || u256_cmp(r.m, BN254_P) >= 0) { U256 s; u256_sub(s, r.m, BN254_P); r.m = s; }
  return r;
}
static inline Fq fq_sub(const Fq& a, const Fq& b) {
  Fq r; uint64_t br = u256_sub(r.m, a.m, b.m);
  if (br) { U256 s; u256_add(s, r.m, BN254_P); r.m = s; }
  return r;
}
static inline Fq fq_mul(const Fq& a, const Fq& b) { return fq_mont_mul(a, b); }
static inline Fq fq_inv(const Fq& a) {
  // a^(p-2)
  U256 e; U256 two = {{2, 0, 0, 0}}; u256_sub(e, BN254_P, two);
  Fq r; r.m = fq_ctx().r1;
  for (int i = 255; i >= 0; i--) {
    r = fq_mul(r, r);
    if ((e.l[i / 64] >> (i % 64)) & 1) r = fq_mul(r, a);
  }
  return r;
}

// 320-bit two's complement integer (enough to recover a |q| < 2^272 exact quotient mod 2^320).
struct I320 {
  uint64_t l[5];
  bool neg() const { return l[4] >> 63; }
};
static inline I320 i320_zero() { I320 z; memset(z.l, 0, sizeof z.l); return z; }
static inline void i320_add_shifted_i64(I320& a, int64_t v, unsigned shift) {
  // a += v << shift  (mod 2^320)
  if (shift >= 320) return;
  uint64_t ext = v < 0 ? ~0ULL : 0;
  // 5-limb sign-extended v, shifted by `shift`
  uint64_t src[6] = {(uint64_t)v, ext, ext, ext, ext, ext};
  unsigned ls = shift / 64, bs = shift % 64;
  u128_t c = 0;
  for (unsigned i = ls; i < 5; i++) {
    unsigned k = i - ls;
    uint64_t w = bs ? ((src[k] << bs) | (k ? (src[k - 1] >> (64 - bs)) : 0)) : src[k];
    c += (u128_t)a.l[i] + w; a.l[i] = (uint64_t)c; c >>= 64;
  }
}
static inline I320 i320_sub_u256(const I320& a, const U256& b) {
  I320 r; uint64_t br = 0;
  for (int i = 0; i < 5; i++) {
    uint64_t bi = i < 4 ? b.l[i] : 0;
    uint64_t t = a.l[i] - bi; uint64_t b1 = a.l[i] < bi;
    uint64_t t2 = t - br; uint64_t b2 = t < br;
    r.l[i] = t2; br = b1 | b2;
  }
  return r;
}
static inline I320 i320_mul(const I320& a, const I320& b) {
  I320 r = i320_zero();
  for (int i = 0; i < 5; i++) {
    u128_t c = 0;
    for (int j = 0; i + j < 5; j++) { c += (u128_t)a.l[i] * b.l[j] + r.l[i + j]; r.l[i + j] = (uint64_t)c; c >>= 64; }
  }
  return r;
}
static inline I320 i320_neg(const I320& a) {
  I320 r; u128_t c = 1;
  for (int i = 0; i < 5; i++) { c += (uint64_t)~a.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
  return r;
}
static inline const I320& bn254_p_inv_320() {
  static const I320 inv = [] {
    I320 p = i320_zero(); for (int i = 0; i < 4; i++) p.l[i] = BN254_P.l[i];
    I320 x = i320_zero(); x.l[0] = 1;
    for (int it = 0; it < 9; it++) {  // Newton: x <- x(2 - p x)
      I320 px = i320_mul(p, x);
      I320 two = i320_zero(); two.l[0] = 2;
      I320 t = i320_neg(px);
      u128_t c = 0;
      for (int i = 0; i < 5; i++) { c += (u128_t)t.l[i] + two.l[i]; t.l[i] = (uint64_t)c; c >>= 64; }
      x = i320_mul(x, t);
    }
    return x;
  }();
  return inv;
}

// 16-bit limb helpers (src/utils/utils.rs `bigint_to_columns`, `columns_to_bigint`).
static inline std::array<int64_t, 16> u256_to_limbs16(const U256& x) {
  std::array<int64_t, 16> r;
  for (int i = 0; i < 16; i++) r[i] = (int64_t)((x.l[i / 4] >> (16 * (i % 4))) & 0xffff);
  return r;
}
static inline U256 limbs16_to_u256(const int64_t* limbs) {
  U256 x = {{0, 0, 0, 0}};
  for (int i = 0; i < 16; i++) { assert(limbs[i] >= 0 && limbs[i] < 65536); x.l[i / 4] |= (uint64_t)limbs[i] << (16 * (i % 4)); }
  return x;
}

struct G1Affine { U256 x, y; };

}  // namespace orc
