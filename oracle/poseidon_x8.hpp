// ORACLE (test infrastructure, NOT product code).
// Eight Poseidon-Goldilocks permutations at once, one per 64-bit lane of an AVX-512 register: the leaf hashing of the CPU
// baseline is 40 M independent sponges, so the CPU denominator of the north-star ratio should use the vector units the box has
// (VERDICT round 2, item 5).  The PLAIN round form of poseidon.hpp, lane-wise: add the round constants, x^7 on all twelve
// elements (full rounds) or on element 0 (partial rounds), the MDS layer with its 6-bit constants on the 32-bit halves.
// Bit-exact with poseidon_permute (tests/test_oracle_core.py); used by MerkleTree::build when the CPU has AVX-512F/DQ
// (run-time check), the scalar sparse form otherwise.  ORC_NO_AVX512=1 forces the scalar path.
#pragma once
#include "poseidon.hpp"
#if defined(__x86_64__)
#include <immintrin.h>
#define ORC_X8_TARGET __attribute__((target("avx512f,avx512dq")))

namespace orc {
namespace px8 {

typedef __m512i V;
ORC_X8_TARGET static inline V bc(u64 x) { return _mm512_set1_epi64((long long)x); }
// canonical + canonical -> canonical
ORC_X8_TARGET static inline V add(V a, V b) {
  V s = _mm512_add_epi64(a, b);
  __mmask8 carry = _mm512_cmplt_epu64_mask(s, a);                    // wrapped past 2^64: 2^64 = 2^32 - 1 (mod p); cannot wrap again (a, b < p)
  s = _mm512_mask_add_epi64(s, carry, s, bc(GL_EPS));
  __mmask8 ge = _mm512_cmpge_epu64_mask(s, bc(GL_P));
  return _mm512_mask_sub_epi64(s, ge, s, bc(GL_P));
}
// (hi : lo) mod p, canonical result; any hi, lo
ORC_X8_TARGET static inline V reduce128(V lo, V hi) {
  const V eps = bc(GL_EPS);
  V hi_hi = _mm512_srli_epi64(hi, 32), hi_lo = _mm512_and_si512(hi, eps);
  V t0 = _mm512_sub_epi64(lo, hi_hi);
  __mmask8 borrow = _mm512_cmplt_epu64_mask(lo, hi_hi);
  t0 = _mm512_mask_sub_epi64(t0, borrow, t0, eps);                   // + p == - eps (mod 2^64)
  V t1 = _mm512_sub_epi64(_mm512_slli_epi64(hi_lo, 32), hi_lo);     // hi_lo * (2^32 - 1) < 2^64
  V r = _mm512_add_epi64(t0, t1);
  __mmask8 carry = _mm512_cmplt_epu64_mask(r, t1);
  r = _mm512_mask_add_epi64(r, carry, r, eps);
  __mmask8 ge = _mm512_cmpge_epu64_mask(r, bc(GL_P));
  return _mm512_mask_sub_epi64(r, ge, r, bc(GL_P));
}
ORC_X8_TARGET static inline V mul(V a, V b) {
  const V m32 = bc(0xffffffffULL);
  V ah = _mm512_srli_epi64(a, 32), bh = _mm512_srli_epi64(b, 32);
  V ll = _mm512_mul_epu32(a, b), lh = _mm512_mul_epu32(a, bh), hl = _mm512_mul_epu32(ah, b), hh = _mm512_mul_epu32(ah, bh);
  V t = _mm512_add_epi64(hl, _mm512_srli_epi64(ll, 32));             // < 2^64: hl <= (2^32-1)^2
  V u = _mm512_add_epi64(lh, _mm512_and_si512(t, m32));              // < 2^64
  V lo = _mm512_or_si512(_mm512_slli_epi64(u, 32), _mm512_and_si512(ll, m32));
  V hi = _mm512_add_epi64(hh, _mm512_add_epi64(_mm512_srli_epi64(t, 32), _mm512_srli_epi64(u, 32)));
  return reduce128(lo, hi);
}
ORC_X8_TARGET static inline V sbox7(V x) {
  V x2 = mul(x, x), x4 = mul(x2, x2), x3 = mul(x2, x);
  return mul(x4, x3);
}
// poseidon.rs `mds_row_shf` on the 32-bit halves (entries < 2^6: the sums stay below 2^42)
ORC_X8_TARGET static inline void mds_layer(V* s) {
  const V m32 = bc(0xffffffffULL);
  V lo[24], hi[24];
  for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = _mm512_and_si512(s[i], m32); hi[i] = hi[i + 12] = _mm512_srli_epi64(s[i], 32); }
  V c[12];
  for (int i = 0; i < 12; i++) c[i] = bc(MDS_CIRC[i]);
  for (int r = 0; r < 12; r++) {
    V sl = _mm512_mul_epu32(lo[r], c[0]), sh = _mm512_mul_epu32(hi[r], c[0]);
    for (int i = 1; i < 12; i++) { sl = _mm512_add_epi64(sl, _mm512_mul_epu32(lo[i + r], c[i])); sh = _mm512_add_epi64(sh, _mm512_mul_epu32(hi[i + r], c[i])); }
    if (MDS_DIAG[r]) { const V d = bc(MDS_DIAG[r]); sl = _mm512_add_epi64(sl, _mm512_mul_epu32(lo[r], d)); sh = _mm512_add_epi64(sh, _mm512_mul_epu32(hi[r], d)); }   // DIAG = [8, 0, ...]
    // value = sl + sh * 2^32 (< 2^75) = (hi128 : lo128)
    V shl = _mm512_slli_epi64(sh, 32);
    V lo128 = _mm512_add_epi64(sl, shl);
    __mmask8 carry = _mm512_cmplt_epu64_mask(lo128, shl);
    V hi128 = _mm512_mask_add_epi64(_mm512_srli_epi64(sh, 32), carry, _mm512_srli_epi64(sh, 32), bc(1));
    s[r] = reduce128(lo128, hi128);
  }
}
// eight states, st[i] = element i of all eight: the plain permutation (poseidon.rs `Poseidon::poseidon`)
ORC_X8_TARGET static inline void permute(V* st) {
  const u64* rc = poseidon_consts().rc;
  int r = 0;
  for (int phase = 0; phase < 3; phase++) {
    const int n = phase == 1 ? N_PARTIAL_ROUNDS : N_FULL_ROUNDS_HALF;
    for (int k = 0; k < n; k++, r++) {
      for (int i = 0; i < 12; i++) st[i] = add(st[i], bc(rc[12 * r + i]));
      if (phase == 1) st[0] = sbox7(st[0]);
      else for (int i = 0; i < 12; i++) st[i] = sbox7(st[i]);
      mds_layer(st);
    }
  }
}
// states[j][i] = element i of state j, j < 8 (canonical in, canonical out)
ORC_X8_TARGET static inline void permute8(u64 (*states)[12]) {
  V st[12];
  const V idx = _mm512_setr_epi64(0, 12, 24, 36, 48, 60, 72, 84);
  for (int i = 0; i < 12; i++) st[i] = _mm512_i64gather_epi64(idx, (const long long*)&states[0][i], 8);
  permute(st);
  for (int i = 0; i < 12; i++) _mm512_i64scatter_epi64((long long*)&states[0][i], idx, st[i], 8);
}
// hash_or_noop of eight rows of `n` elements (n > 4): rows[j] points at row j; out[j] = its digest
ORC_X8_TARGET static inline void hash_rows8(const GF* const* rows, size_t n, Digest* out) {
  V st[12];
  for (int i = 0; i < 12; i++) st[i] = _mm512_setzero_si512();
  const V base = _mm512_setr_epi64((long long)rows[0], (long long)rows[1], (long long)rows[2], (long long)rows[3], (long long)rows[4], (long long)rows[5],
                                   (long long)rows[6], (long long)rows[7]);
  for (size_t off = 0; off < n; off += SPONGE_RATE) {
    const size_t len = n - off < (size_t)SPONGE_RATE ? n - off : (size_t)SPONGE_RATE;
    for (size_t i = 0; i < len; i++) st[i] = _mm512_i64gather_epi64(_mm512_add_epi64(base, bc((off + i) * 8)), nullptr, 1);   // overwrite mode
    permute(st);
  }
  alignas(64) u64 tmp[4][8];
  for (int i = 0; i < 4; i++) _mm512_store_si512((void*)tmp[i], st[i]);
  for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++) out[j].e[i].v = tmp[i][j];
}
// two_to_one of eight pairs: left[j], right[j] -> out[j]
ORC_X8_TARGET static inline void two_to_one8(const Digest* left, const Digest* right, size_t stride_digests, Digest* out) {
  alignas(64) u64 in[12][8];
  for (int j = 0; j < 8; j++) {
    for (int i = 0; i < 4; i++) { in[i][j] = left[j * stride_digests].e[i].v; in[4 + i][j] = right[j * stride_digests].e[i].v; }
    for (int i = 8; i < 12; i++) in[i][j] = 0;
  }
  V st[12];
  for (int i = 0; i < 12; i++) st[i] = _mm512_load_si512((const void*)in[i]);
  permute(st);
  alignas(64) u64 tmp[4][8];
  for (int i = 0; i < 4; i++) _mm512_store_si512((void*)tmp[i], st[i]);
  for (int j = 0; j < 8; j++) for (int i = 0; i < 4; i++) out[j].e[i].v = tmp[i][j];
}
static inline bool available() {
  static const bool ok = [] {
    const char* e = getenv("ORC_NO_AVX512");
    if (e && e[0] == '1') return false;
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512dq");
  }();
  return ok;
}

}  // namespace px8
}  // namespace orc
#else
namespace orc { namespace px8 { static inline bool available() { return false; } } }
#endif
