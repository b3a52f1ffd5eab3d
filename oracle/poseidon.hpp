// ORACLE (test infrastructure, NOT product code).
// Poseidon-Goldilocks permutation (width 12, rate 8, x^7, 4+22+4 rounds), sponge hashing and the
// duplex Challenger of plonky2 0.1.3 @ InternetMaximalism/plonky2 541e127 (un-vendored dependency,
// reference Cargo.lock:529-531): `plonky2/src/hash/{poseidon,poseidon_goldilocks,hashing}.rs`,
// `plonky2/src/iop/challenger.rs`.
//
// PINNING. The 360 round constants are NOT stored here: they are re-derived by the procedure
// plonky2 documents for them (`plonky2/src/bin/generate_constants.rs`: ChaCha8Rng::seed_from_u64(0),
// `gen_range(0..ORDER)` per constant; rand 0.8 / rand_chacha 0.3 / rand_core 0.6 algorithms restated
// below).  The derived stream starts 0xb585f766f2144405, 0x7746a55f43921ad7, 0xb2fb0d31cee799b4,
// 0x0f6760a4803427d7 -- the values SURVEY.md recalls for ALL_ROUND_CONSTANTS -- and with the MDS
// (circulant [17,15,41,16,2,28,13,13,39,18,34,20], diagonal [8,0,...]) the permutation reproduces
// plonky2's own `poseidon_goldilocks.rs::test_vectors` outputs for inputs 0^12, 0..11 and (-1)^12
// (tests/golden/poseidon_kat.json).  The naive round form is used (algebraically identical to
// plonky2's "fast partial rounds").
#pragma once
#include "gl.hpp"
#include <array>
#include <cstring>

namespace orc {

static const int SPONGE_WIDTH = 12;
static const int SPONGE_RATE = 8;
static const int N_FULL_ROUNDS_HALF = 4;
static const int N_PARTIAL_ROUNDS = 22;
static const int N_ROUNDS = 30;

static const u64 MDS_CIRC[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
static const u64 MDS_DIAG[12] = {8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

// ---- ChaCha8Rng::seed_from_u64(0).gen_range(0..p) -------------------------------------------
struct ChaCha8Rng {
  uint32_t key[8];
  u64 ctr;
  uint32_t buf[16];
  int pos;
  static uint32_t rotl(uint32_t x, int n) { return (x << n) | (x >> (32 - n)); }
  explicit ChaCha8Rng(u64 seed) {
    // rand_core 0.6 SeedableRng::seed_from_u64: PCG32 expansion of the u64 into the 32-byte key.
    const u64 MUL = 6364136223846793005ULL, INC = 11634580027462260723ULL;
    u64 st = seed;
    for (int i = 0; i < 8; i++) {
      st = st * MUL + INC;
      uint32_t xs = (uint32_t)(((st >> 18) ^ st) >> 27);
      uint32_t rot = (uint32_t)(st >> 59);
      key[i] = (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    ctr = 0;
    pos = 16;
  }
  static void qr(uint32_t* s, int a, int b, int c, int d) {
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 16);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 12);
    s[a] += s[b]; s[d] = rotl(s[d] ^ s[a], 8);
    s[c] += s[d]; s[b] = rotl(s[b] ^ s[c], 7);
  }
  void refill() {
    uint32_t init[16] = {0x61707865, 0x3320646e, 0x79622d32, 0x6b206574};
    for (int i = 0; i < 8; i++) init[4 + i] = key[i];
    init[12] = (uint32_t)ctr; init[13] = (uint32_t)(ctr >> 32); init[14] = 0; init[15] = 0;
    uint32_t s[16];
    memcpy(s, init, sizeof s);
    for (int r = 0; r < 4; r++) {  // 8 rounds = 4 double rounds
      qr(s, 0, 4, 8, 12); qr(s, 1, 5, 9, 13); qr(s, 2, 6, 10, 14); qr(s, 3, 7, 11, 15);
      qr(s, 0, 5, 10, 15); qr(s, 1, 6, 11, 12); qr(s, 2, 7, 8, 13); qr(s, 3, 4, 9, 14);
    }
    for (int i = 0; i < 16; i++) buf[i] = s[i] + init[i];
    ctr++;
    pos = 0;
  }
  uint32_t next_u32() { if (pos == 16) refill(); return buf[pos++]; }
  u64 next_u64() { u64 lo = next_u32(); u64 hi = next_u32(); return (hi << 32) | lo; }
  // rand 0.8 UniformInt<u64>::sample_single for range 0..range (range has its top bit set here).
  u64 gen_range(u64 range) {
    unsigned lz = __builtin_clzll(range);
    u64 zone = (range << lz) - 1;
    for (;;) {
      u128 m = (u128)next_u64() * range;
      if ((u64)m <= zone) return (u64)(m >> 64);
    }
  }
};

struct PoseidonConsts {
  u64 rc[N_ROUNDS * SPONGE_WIDTH];
  PoseidonConsts() {
    ChaCha8Rng rng(0);
    for (int i = 0; i < N_ROUNDS * SPONGE_WIDTH; i++) rc[i] = rng.gen_range(GL_P);
  }
};
static inline const PoseidonConsts& poseidon_consts() {
  static const PoseidonConsts c;
  return c;
}

typedef std::array<GF, SPONGE_WIDTH> PState;

static inline GF sbox7(GF x) {
  GF x2 = x * x, x4 = x2 * x2, x3 = x2 * x;
  return x4 * x3;
}
static inline void mds_layer(PState& s) {
  // poseidon.rs `mds_row_shf`: out[r] = sum_i s[(i+r)%12]*CIRC[i] + s[r]*DIAG[r].
  // The entries are < 2^6, so the 32-bit halves of the state are accumulated separately in u64
  // (as plonky2's `mds_layer` does) and recombined with one 128-bit reduction per output.
  u64 lo[24], hi[24];
  for (int i = 0; i < 12; i++) { lo[i] = lo[i + 12] = s[i].v & 0xffffffffULL; hi[i] = hi[i + 12] = s[i].v >> 32; }
#pragma GCC unroll 12
  for (int r = 0; r < 12; r++) {
    u64 sl = 0, sh = 0;
#pragma GCC unroll 12
    for (int i = 0; i < 12; i++) { sl += lo[i + r] * MDS_CIRC[i]; sh += hi[i + r] * MDS_CIRC[i]; }
    sl += lo[r] * MDS_DIAG[r]; sh += hi[r] * MDS_DIAG[r];
    s[r].v = gl_reduce128((u128)sl + ((u128)sh << 32));
  }
}
// poseidon.rs `Poseidon::poseidon` (naive form).
static inline void poseidon_permute(PState& s) {
  const u64* rc = poseidon_consts().rc;
  int r = 0;
  for (int phase = 0; phase < 3; phase++) {
    int n = phase == 1 ? N_PARTIAL_ROUNDS : N_FULL_ROUNDS_HALF;
    for (int k = 0; k < n; k++, r++) {
      for (int i = 0; i < 12; i++) s[i] = s[i] + GF(rc[12 * r + i]);
      if (phase == 1) s[0] = sbox7(s[0]);
      else for (int i = 0; i < 12; i++) s[i] = sbox7(s[i]);
      mds_layer(s);
    }
  }
}

// The same permutation with the 22 partial rounds in their sparse form (Poseidon paper appendix B; plonky2's "fast partial
// rounds" are the same rewrite): one dense 11x11 product up front, then per round  s0' = m00 s0 + <vhat_r, s[1..]>,
// s[i] += what_r[i] s0.  Tables generated by tools/gen_poseidon_fast_partial.py (SBN_GEN_OUT=oracle), which checks the
// rewrite against the plain form; tests/test_oracle_core.py compares the two here as well.  poseidon_permute above stays
// the definition; the hashing below uses this one (4x fewer multiplications: the CPU baseline should not be a strawman).
namespace pfast {
static const u64 EFF_RC[360] = {
#include "poseidon_eff_consts.inc"
};
#include "poseidon_fast_consts.inc"
struct Acc { u128 lo = 0, hi = 0; };  // sum of 64x64 products, low and high words summed apart
static inline void mac(Acc& a, u64 x, u64 y) { u128 p = (u128)x * y; a.lo += (u64)p; a.hi += (u64)(p >> 64); }
static inline u64 fold(const Acc& a) { u128 t = a.lo + a.hi * (u128)GL_EPS; return gl_reduce128(t); }  // 2^64 = 2^32 - 1
}  // namespace pfast
static inline void poseidon_permute_fast(PState& s) {
  const u64* rc = pfast::EFF_RC;
  int r = 0;
  for (int k = 0; k < 4; k++, r++) {
    for (int i = 0; i < 12; i++) s[i] = sbox7(s[i] + GF(rc[12 * r + i]));
    mds_layer(s);
  }
  for (int i = 0; i < 12; i++) s[i] = s[i] + GF(rc[12 * r + i]);
  {
    u64 t[11];
    for (int i = 0; i < 11; i++) { pfast::Acc a; for (int j = 0; j < 11; j++) pfast::mac(a, pfast::PFAST_INIT[11 * i + j], s[1 + j].v); t[i] = pfast::fold(a); }
    for (int i = 0; i < 11; i++) s[1 + i].v = t[i];
  }
  for (int k = 0; k < 22; k++, r++) {
    const GF s0 = sbox7(k ? s[0] + GF(rc[12 * r]) : s[0]);
    pfast::Acc a; pfast::mac(a, pfast::PFAST_M00, s0.v);
    const u64* vh = pfast::PFAST_VHAT + 11 * k; const u64* wh = pfast::PFAST_WHAT + 11 * k;
    for (int i = 0; i < 11; i++) pfast::mac(a, vh[i], s[1 + i].v);
    for (int i = 0; i < 11; i++) { pfast::Acc b; b.lo = s[1 + i].v; pfast::mac(b, wh[i], s0.v); s[1 + i].v = pfast::fold(b); }
    s[0].v = pfast::fold(a);
  }
  for (int k = 0; k < 4; k++, r++) {
    for (int i = 0; i < 12; i++) s[i] = sbox7(s[i] + GF(rc[12 * r + i]));
    mds_layer(s);
  }
}

struct Digest {
  GF e[4];
  bool operator==(const Digest& o) const {
    return e[0] == o.e[0] && e[1] == o.e[1] && e[2] == o.e[2] && e[3] == o.e[3];
  }
};

// hashing.rs `hash_n_to_m_no_pad` with 4 outputs (overwrite-mode sponge).
static inline Digest hash_no_pad(const GF* in, size_t n) {
  PState st; for (auto& x : st) x = GF();
  for (size_t off = 0; off < n; off += SPONGE_RATE) {
    size_t len = n - off < (size_t)SPONGE_RATE ? n - off : SPONGE_RATE;
    for (size_t i = 0; i < len; i++) st[i] = in[off + i];
    poseidon_permute_fast(st);
  }
  Digest d; for (int i = 0; i < 4; i++) d.e[i] = st[i];
  return d;
}
// config.rs / hash_types: `hash_or_noop`.
static inline Digest hash_or_noop(const GF* in, size_t n) {
  if (n <= 4) { Digest d; for (int i = 0; i < 4; i++) d.e[i] = i < (int)n ? in[i] : GF(); return d; }
  return hash_no_pad(in, n);
}
// hashing.rs `compress` (= PoseidonHash::two_to_one).
static inline Digest two_to_one(const Digest& l, const Digest& r) {
  PState st; for (auto& x : st) x = GF();
  for (int i = 0; i < 4; i++) { st[i] = l.e[i]; st[4 + i] = r.e[i]; }
  poseidon_permute_fast(st);
  Digest d; for (int i = 0; i < 4; i++) d.e[i] = st[i];
  return d;
}

// iop/challenger.rs `Challenger` (duplex, overwrite mode; outputs popped from the back).
struct Challenger {
  PState state;
  std::vector<GF> input, output;
  Challenger() { for (auto& x : state) x = GF(); }
  void duplexing() {
    assert(input.size() <= (size_t)SPONGE_RATE);
    for (size_t i = 0; i < input.size(); i++) state[i] = input[i];
    input.clear();
    poseidon_permute(state);
    output.assign(state.begin(), state.begin() + SPONGE_RATE);
  }
  void observe(GF e) {
    output.clear();
    input.push_back(e);
    if (input.size() == (size_t)SPONGE_RATE) duplexing();
  }
  void observe(const Ext& e) { observe(e.a); observe(e.b); }
  void observe(const Digest& d) { for (int i = 0; i < 4; i++) observe(d.e[i]); }
  void observe_cap(const std::vector<Digest>& cap) { for (auto& d : cap) observe(d); }
  GF get_challenge() {
    if (!input.empty() || output.empty()) duplexing();
    GF r = output.back(); output.pop_back(); return r;
  }
  Ext get_ext_challenge() { GF a = get_challenge(); GF b = get_challenge(); return Ext(a, b); }
};

}  // namespace orc
