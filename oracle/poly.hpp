// ORACLE (test infrastructure, NOT product code).
// Polynomial / commitment layer of the reference's prover dependency, restated:
//   plonky2_field `fft.rs`, `polynomial/mod.rs` (ifft, lde, coset_fft, coset_ifft),
//   plonky2 `fri/oracle.rs::PolynomialBatch::{from_values,from_coeffs,get_lde_values}`,
//   plonky2 `hash/merkle_tree.rs::MerkleTree::{new,prove}`, `merkle_proofs.rs`.
// (un-vendored: reference Cargo.lock:529-531,591-593.)  Exact field arithmetic, so any FFT
// schedule yields the same bits; a plain iterative radix-2 is used.
#pragma once
#include "poseidon.hpp"
#include "poseidon_x8.hpp"
#include <algorithm>
#include <chrono>
#include <string>

namespace orc {

template <class T>
static inline void bit_reverse_inplace(std::vector<T>& a) {
  size_t n = a.size();
  unsigned lg = log2_strict(n);
  for (size_t i = 0; i < n; i++) {
    size_t j = reverse_bits(i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
}

struct Twiddles {
  // root powers w^0..w^(n/2-1) for n = 2^lg
  std::vector<GF> w;
  unsigned lg;
};
static inline const Twiddles& twiddles(unsigned lg, bool inverse) {
  static std::vector<Twiddles> fwd(33), inv(33);
  auto& t = inverse ? inv[lg] : fwd[lg];
#pragma omp critical(orc_twiddles)
  {
    if (t.w.empty()) {
      size_t half = lg ? ((size_t)1 << (lg - 1)) : 1;
      GF r = gf_root_of_unity(lg);
      if (inverse) r = gf_inv(r);
      std::vector<GF> w(half);
      GF acc = GF::one();
      for (size_t i = 0; i < half; i++) { w[i] = acc; acc = acc * r; }
      t.lg = lg;
      t.w.swap(w);
    }
  }
  return t;
}

// In-place NTT: a[k] <- sum_j a[j] w^{jk}, natural order in and out.
static inline void ntt_inplace(GF* a, size_t n, bool inverse) {
  unsigned lg = log2_strict(n);
  if (n == 1) return;
  for (size_t i = 0; i < n; i++) {
    size_t j = reverse_bits(i, lg);
    if (i < j) std::swap(a[i], a[j]);
  }
  const std::vector<GF>& w = twiddles(lg, inverse).w;
  for (unsigned s = 1; s <= lg; s++) {
    size_t m = (size_t)1 << s, half = m >> 1, step = n >> s;
    for (size_t k = 0; k < n; k += m)
      for (size_t j = 0; j < half; j++) {
        GF t = a[k + j + half] * w[j * step];
        GF u = a[k + j];
        a[k + j] = u + t;
        a[k + j + half] = u - t;
      }
  }
  if (inverse) {
    GF ninv = gf_inv(GF((u64)n));
    for (size_t i = 0; i < n; i++) a[i] = a[i] * ninv;
  }
}

// polynomial/mod.rs: values.ifft()
static inline std::vector<GF> ifft(std::vector<GF> v) { ntt_inplace(v.data(), v.size(), true); return v; }
static inline std::vector<GF> fft(std::vector<GF> v) { ntt_inplace(v.data(), v.size(), false); return v; }
// coeffs.lde(rate_bits).coset_fft(shift): scale coeff i by shift^i, zero-pad, FFT.
static inline std::vector<GF> coset_lde(const std::vector<GF>& coeffs, unsigned rate_bits, GF shift) {
  size_t n = coeffs.size();
  std::vector<GF> v(n << rate_bits);
  GF s = GF::one();
  for (size_t i = 0; i < n; i++) { v[i] = coeffs[i] * s; s = s * shift; }
  ntt_inplace(v.data(), v.size(), false);
  return v;
}
// values.coset_ifft(shift): iFFT then scale coeff i by shift^-i.
static inline std::vector<GF> coset_ifft(std::vector<GF> v, GF shift) {
  ntt_inplace(v.data(), v.size(), true);
  GF si = gf_inv(shift), s = GF::one();
  for (size_t i = 0; i < v.size(); i++) { v[i] = v[i] * s; s = s * si; }
  return v;
}

// Extension-field coset FFT = two base-field FFTs (the transform is F-linear).
static inline std::vector<Ext> ext_coset_fft(const std::vector<Ext>& coeffs, GF shift) {
  size_t n = coeffs.size();
  std::vector<GF> a(n), b(n);
  GF s = GF::one();
  for (size_t i = 0; i < n; i++) { a[i] = coeffs[i].a * s; b[i] = coeffs[i].b * s; s = s * shift; }
  ntt_inplace(a.data(), n, false);
  ntt_inplace(b.data(), n, false);
  std::vector<Ext> out(n);
  for (size_t i = 0; i < n; i++) out[i] = Ext(a[i], b[i]);
  return out;
}

// hash/merkle_tree.rs. `levels[0]` = leaf digests, last level = cap (2^cap_height digests).
struct MerkleTree {
  std::vector<std::vector<GF>> leaves;  // leaf data rows
  std::vector<std::vector<Digest>> levels;
  unsigned cap_height;
  const std::vector<Digest>& cap() const { return levels.back(); }
  void build(unsigned cap_h) {
    cap_height = cap_h;
    size_t n = leaves.size();
    unsigned lg = log2_strict(n);
    assert(cap_h <= lg);
    levels.clear();
    levels.emplace_back(n);
    auto& d0 = levels[0];
    // eight leaves / eight parents per AVX-512 permutation when the CPU has it (poseidon_x8.hpp: the plain round form, lane-wise;
    // the same digests as the scalar path, tests/test_oracle_core.py), the scalar sparse form otherwise
    const size_t width = n ? leaves[0].size() : 0;
    bool uniform = px8::available() && n >= 8 && width > 4;
    for (size_t i = 0; uniform && i < n; i++) uniform = leaves[i].size() == width;
#if defined(__x86_64__)
    if (uniform) {
#pragma omp parallel for schedule(static)
      for (size_t i0 = 0; i0 < n; i0 += 8) {
        const GF* rows[8];
        for (int j = 0; j < 8; j++) rows[j] = leaves[i0 + j].data();
        px8::hash_rows8(rows, width, &d0[i0]);
      }
    } else
#endif
    {
#pragma omp parallel for schedule(static)
      for (size_t i = 0; i < n; i++) d0[i] = hash_or_noop(leaves[i].data(), leaves[i].size());
    }
    for (unsigned l = lg; l > cap_h; l--) {
      const auto& prev = levels.back();
      std::vector<Digest> next(prev.size() / 2);
#if defined(__x86_64__)
      if (px8::available() && next.size() >= 8) {
#pragma omp parallel for schedule(static)
        for (size_t i0 = 0; i0 < next.size(); i0 += 8) px8::two_to_one8(&prev[2 * i0], &prev[2 * i0 + 1], 2, &next[i0]);
      } else
#endif
      {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < next.size(); i++) next[i] = two_to_one(prev[2 * i], prev[2 * i + 1]);
      }
      levels.push_back(std::move(next));
    }
  }
  // MerkleTree::prove: siblings bottom-up until the cap level.
  std::vector<Digest> prove(size_t leaf_index) const {
    std::vector<Digest> sib;
    size_t idx = leaf_index;
    for (size_t l = 0; l + 1 < levels.size(); l++) { sib.push_back(levels[l][idx ^ 1]); idx >>= 1; }
    return sib;
  }
};

// merkle_proofs.rs `verify_merkle_proof_to_cap`.
static inline bool verify_merkle_proof_to_cap(const std::vector<GF>& leaf, size_t leaf_index,
                                              const std::vector<Digest>& cap,
                                              const std::vector<Digest>& siblings) {
  size_t idx = leaf_index;
  Digest cur = hash_or_noop(leaf.data(), leaf.size());
  for (auto& s : siblings) {
    cur = (idx & 1) ? two_to_one(s, cur) : two_to_one(cur, s);
    idx >>= 1;
  }
  if (idx >= cap.size()) return false;
  return cur == cap[idx];
}

// Wall time of the phases inside from_values / from_coeffs, accumulated over the commitments of one prove() (cleared by
// the StageTimer of prove(), read back through orc_last_stage_seconds as "commit: ..." entries): what the CPU baseline's
// commitment time is made of, and which phases stop scaling with the thread count (tools/oracle_scaling.py).
static inline std::vector<std::pair<std::string, double>>& phase_log() { static std::vector<std::pair<std::string, double>> v; return v; }
struct PhaseTimer {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void lap(const char* name) {
    auto n = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(n - t).count();
    t = n;
    for (auto& e : phase_log()) if (e.first == name) { e.second += s; return; }
    phase_log().emplace_back(name, s);
  }
};

// fri/oracle.rs `PolynomialBatch` (blinding = false).
struct PolynomialBatch {
  std::vector<std::vector<GF>> polynomials;  // coefficient form, length N each
  MerkleTree tree;                           // leaves = LDE rows, bit-reversed index order
  unsigned degree_log, rate_bits;

  static PolynomialBatch from_coeffs(std::vector<std::vector<GF>> polys, unsigned rate_bits, unsigned cap_height) {
    PolynomialBatch b;
    size_t ncols = polys.size();
    size_t n = polys[0].size();
    b.degree_log = log2_strict(n);
    b.rate_bits = rate_bits;
    size_t m = n << rate_bits;
    unsigned lgm = b.degree_log + rate_bits;
    PhaseTimer pt;
    // transpose + reverse_index_bits_in_place, in CHUNKS of columns: a chunk's LDE columns are scattered into the leaf rows and
    // freed before the next chunk is transformed, so the transient column-major copy of the LDE matrix is bounded (1 GiB) instead
    // of a second whole matrix (49 GB for the 2^18-row Fq12 table; tools/oracle_digest.py).  Same leaves, same digests.
    // (blocks of 64 natural-order points: every column is read in contiguous runs and the 64 destination rows stay in cache)
    b.tree.leaves.assign(m, std::vector<GF>());
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < m; i++) b.tree.leaves[i].resize(ncols);
    pt.lap("commit: leaf allocation");
    size_t chunk = ((size_t)1 << 27) / m;          // 2^27 words = 1 GiB of LDE columns in flight
    if (chunk < 64) chunk = 64;
    const size_t TB = m < 64 ? m : 64;
    for (size_t c0 = 0; c0 < ncols; c0 += chunk) {
      const size_t nc = ncols - c0 < chunk ? ncols - c0 : chunk;
      std::vector<std::vector<GF>> lde(nc);
#pragma omp parallel for schedule(dynamic, 4)
      for (size_t c = 0; c < nc; c++) lde[c] = coset_lde(polys[c0 + c], rate_bits, GF(GL_GENERATOR));
      pt.lap("commit: coset LDE (per column)");
#pragma omp parallel for schedule(static)
      for (size_t s0 = 0; s0 < m; s0 += TB) {
        GF* dst[64];
        for (size_t k = 0; k < TB; k++) dst[k] = b.tree.leaves[reverse_bits(s0 + k, lgm)].data() + c0;
        for (size_t c = 0; c < nc; c++) {
          const GF* col = lde[c].data() + s0;
          for (size_t k = 0; k < TB; k++) dst[k][c] = col[k];
        }
      }
      pt.lap("commit: transpose + bit reversal");
    }
    b.tree.build(cap_height);
    pt.lap("commit: leaf hashes + Merkle levels");
    b.polynomials = std::move(polys);
    return b;
  }
  static PolynomialBatch from_values(const std::vector<std::vector<GF>>& values, unsigned rate_bits, unsigned cap_height) {
    PhaseTimer pt;
    std::vector<std::vector<GF>> coeffs(values.size());
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t c = 0; c < values.size(); c++) coeffs[c] = ifft(values[c]);
    pt.lap("commit: iFFT (per column)");
    return from_coeffs(std::move(coeffs), rate_bits, cap_height);
  }
  // get_lde_values(index, step): LDE row at natural index `index*step`.
  const std::vector<GF>& get_lde_values(size_t index, size_t step) const {
    return tree.leaves[reverse_bits(index * step, degree_log + rate_bits)];
  }
};

}  // namespace orc
