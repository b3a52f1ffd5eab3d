// ORACLE (test infrastructure, NOT product code).
// FRI of plonky2 0.1.3 @ InternetMaximalism/plonky2 541e127 (un-vendored; reference
// Cargo.lock:529-531), restated from `plonky2/src/fri/{mod,reduction_strategies,oracle,prover,
// verifier,challenges,proof}.rs` and `plonky2/src/util/reducing.rs`.  [DEP-RECALL]: the dependency
// source is absent from /root/reference; this follows the upstream mid-2023 algorithm (SURVEY.md
// Appendix C).  Deterministic PoW: the SMALLEST witness is taken (upstream uses rayon find_any).
#pragma once
#include "poly.hpp"

namespace orc {

struct FriConfig {
  unsigned rate_bits = 1, cap_height = 4, proof_of_work_bits = 16;
  unsigned arity_bits = 4, final_poly_bits = 5;  // FriReductionStrategy::ConstantArityBits(4, 5)
  unsigned num_query_rounds = 84;
  // fri/oracle.rs `prove_openings`: "Multiply the final polynomial by `X`, so that `final_poly` has the maximum degree for
  // which the LDT will pass" (mir-protocol/plonky2 PR #436): the quotients keep their n-1 coefficients and a zero is
  // inserted in FRONT; the verifier's fri_combine_initial returns `sum * subgroup_x`.  Present in the 0.1.x line the
  // reference pins (recalled, [DEP-RECALL]); later upstream versions dropped the step and pad a zero at the END instead
  // (final_poly_times_x = false).  Not a StarkConfig field upstream: a switch of this restatement only.
  bool final_poly_times_x = true;
};
// starky config.rs `StarkConfig::standard_fast_config`.
struct StarkConfig {
  unsigned security_bits = 100;
  unsigned num_challenges = 2;
  FriConfig fri;
};
struct FriParams {
  FriConfig config;
  unsigned degree_bits;
  std::vector<unsigned> reduction_arity_bits;
  size_t lde_size() const { return (size_t)1 << (degree_bits + config.rate_bits); }
  unsigned total_arities() const { unsigned s = 0; for (auto a : reduction_arity_bits) s += a; return s; }
  size_t final_poly_len() const { return (size_t)1 << (degree_bits - total_arities()); }
};
// reduction_strategies.rs ConstantArityBits.
static inline FriParams fri_params(const FriConfig& c, unsigned degree_bits) {
  FriParams p; p.config = c; p.degree_bits = degree_bits;
  unsigned d = degree_bits;
  while (d > c.final_poly_bits && d + c.rate_bits - c.arity_bits >= c.cap_height) {
    p.reduction_arity_bits.push_back(c.arity_bits);
    d -= c.arity_bits;
  }
  return p;
}

struct FriQueryStep { std::vector<Ext> evals; std::vector<Digest> siblings; };
struct FriInitialTreeProof { std::vector<GF> evals; std::vector<Digest> siblings; };
struct FriQueryRound {
  std::vector<FriInitialTreeProof> initial;  // one per oracle (trace, perm Z, quotient)
  std::vector<FriQueryStep> steps;
};
struct FriProof {
  std::vector<std::vector<Digest>> commit_phase_merkle_caps;
  std::vector<FriQueryRound> query_round_proofs;
  std::vector<Ext> final_poly;
  GF pow_witness;
};

// One FRI opening batch: point + list of (oracle, polynomial index).
struct FriBatchInfo { Ext point; std::vector<std::pair<int, int>> polys; };
struct FriInstanceInfo { std::vector<size_t> oracle_num_polys; std::vector<FriBatchInfo> batches; };
struct FriOpenings { std::vector<std::vector<Ext>> batches; };

struct FriChallenges {
  Ext fri_alpha;
  std::vector<Ext> fri_betas;
  GF fri_pow_response;
  std::vector<size_t> fri_query_indices;
};

// ---------------------------------------------------------------------------------------------
// Prover side: fri/oracle.rs `prove_openings`, fri/prover.rs.
static inline FriProof prove_openings(const FriInstanceInfo& instance,
                                      const std::vector<const PolynomialBatch*>& oracles,
                                      Challenger& ch, const FriParams& params) {
  Ext alpha = ch.get_ext_challenge();
  size_t n = oracles[0]->polynomials[0].size();
  std::vector<Ext> final_poly;  // coefficient form
  for (auto& batch : instance.batches) {
    // alpha.reduce_polys_base: sum_j alpha^j f_j
    size_t k = batch.polys.size();
    std::vector<Ext> apow(k);
    { Ext a = Ext::one(); for (size_t j = 0; j < k; j++) { apow[j] = a; a = a * alpha; } }
    std::vector<Ext> comp(n);
    const size_t CH = 512;
#pragma omp parallel for schedule(static)
    for (size_t i0 = 0; i0 < n; i0 += CH) {
      size_t i1 = i0 + CH < n ? i0 + CH : n;
      for (size_t i = i0; i < i1; i++) comp[i] = Ext::zero();
      for (size_t j = 0; j < k; j++) {
        const GF* f = oracles[batch.polys[j].first]->polynomials[batch.polys[j].second].data();
        Ext a = apow[j];
        for (size_t i = i0; i < i1; i++) comp[i] += a * f[i];
      }
    }
    // divide_by_linear(point): synthetic division, remainder dropped; pad back to n.
    std::vector<Ext> quot(n);
    {
      Ext acc = Ext::zero();
      std::vector<Ext> bs(n);
      for (size_t i = n; i-- > 0;) { acc = acc * batch.point + comp[i]; bs[i] = acc; }
      // bs[i] for i>=1 are the quotient coefficients q[i-1]; bs[0] is the remainder.
      for (size_t i = 0; i + 1 < n; i++) quot[i] = bs[i + 1];
      quot[n - 1] = Ext::zero();
    }
    // alpha.shift_poly(final_poly): final_poly *= alpha^count ; count = k
    Ext shift = ext_pow(alpha, (u64)k);
    if (final_poly.empty()) final_poly.assign(n, Ext::zero());
    for (size_t i = 0; i < n; i++) final_poly[i] = final_poly[i] * shift + quot[i];
  }
  // lde + coset FFT over the extension (with the multiply-by-X step: final_poly.coeffs.insert(0, ZERO); the last padded
  // coefficient of every quotient is zero, so the shift loses nothing and the length stays n)
  std::vector<Ext> coeffs(n << params.config.rate_bits, Ext::zero());
  if (params.config.final_poly_times_x) { for (size_t i = 0; i + 1 < n; i++) coeffs[i + 1] = final_poly[i]; }
  else for (size_t i = 0; i < n; i++) coeffs[i] = final_poly[i];
  std::vector<Ext> values = ext_coset_fft(coeffs, GF(GL_GENERATOR));

  FriProof proof;
  // fri_committed_trees
  std::vector<MerkleTree> trees;
  GF shift = GF(GL_GENERATOR);
  for (unsigned arity_bits : params.reduction_arity_bits) {
    size_t arity = (size_t)1 << arity_bits;
    bit_reverse_inplace(values);
    MerkleTree t;
    t.leaves.resize(values.size() / arity);
    for (size_t i = 0; i < t.leaves.size(); i++) {
      auto& leaf = t.leaves[i];
      leaf.resize(2 * arity);
      for (size_t j = 0; j < arity; j++) { leaf[2 * j] = values[i * arity + j].a; leaf[2 * j + 1] = values[i * arity + j].b; }
    }
    t.build(params.config.cap_height);
    ch.observe_cap(t.cap());
    proof.commit_phase_merkle_caps.push_back(t.cap());
    trees.push_back(std::move(t));
    Ext beta = ch.get_ext_challenge();
    // coeffs <- chunks(arity).map(reduce_with_powers(chunk, beta))
    std::vector<Ext> folded(coeffs.size() / arity);
    for (size_t i = 0; i < folded.size(); i++) {
      Ext acc = Ext::zero();
      for (size_t j = arity; j-- > 0;) acc = acc * beta + coeffs[i * arity + j];
      folded[i] = acc;
    }
    coeffs.swap(folded);
    shift = gf_pow(shift, arity);
    values = ext_coset_fft(coeffs, shift);
  }
  coeffs.resize(coeffs.size() >> params.config.rate_bits);
  for (auto& c : coeffs) ch.observe(c);
  proof.final_poly = coeffs;

  // fri_proof_of_work: smallest witness with >= pow_bits leading zeros in the response.
  {
    PState inter = ch.state;
    size_t wpos = ch.input.size();
    for (size_t i = 0; i < ch.input.size(); i++) inter[i] = ch.input[i];
    unsigned min_lz = params.config.proof_of_work_bits;
    u64 found = ~0ULL;
    for (u64 base = 0; found == ~0ULL; base += (1u << 16)) {
#pragma omp parallel for schedule(static)
      for (long c = 0; c < (1 << 16); c++) {
        PState st = inter;
        st[wpos] = GF(base + (u64)c);
        poseidon_permute_fast(st);
        u64 resp = st[SPONGE_RATE - 1].v;
        unsigned lz = resp ? (unsigned)__builtin_clzll(resp) : 64;
        if (lz >= min_lz) {
#pragma omp critical(orc_pow)
          { if (base + (u64)c < found) found = base + (u64)c; }
        }
      }
    }
    proof.pow_witness = GF(found);
    ch.observe(proof.pow_witness);
    GF resp = ch.get_challenge();
    unsigned lz = resp.v ? (unsigned)__builtin_clzll(resp.v) : 64;
    assert(lz >= min_lz);
    (void)lz;
  }

  // fri_prover_query_rounds
  size_t lde_n = params.lde_size();
  for (unsigned q = 0; q < params.config.num_query_rounds; q++) {
    size_t x_index = (size_t)(ch.get_challenge().v % lde_n);
    FriQueryRound qr;
    for (auto* o : oracles) {
      FriInitialTreeProof ip;
      ip.evals = o->tree.leaves[x_index];
      ip.siblings = o->tree.prove(x_index);
      qr.initial.push_back(std::move(ip));
    }
    for (size_t i = 0; i < trees.size(); i++) {
      unsigned ab = params.reduction_arity_bits[i];
      size_t idx = x_index >> ab;
      FriQueryStep st;
      const auto& leaf = trees[i].leaves[idx];
      for (size_t j = 0; j < leaf.size(); j += 2) st.evals.push_back(Ext(leaf[j], leaf[j + 1]));
      st.siblings = trees[i].prove(idx);
      qr.steps.push_back(std::move(st));
      x_index = idx;
    }
    proof.query_round_proofs.push_back(std::move(qr));
  }
  return proof;
}

// ---------------------------------------------------------------------------------------------
// Verifier side: fri/challenges.rs, fri/verifier.rs.
static inline FriChallenges fri_challenges(Challenger& ch, const std::vector<std::vector<Digest>>& caps,
                                           const std::vector<Ext>& final_poly, GF pow_witness,
                                           unsigned degree_bits, const FriConfig& c) {
  FriChallenges r;
  size_t lde_size = (size_t)1 << (degree_bits + c.rate_bits);
  r.fri_alpha = ch.get_ext_challenge();
  for (auto& cap : caps) { ch.observe_cap(cap); r.fri_betas.push_back(ch.get_ext_challenge()); }
  for (auto& e : final_poly) ch.observe(e);
  ch.observe(pow_witness);
  r.fri_pow_response = ch.get_challenge();
  for (unsigned q = 0; q < c.num_query_rounds; q++) r.fri_query_indices.push_back((size_t)(ch.get_challenge().v % lde_size));
  return r;
}

// verifier.rs `compute_evaluation`: interpolate the coset {x*g^i} -> evals and evaluate at beta.
static inline Ext compute_evaluation(GF x, size_t x_index_within_coset, unsigned arity_bits,
                                     const std::vector<Ext>& evals_in, Ext beta) {
  size_t arity = (size_t)1 << arity_bits;
  GF g = gf_root_of_unity(arity_bits);
  std::vector<Ext> evals = evals_in;
  bit_reverse_inplace(evals);
  size_t rev = reverse_bits(x_index_within_coset, arity_bits);
  GF coset_start = x * gf_pow(g, arity - rev);
  std::vector<GF> pts(arity);
  { GF y = GF::one(); for (size_t i = 0; i < arity; i++) { pts[i] = coset_start * y; y = y * g; } }
  // Lagrange interpolation evaluated at beta.
  Ext res = Ext::zero();
  for (size_t i = 0; i < arity; i++) {
    Ext num = Ext::one();
    GF den = GF::one();
    for (size_t j = 0; j < arity; j++) if (j != i) { num = num * (beta - pts[j]); den = den * (pts[i] - pts[j]); }
    res += evals[i] * num * gf_inv(den);
  }
  return res;
}

static inline bool verify_fri_proof(const FriInstanceInfo& instance, const FriOpenings& openings,
                                    const FriChallenges& chal, const std::vector<std::vector<Digest>>& initial_caps,
                                    const FriProof& proof, const FriParams& params, const char** why) {
  static const char* dummy; if (!why) why = &dummy;
  size_t n = params.lde_size();
  unsigned log_n = params.degree_bits + params.config.rate_bits;
  // shape
  if (proof.commit_phase_merkle_caps.size() != params.reduction_arity_bits.size()) { *why = "fri: caps count"; return false; }
  if (proof.query_round_proofs.size() != params.config.num_query_rounds) { *why = "fri: query count"; return false; }
  if (proof.final_poly.size() != params.final_poly_len()) { *why = "fri: final poly len"; return false; }
  // PoW
  {
    u64 r = chal.fri_pow_response.v;
    unsigned lz = r ? (unsigned)__builtin_clzll(r) : 64;
    if (lz < params.config.proof_of_work_bits) { *why = "fri: pow"; return false; }
  }
  // PrecomputedReducedOpenings::from_os_and_alpha
  std::vector<Ext> reduced_openings;
  for (auto& b : openings.batches) {
    Ext acc = Ext::zero();
    for (size_t j = b.size(); j-- > 0;) acc = acc * chal.fri_alpha + b[j];
    reduced_openings.push_back(acc);
  }
  for (size_t q = 0; q < proof.query_round_proofs.size(); q++) {
    size_t x_index = chal.fri_query_indices[q];
    const FriQueryRound& rp = proof.query_round_proofs[q];
    if (rp.initial.size() != initial_caps.size()) { *why = "fri: initial trees count"; return false; }
    for (size_t t = 0; t < initial_caps.size(); t++) {
      if (rp.initial[t].evals.size() != instance.oracle_num_polys[t]) { *why = "fri: initial evals len"; return false; }
      if (!verify_merkle_proof_to_cap(rp.initial[t].evals, x_index, initial_caps[t], rp.initial[t].siblings)) { *why = "fri: initial merkle"; return false; }
    }
    GF subgroup_x = GF(GL_GENERATOR) * gf_pow(gf_root_of_unity(log_n), reverse_bits(x_index, log_n));
    // fri_combine_initial
    Ext old_eval = Ext::zero();
    {
      Ext sum = Ext::zero();
      for (size_t b = 0; b < instance.batches.size(); b++) {
        const auto& batch = instance.batches[b];
        Ext acc = Ext::zero();
        for (size_t j = batch.polys.size(); j-- > 0;) {
          GF e = rp.initial[batch.polys[j].first].evals[batch.polys[j].second];
          acc = acc * chal.fri_alpha + e;
        }
        Ext numerator = acc - reduced_openings[b];
        Ext denominator = Ext(subgroup_x) - batch.point;
        sum = sum * ext_pow(chal.fri_alpha, batch.polys.size());
        sum += numerator * ext_inv(denominator);
      }
      old_eval = params.config.final_poly_times_x ? sum * Ext(subgroup_x) : sum;
    }
    if (rp.steps.size() != params.reduction_arity_bits.size()) { *why = "fri: steps count"; return false; }
    for (size_t i = 0; i < params.reduction_arity_bits.size(); i++) {
      unsigned ab = params.reduction_arity_bits[i];
      size_t arity = (size_t)1 << ab;
      const auto& evals = rp.steps[i].evals;
      if (evals.size() != arity) { *why = "fri: step evals len"; return false; }
      size_t coset_index = x_index >> ab;
      size_t within = x_index & (arity - 1);
      if (evals[within] != old_eval) { *why = "fri: fold consistency"; return false; }
      old_eval = compute_evaluation(subgroup_x, within, ab, evals, chal.fri_betas[i]);
      std::vector<GF> flat(2 * arity);
      for (size_t j = 0; j < arity; j++) { flat[2 * j] = evals[j].a; flat[2 * j + 1] = evals[j].b; }
      if (!verify_merkle_proof_to_cap(flat, coset_index, proof.commit_phase_merkle_caps[i], rp.steps[i].siblings)) { *why = "fri: step merkle"; return false; }
      subgroup_x = gf_exp_pow2(subgroup_x, ab);
      x_index = coset_index;
    }
    // final poly check
    Ext fe = Ext::zero();
    for (size_t j = proof.final_poly.size(); j-- > 0;) fe = fe * Ext(subgroup_x) + proof.final_poly[j];
    if (fe != old_eval) { *why = "fri: final poly"; return false; }
  }
  (void)n;
  return true;
}

}  // namespace orc
