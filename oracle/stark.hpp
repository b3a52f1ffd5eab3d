// ORACLE (test infrastructure, NOT product code).
// Starky prove()/verify_stark_proof() of starky 0.1.1 @ InternetMaximalism/plonky2 541e127
// (un-vendored; reference Cargo.lock:797-799), restated from `starky/src/{prover,verifier,
// permutation,stark,constraint_consumer,vanishing_poly,proof,get_challenges,config}.rs`.
// Call sites in the reference: src/curves/g1/exp.rs:818-826, src/curves/g1/muladd.rs:669-677.
// [DEP-RECALL] -- "parity unpinned" for proof bytes: the reference holds no golden proof and cannot
// be built here (no Rust toolchain), so this restatement is pinned only by (a) Poseidon KATs,
// (b) its own verifier accepting, (c) tamper tests.  See DESIGN.md.
#pragma once
#include "fri.hpp"
#include <memory>
#include <string>
#include <chrono>
#include <cstdio>
#include <cstdlib>

namespace orc {

// constraint_consumer.rs
template <class P>
struct Consumer {
  std::vector<P> alphas, acc;
  P z_last, l_first, l_last;
  Consumer(const std::vector<P>& al, P zl, P lf, P ll) : alphas(al), acc(al.size(), P()), z_last(zl), l_first(lf), l_last(ll) {}
  inline void constraint(P c) { for (size_t j = 0; j < acc.size(); j++) acc[j] = acc[j] * alphas[j] + c; }
  inline void constraint_transition(P c) { constraint(c * z_last); }
  inline void constraint_first_row(P c) { constraint(c * l_first); }
  inline void constraint_last_row(P c) { constraint(c * l_last); }
};

// stark.rs `Stark` trait as the reference's AIRs implement it (src/curves/g1/exp.rs:330-742).
struct Air {
  virtual ~Air() {}
  virtual size_t num_columns() const = 0;
  virtual size_t num_public_inputs() const = 0;
  virtual unsigned constraint_degree() const { return 3; }
  // PermutationPair::singletons(lhs, rhs) list.
  virtual std::vector<std::pair<size_t, size_t>> permutation_pairs() const = 0;
  virtual void eval(const GF* lv, const GF* nv, const GF* pi, Consumer<GF>& c) const = 0;
  virtual void eval(const Ext* lv, const Ext* nv, const Ext* pi, Consumer<Ext>& c) const = 0;
  unsigned quotient_degree_factor() const { unsigned d = constraint_degree(); return d > 1 ? d - 1 : 1; }
  unsigned permutation_batch_size() const { return constraint_degree() - 1; }
  size_t num_permutation_batches(const StarkConfig& cfg) const {
    size_t inst = permutation_pairs().size() * cfg.num_challenges;
    size_t bs = permutation_batch_size();
    return (inst + bs - 1) / bs;
  }
};

struct StarkOpeningSet {
  std::vector<Ext> local_values, next_values, permutation_zs, permutation_zs_next, quotient_polys;
};
struct StarkProof {
  std::vector<Digest> trace_cap, permutation_zs_cap, quotient_polys_cap;
  StarkOpeningSet openings;
  FriProof opening_proof;
};
struct StarkProofWithPublicInputs {
  StarkProof proof;
  std::vector<GF> public_inputs;
  unsigned degree_bits = 0;  // carried in the serialized header (reference: recover_degree_bits)
};

// permutation.rs ------------------------------------------------------------------------------
struct PermChallenge { GF beta, gamma; };
typedef std::vector<PermChallenge> PermChallengeSet;  // num_challenges entries

static inline std::vector<PermChallengeSet> get_n_permutation_challenge_sets(Challenger& ch, unsigned num_challenges, unsigned num_sets) {
  std::vector<PermChallengeSet> sets(num_sets);
  for (auto& s : sets)
    for (unsigned c = 0; c < num_challenges; c++) { PermChallenge pc; pc.beta = ch.get_challenge(); pc.gamma = ch.get_challenge(); s.push_back(pc); }
  return sets;
}
struct PermInstance { size_t lhs, rhs; PermChallenge ch; };
// get_permutation_batches: cartesian_product(pairs, 0..num_challenges).chunks(batch_size);
// the i-th instance of a chunk takes challenge_sets[i].challenges[chal].
static inline std::vector<std::vector<PermInstance>> get_permutation_batches(
    const std::vector<std::pair<size_t, size_t>>& pairs, const std::vector<PermChallengeSet>& sets,
    unsigned num_challenges, unsigned batch_size) {
  std::vector<std::vector<PermInstance>> out;
  std::vector<PermInstance> cur;
  for (auto& p : pairs)
    for (unsigned chal = 0; chal < num_challenges; chal++) {
      PermInstance pi; pi.lhs = p.first; pi.rhs = p.second; pi.ch = sets[cur.size()][chal];
      cur.push_back(pi);
      if (cur.size() == batch_size) { out.push_back(cur); cur.clear(); }
    }
  if (!cur.empty()) out.push_back(cur);
  return out;
}
static inline std::vector<GF> compute_permutation_z_poly(const std::vector<PermInstance>& inst, const std::vector<std::vector<GF>>& trace) {
  size_t n = trace[0].size();
  std::vector<GF> num(n, GF::one()), den(n, GF::one());
  for (auto& in : inst) {
    const auto& l = trace[in.lhs]; const auto& r = trace[in.rhs];
    for (size_t i = 0; i < n; i++) { num[i] = num[i] * (l[i] + in.ch.gamma); den[i] = den[i] * (r[i] + in.ch.gamma); }
  }
  std::vector<GF> dinv(n);
  gf_batch_inv(den.data(), dinv.data(), n);
  std::vector<GF> z(n);
  GF acc = GF::one();
  for (size_t i = 0; i < n; i++) { z[i] = acc; acc = acc * (num[i] * dinv[i]); }
  return z;
}
template <class P>
static inline void eval_permutation_checks(const std::vector<std::vector<PermInstance>>& batches, const P* lv,
                                           const P* local_zs, const P* next_zs, Consumer<P>& c) {
  for (size_t i = 0; i < batches.size(); i++) c.constraint_first_row(local_zs[i] - GF::one());
  for (size_t i = 0; i < batches.size(); i++) {
    P lhs = P(GF::one()), rhs = P(GF::one());
    for (auto& in : batches[i]) { lhs = lhs * (lv[in.lhs] + in.ch.gamma); rhs = rhs * (lv[in.rhs] + in.ch.gamma); }
    c.constraint(next_zs[i] * rhs - local_zs[i] * lhs);
  }
}

static inline FriInstanceInfo stark_fri_instance(const Air& air, const StarkConfig& cfg, Ext zeta, GF g) {
  FriInstanceInfo fi;
  size_t ncol = air.num_columns();
  size_t nz = air.permutation_pairs().empty() ? 0 : air.num_permutation_batches(cfg);
  size_t nq = air.quotient_degree_factor() * cfg.num_challenges;
  FriBatchInfo b0, b1;
  b0.point = zeta; b1.point = zeta * g;
  int oi = 0;
  fi.oracle_num_polys.push_back(ncol);
  for (size_t i = 0; i < ncol; i++) { b0.polys.push_back({oi, (int)i}); b1.polys.push_back({oi, (int)i}); }
  oi++;
  if (nz) {
    fi.oracle_num_polys.push_back(nz);
    for (size_t i = 0; i < nz; i++) { b0.polys.push_back({oi, (int)i}); b1.polys.push_back({oi, (int)i}); }
    oi++;
  }
  fi.oracle_num_polys.push_back(nq);
  for (size_t i = 0; i < nq; i++) b0.polys.push_back({oi, (int)i});
  fi.batches.push_back(b0); fi.batches.push_back(b1);
  return fi;
}
static inline FriOpenings to_fri_openings(const StarkOpeningSet& o) {
  FriOpenings f; f.batches.resize(2);
  auto& z = f.batches[0]; auto& zn = f.batches[1];
  z.insert(z.end(), o.local_values.begin(), o.local_values.end());
  z.insert(z.end(), o.permutation_zs.begin(), o.permutation_zs.end());
  z.insert(z.end(), o.quotient_polys.begin(), o.quotient_polys.end());
  zn.insert(zn.end(), o.next_values.begin(), o.next_values.end());
  zn.insert(zn.end(), o.permutation_zs_next.begin(), o.permutation_zs_next.end());
  return f;
}

// Per-stage wall times of the last prove() (read back through orc_last_stage_seconds; ORC_TIMING=1 also prints them).
static inline std::vector<std::pair<std::string, double>>& stage_log() { static std::vector<std::pair<std::string, double>> v; return v; }
struct StageTimer {
  bool on; std::chrono::steady_clock::time_point t;
  StageTimer() : on(getenv("ORC_TIMING") != nullptr), t(std::chrono::steady_clock::now()) { stage_log().clear(); phase_log().clear(); }
  void lap(const char* name) {
    auto n = std::chrono::steady_clock::now();
    const double s = std::chrono::duration<double>(n - t).count();
    stage_log().emplace_back(name, s);
    if (on) fprintf(stderr, "[oracle] %-22s %8.3f s\n", name, s);
    t = n;
  }
};

struct ProveStages {  // optional per-stage outputs for parity tests
  std::vector<PermChallengeSet> perm_sets;
  std::vector<GF> alphas;
  Ext zeta;
};

// prover.rs `prove`.
static inline StarkProofWithPublicInputs prove(const Air& air, const StarkConfig& cfg,
                                               const std::vector<std::vector<GF>>& trace,
                                               const std::vector<GF>& public_inputs, ProveStages* stages = nullptr) {
  size_t degree = trace[0].size();
  unsigned degree_bits = log2_strict(degree);
  FriParams fp = fri_params(cfg.fri, degree_bits);
  unsigned rate_bits = cfg.fri.rate_bits, cap_height = cfg.fri.cap_height;
  assert(fp.total_arities() <= degree_bits + rate_bits - cap_height);
  assert(trace.size() == air.num_columns() && public_inputs.size() == air.num_public_inputs());

  StageTimer tm;
  PolynomialBatch trace_commitment = PolynomialBatch::from_values(trace, rate_bits, cap_height);
  tm.lap("trace commitment");
  Challenger ch;
  ch.observe_cap(trace_commitment.tree.cap());

  auto pairs = air.permutation_pairs();
  bool uses_perm = !pairs.empty();
  std::vector<PermChallengeSet> perm_sets;
  std::vector<std::vector<PermInstance>> batches;
  PolynomialBatch zs_commitment;
  if (uses_perm) {
    perm_sets = get_n_permutation_challenge_sets(ch, cfg.num_challenges, air.permutation_batch_size());
    batches = get_permutation_batches(pairs, perm_sets, cfg.num_challenges, air.permutation_batch_size());
    std::vector<std::vector<GF>> zs(batches.size());
#pragma omp parallel for schedule(dynamic, 4)
    for (size_t i = 0; i < batches.size(); i++) zs[i] = compute_permutation_z_poly(batches[i], trace);
    tm.lap("permutation z polys");
    zs_commitment = PolynomialBatch::from_values(zs, rate_bits, cap_height);
    tm.lap("z commitment");
    ch.observe_cap(zs_commitment.tree.cap());
  }
  std::vector<GF> alphas;
  for (unsigned i = 0; i < cfg.num_challenges; i++) alphas.push_back(ch.get_challenge());

  // compute_quotient_polys
  unsigned qdf = air.quotient_degree_factor();
  unsigned qdb = 0; while ((1u << qdb) < qdf) qdb++;
  assert(qdb <= rate_bits);
  size_t step = (size_t)1 << (rate_bits - qdb);
  size_t next_step = (size_t)1 << qdb;
  size_t size = degree << qdb;
  GF last = gf_inv(gf_root_of_unity(degree_bits));
  GF wq = gf_root_of_unity(degree_bits + qdb);
  std::vector<GF> coset(size);
  { GF x = GF(GL_GENERATOR); for (size_t i = 0; i < size; i++) { coset[i] = x; x = x * wq; } }
  // ZeroPolyOnCoset: Z_H(x_i) = shift^N * w_r^(i mod 2^r) - 1
  std::vector<GF> zh_inv((size_t)1 << qdb);
  {
    GF gn = gf_exp_pow2(GF(GL_GENERATOR), degree_bits);
    GF wr = gf_root_of_unity(qdb), y = GF::one();
    for (auto& z : zh_inv) { z = gf_inv(gn * y - GF::one()); y = y * wr; }
  }
  // Lagrange selectors of rows 0 and N-1 on the coset: L_0(x)=(x^N-1)/(N(x-1)), L_last(x)=(x^N-1)/(N(g x-1)).
  std::vector<GF> lag_first(size), lag_last(size);
  {
    GF g = gf_root_of_unity(degree_bits), nn = GF((u64)degree);
    std::vector<GF> d(2 * size), di(2 * size);
    for (size_t i = 0; i < size; i++) { d[i] = nn * (coset[i] - GF::one()); d[size + i] = nn * (g * coset[i] - GF::one()); }
    gf_batch_inv(d.data(), di.data(), 2 * size);
    for (size_t i = 0; i < size; i++) {
      GF zx = gf_exp_pow2(coset[i], degree_bits) - GF::one();
      lag_first[i] = zx * di[i]; lag_last[i] = zx * di[size + i];
    }
  }
  size_t nch = alphas.size();
  std::vector<std::vector<GF>> qvals(nch, std::vector<GF>(size));
#pragma omp parallel for schedule(dynamic, 16)
  for (size_t i = 0; i < size; i++) {
    size_t inext = (i + next_step) % size;
    Consumer<GF> cons(alphas, coset[i] - last, lag_first[i], lag_last[i]);
    const GF* lv = trace_commitment.get_lde_values(i, step).data();
    const GF* nv = trace_commitment.get_lde_values(inext, step).data();
    air.eval(lv, nv, public_inputs.data(), cons);
    if (uses_perm)
      eval_permutation_checks<GF>(batches, lv, zs_commitment.get_lde_values(i, step).data(),
                                  zs_commitment.get_lde_values(inext, step).data(), cons);
    GF dinv = zh_inv[i % zh_inv.size()];
    for (size_t j = 0; j < nch; j++) qvals[j][i] = cons.acc[j] * dinv;
  }
  tm.lap("quotient evaluation");
  std::vector<std::vector<GF>> quotient_chunks;
  for (size_t j = 0; j < nch; j++) {
    std::vector<GF> coeffs = coset_ifft(qvals[j], GF(GL_GENERATOR));
    // trim_to_len(degree * qdf): coefficients beyond must be zero.
    for (size_t i = degree * qdf; i < coeffs.size(); i++) assert(coeffs[i].v == 0);
    for (unsigned k = 0; k < qdf; k++) quotient_chunks.emplace_back(coeffs.begin() + k * degree, coeffs.begin() + (k + 1) * degree);
  }
  PolynomialBatch quotient_commitment = PolynomialBatch::from_coeffs(quotient_chunks, rate_bits, cap_height);
  ch.observe_cap(quotient_commitment.tree.cap());
  tm.lap("quotient commitment");

  Ext zeta = ch.get_ext_challenge();
  GF g = gf_root_of_unity(degree_bits);
  assert(ext_exp_pow2(zeta, degree_bits) != Ext::one());
  if (stages) { stages->perm_sets = perm_sets; stages->alphas = alphas; stages->zeta = zeta; }

  // StarkOpeningSet::new
  StarkProofWithPublicInputs out;
  out.degree_bits = degree_bits;
  StarkOpeningSet& os = out.proof.openings;
  auto eval_commitment = [&](Ext z, const PolynomialBatch& c) {
    std::vector<Ext> r(c.polynomials.size());
#pragma omp parallel for schedule(static)
    for (size_t p = 0; p < r.size(); p++) {
      const auto& co = c.polynomials[p];
      Ext acc = Ext::zero();
      for (size_t i = co.size(); i-- > 0;) acc = acc * z + co[i];
      r[p] = acc;
    }
    return r;
  };
  Ext zeta_next = zeta * g;
  os.local_values = eval_commitment(zeta, trace_commitment);
  os.next_values = eval_commitment(zeta_next, trace_commitment);
  if (uses_perm) {
    os.permutation_zs = eval_commitment(zeta, zs_commitment);
    os.permutation_zs_next = eval_commitment(zeta_next, zs_commitment);
  }
  os.quotient_polys = eval_commitment(zeta, quotient_commitment);
  FriOpenings fo = to_fri_openings(os);
  for (auto& b : fo.batches) for (auto& e : b) ch.observe(e);
  tm.lap("openings");

  std::vector<const PolynomialBatch*> oracles;
  oracles.push_back(&trace_commitment);
  if (uses_perm) oracles.push_back(&zs_commitment);
  oracles.push_back(&quotient_commitment);
  out.proof.opening_proof = prove_openings(stark_fri_instance(air, cfg, zeta, g), oracles, ch, fp);
  tm.lap("fri (prove_openings)");
  out.proof.trace_cap = trace_commitment.tree.cap();
  if (uses_perm) out.proof.permutation_zs_cap = zs_commitment.tree.cap();
  out.proof.quotient_polys_cap = quotient_commitment.tree.cap();
  out.public_inputs = public_inputs;
  return out;
}

// verifier.rs `verify_stark_proof` (+ get_challenges.rs).
static inline bool verify(const Air& air, const StarkConfig& cfg, const StarkProofWithPublicInputs& pw, const char** why = nullptr) {
  static const char* dummy; if (!why) why = &dummy;
  const StarkProof& proof = pw.proof;
  unsigned degree_bits = pw.degree_bits;
  size_t ncol = air.num_columns();
  auto pairs = air.permutation_pairs();
  bool uses_perm = !pairs.empty();
  size_t nz = uses_perm ? air.num_permutation_batches(cfg) : 0;
  size_t nq = air.quotient_degree_factor() * cfg.num_challenges;
  size_t capn = (size_t)1 << cfg.fri.cap_height;
  // validate_proof_shape
  if (pw.public_inputs.size() != air.num_public_inputs()) { *why = "shape: public inputs"; return false; }
  if (proof.trace_cap.size() != capn || proof.quotient_polys_cap.size() != capn) { *why = "shape: caps"; return false; }
  if (uses_perm != !proof.permutation_zs_cap.empty()) { *why = "shape: perm cap"; return false; }
  if (uses_perm && proof.permutation_zs_cap.size() != capn) { *why = "shape: perm cap size"; return false; }
  const StarkOpeningSet& os = proof.openings;
  if (os.local_values.size() != ncol || os.next_values.size() != ncol || os.permutation_zs.size() != nz ||
      os.permutation_zs_next.size() != nz || os.quotient_polys.size() != nq) { *why = "shape: openings"; return false; }
  // recover_degree_bits cross-check
  if (proof.opening_proof.query_round_proofs.empty()) { *why = "shape: no queries"; return false; }
  {
    size_t sl = proof.opening_proof.query_round_proofs[0].initial[0].siblings.size();
    if (cfg.fri.cap_height + sl - cfg.fri.rate_bits != degree_bits) { *why = "shape: degree bits"; return false; }
  }
  // get_challenges
  Challenger ch;
  ch.observe_cap(proof.trace_cap);
  std::vector<PermChallengeSet> perm_sets;
  if (uses_perm) {
    perm_sets = get_n_permutation_challenge_sets(ch, cfg.num_challenges, air.permutation_batch_size());
    ch.observe_cap(proof.permutation_zs_cap);
  }
  std::vector<GF> alphas;
  for (unsigned i = 0; i < cfg.num_challenges; i++) alphas.push_back(ch.get_challenge());
  ch.observe_cap(proof.quotient_polys_cap);
  Ext zeta = ch.get_ext_challenge();
  FriOpenings fo = to_fri_openings(os);
  for (auto& b : fo.batches) for (auto& e : b) ch.observe(e);
  FriChallenges fc = fri_challenges(ch, proof.opening_proof.commit_phase_merkle_caps, proof.opening_proof.final_poly,
                                    proof.opening_proof.pow_witness, degree_bits, cfg.fri);

  // vanishing poly at zeta
  GF g = gf_root_of_unity(degree_bits);
  Ext zx = ext_exp_pow2(zeta, degree_bits) - GF::one();
  GF nn = GF((u64)1 << degree_bits);
  Ext l0 = zx * ext_inv((zeta - GF::one()) * nn);
  Ext llast = zx * ext_inv((zeta * g - GF::one()) * nn);
  GF last = gf_inv(g);
  std::vector<Ext> ealphas; for (auto a : alphas) ealphas.push_back(Ext(a));
  Consumer<Ext> cons(ealphas, zeta - last, l0, llast);
  std::vector<Ext> epi; for (auto p : pw.public_inputs) epi.push_back(Ext(p));
  air.eval(os.local_values.data(), os.next_values.data(), epi.data(), cons);
  if (uses_perm) {
    auto batches = get_permutation_batches(pairs, perm_sets, cfg.num_challenges, air.permutation_batch_size());
    eval_permutation_checks<Ext>(batches, os.local_values.data(), os.permutation_zs.data(), os.permutation_zs_next.data(), cons);
  }
  Ext zeta_pow_deg = ext_exp_pow2(zeta, degree_bits);
  Ext z_h_zeta = zeta_pow_deg - GF::one();
  unsigned qdf = air.quotient_degree_factor();
  for (size_t i = 0; i < cfg.num_challenges; i++) {
    Ext acc = Ext::zero();
    for (unsigned k = qdf; k-- > 0;) acc = acc * zeta_pow_deg + os.quotient_polys[i * qdf + k];
    if (cons.acc[i] != z_h_zeta * acc) { *why = "quotient mismatch"; return false; }
  }
  std::vector<std::vector<Digest>> caps;
  caps.push_back(proof.trace_cap);
  if (uses_perm) caps.push_back(proof.permutation_zs_cap);
  caps.push_back(proof.quotient_polys_cap);
  return verify_fri_proof(stark_fri_instance(air, cfg, zeta, g), fo, fc, caps, proof.opening_proof,
                          fri_params(cfg.fri, degree_bits), why);
}

// Canonical proof byte layout (LE u64 words; include/sbn.h documents it).
static const u64 PROOF_MAGIC = 0x31564f5250424e53ULL;  // "SNBPROV1"
static inline std::vector<u64> serialize_proof(const StarkProofWithPublicInputs& pw, const StarkConfig& cfg) {
  const StarkProof& p = pw.proof;
  std::vector<u64> w;
  auto dig = [&](const Digest& d) { for (int i = 0; i < 4; i++) w.push_back(d.e[i].v); };
  auto ext = [&](const Ext& e) { w.push_back(e.a.v); w.push_back(e.b.v); };
  const FriProof& f = p.opening_proof;
  w.push_back(PROOF_MAGIC);
  w.push_back(pw.degree_bits);
  w.push_back(p.openings.local_values.size());
  w.push_back(p.openings.permutation_zs.size());
  w.push_back(p.openings.quotient_polys.size());
  w.push_back(pw.public_inputs.size());
  w.push_back(cfg.fri.cap_height);
  w.push_back(cfg.fri.rate_bits);
  w.push_back(f.commit_phase_merkle_caps.size());
  w.push_back(cfg.fri.arity_bits);
  w.push_back(f.final_poly.size());
  w.push_back(f.query_round_proofs.size());
  for (auto& d : p.trace_cap) dig(d);
  for (auto& d : p.permutation_zs_cap) dig(d);
  for (auto& d : p.quotient_polys_cap) dig(d);
  for (auto& e : p.openings.local_values) ext(e);
  for (auto& e : p.openings.next_values) ext(e);
  for (auto& e : p.openings.permutation_zs) ext(e);
  for (auto& e : p.openings.permutation_zs_next) ext(e);
  for (auto& e : p.openings.quotient_polys) ext(e);
  for (auto& cap : f.commit_phase_merkle_caps) for (auto& d : cap) dig(d);
  for (auto& qr : f.query_round_proofs) {
    for (auto& ip : qr.initial) { for (auto& v : ip.evals) w.push_back(v.v); for (auto& d : ip.siblings) dig(d); }
    for (auto& st : qr.steps) { for (auto& e : st.evals) ext(e); for (auto& d : st.siblings) dig(d); }
  }
  for (auto& e : f.final_poly) ext(e);
  w.push_back(f.pow_witness.v);
  for (auto& v : pw.public_inputs) w.push_back(v.v);
  return w;
}

static inline bool deserialize_proof(const u64* w, size_t nw, StarkProofWithPublicInputs& pw) {
  size_t pos = 0;
  auto need = [&](size_t k) { return pos + k <= nw; };
  if (!need(12) || w[0] != PROOF_MAGIC) return false;
  u64 degree_bits = w[1], ncol = w[2], nz = w[3], nq = w[4], npi = w[5], cap_h = w[6], rate_bits = w[7],
      nlayers = w[8], arity_bits = w[9], fpl = w[10], nqueries = w[11];
  pos = 12;
  if (degree_bits > 32 || cap_h > 16 || arity_bits > 8 || nlayers > 16 || nqueries > 1024) return false;
  for (u64 x : {ncol, nz, nq, npi, fpl}) if (x > ((u64)1 << 24)) return false;
  bool ok = true;
  auto rd = [&]() -> GF { if (!need(1)) { ok = false; return GF(); } u64 v = w[pos++]; if (v >= GL_P) ok = false; return GF(v); };
  auto dig = [&]() { Digest d; for (int i = 0; i < 4; i++) d.e[i] = rd(); return d; };
  auto ext = [&]() { GF a = rd(); GF b = rd(); return Ext(a, b); };
  size_t capn = (size_t)1 << cap_h;
  StarkProof& p = pw.proof;
  pw.degree_bits = (unsigned)degree_bits;
  p.trace_cap.clear(); p.permutation_zs_cap.clear(); p.quotient_polys_cap.clear();
  for (size_t i = 0; i < capn && ok; i++) p.trace_cap.push_back(dig());
  if (nz) for (size_t i = 0; i < capn && ok; i++) p.permutation_zs_cap.push_back(dig());
  for (size_t i = 0; i < capn && ok; i++) p.quotient_polys_cap.push_back(dig());
  auto rdv = [&](std::vector<Ext>& v, size_t k) { v.clear(); for (size_t i = 0; i < k && ok; i++) v.push_back(ext()); };
  rdv(p.openings.local_values, ncol); rdv(p.openings.next_values, ncol);
  rdv(p.openings.permutation_zs, nz); rdv(p.openings.permutation_zs_next, nz);
  rdv(p.openings.quotient_polys, nq);
  FriProof& f = p.opening_proof;
  f.commit_phase_merkle_caps.assign(nlayers, {});
  for (auto& cap : f.commit_phase_merkle_caps) for (size_t i = 0; i < capn && ok; i++) cap.push_back(dig());
  size_t lde_bits = degree_bits + rate_bits;
  if (lde_bits < cap_h) return false;
  std::vector<size_t> widths; widths.push_back(ncol); if (nz) widths.push_back(nz); widths.push_back(nq);
  f.query_round_proofs.assign(nqueries, {});
  for (auto& qr : f.query_round_proofs) {
    for (size_t wd : widths) {
      FriInitialTreeProof ip;
      for (size_t i = 0; i < wd && ok; i++) ip.evals.push_back(rd());
      for (size_t i = 0; i < lde_bits - cap_h && ok; i++) ip.siblings.push_back(dig());
      qr.initial.push_back(std::move(ip));
    }
    size_t bits = lde_bits;
    for (size_t l = 0; l < nlayers; l++) {
      if (bits < arity_bits + cap_h) return false;
      bits -= arity_bits;
      FriQueryStep st;
      for (size_t i = 0; i < ((size_t)1 << arity_bits) && ok; i++) st.evals.push_back(ext());
      for (size_t i = 0; i < bits - cap_h && ok; i++) st.siblings.push_back(dig());
      qr.steps.push_back(std::move(st));
    }
    if (!ok) return false;
  }
  rdv(f.final_poly, fpl);
  f.pow_witness = rd();
  pw.public_inputs.clear();
  for (size_t i = 0; i < npi && ok; i++) pw.public_inputs.push_back(rd());
  return ok && pos == nw;
}

}  // namespace orc
