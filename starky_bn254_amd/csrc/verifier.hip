// Host verifier: replaces starky verifier.rs `verify_stark_proof` (+ get_challenges.rs) and plonky2
// fri/verifier.rs `verify_fri_proof` for the reference's call at src/curves/g1/exp.rs:826.
// The constraint code is the SAME templates the device quotient kernel runs (air.cuh), instantiated
// over the quadratic extension -- as the reference's verifier reuses eval_packed_generic with
// P = F::Extension.  Verification is a few thousand field operations and 84 Merkle paths; it runs on
// the calling thread like the reference's.
#include "host_common.hpp"
#include <cstring>

using namespace sbn;

namespace {

struct HostRow {
  const E2* lv; const E2* nv;
  E2 l(int c) const { return lv[c]; }
  E2 n(int c) const { return nv[c]; }
};
struct HostZRow {
  const E2* lz; const E2* nz;
  E2 zl(int z) const { return lz[z]; }
  E2 zn(int z) const { return nz[z]; }
};

struct Reader {
  const u64* w; size_t n, pos = 0; bool ok = true;
  u64 word() { if (pos >= n) { ok = false; return 0; } return w[pos++]; }
  F f() { u64 v = word(); if (v >= GLP) ok = false; return F(v >= GLP ? 0 : v); }
  E2 e() { F a = f(); F b = f(); return E2(a, b); }
  Digest4 d() { Digest4 r; for (int i = 0; i < 4; i++) r.e[i] = f(); return r; }
};

Digest4 hash_or_noop(const F* v, size_t n) {
  Digest4 d;
  if (n <= 4) { for (int i = 0; i < 4; i++) d.e[i] = i < (int)n ? v[i] : F(0); return d; }
  F s[12]; for (auto& x : s) x = F(0);
  for (size_t off = 0; off < n; off += 8) {
    size_t len = n - off < 8 ? n - off : 8;
    for (size_t i = 0; i < len; i++) s[i] = v[off + i];
    poseidon_permute(s);
  }
  for (int i = 0; i < 4; i++) d.e[i] = s[i];
  return d;
}
bool digest_eq(const Digest4& a, const Digest4& b) { for (int i = 0; i < 4; i++) if (a.e[i] != b.e[i]) return false; return true; }
// merkle_proofs.rs `verify_merkle_proof_to_cap`
bool merkle_verify(const F* leaf, size_t leaf_len, size_t index, const std::vector<Digest4>& cap, const std::vector<Digest4>& sib) {
  Digest4 cur = hash_or_noop(leaf, leaf_len);
  for (auto& s : sib) { cur = (index & 1) ? poseidon_two_to_one(s, cur) : poseidon_two_to_one(cur, s); index >>= 1; }
  return index < cap.size() && digest_eq(cur, cap[index]);
}

// fri/verifier.rs `compute_evaluation`: interpolate on the coset and evaluate at beta.
E2 compute_evaluation(F x, size_t within, u32 arity_bits, const std::vector<E2>& evals_in, E2 beta) {
  size_t arity = (size_t)1 << arity_bits;
  F g = f_root_of_unity(arity_bits);
  std::vector<E2> evals(arity);
  for (size_t i = 0; i < arity; i++) evals[bitrev32((u32)i, arity_bits)] = evals_in[i];
  size_t rev = bitrev32((u32)within, arity_bits);
  F start = x * f_pow(g, arity - rev);
  std::vector<F> pts(arity);
  { F y(1); for (size_t i = 0; i < arity; i++) { pts[i] = start * y; y = y * g; } }
  E2 res{F(0), F(0)};
  for (size_t i = 0; i < arity; i++) {
    E2 num{F(1), F(0)}; F den(1);
    for (size_t j = 0; j < arity; j++) if (j != i) { num = num * (beta - pts[j]); den = den * (pts[i] - pts[j]); }
    res = res + evals[i] * num * f_inv(den);
  }
  return res;
}

}  // namespace

extern "C" int sbn_verify(const sbn_air_desc* air, const sbn_config* cfg, const uint8_t* bytes, size_t len) {
  if (!air || !cfg || !bytes) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (!config_supported(cfg)) return fail(SBN_ERR_UNSUPPORTED, "unsupported StarkConfig");
  AirShape as;
  if (!air_shape(air, cfg, as)) return fail(SBN_ERR_BAD_ARG, "unknown air kind / num_io");
  if (len % 8 || len < 12 * 8) return fail(SBN_ERR_MALFORMED_PROOF, "proof length is not a whole number of words");
  std::vector<u64> words(len / 8);
  memcpy(words.data(), bytes, len);  // LE host
  Reader rd{words.data(), words.size()};
  // header + validate_proof_shape
  u64 magic = rd.word(), degree_bits = rd.word(), ncol = rd.word(), nz = rd.word(), nq = rd.word(), npi = rd.word(), cap_h = rd.word(),
      rate_bits = rd.word(), nlayers = rd.word(), arity_bits = rd.word(), fpl = rd.word(), nqueries = rd.word();
  if (magic != PROOF_MAGIC) return fail(SBN_ERR_MALFORMED_PROOF, "bad magic");
  if (degree_bits < 1 || degree_bits > 30) return fail(SBN_ERR_MALFORMED_PROOF, "bad degree_bits");
  FriShape fs = fri_shape(*cfg, (u32)degree_bits);
  if (ncol != as.ncols || nz != as.nzs || nq != 2 * cfg->num_challenges || npi != as.npi || cap_h != cfg->cap_height || rate_bits != cfg->rate_bits ||
      nlayers != fs.arity_bits.size() || arity_bits != cfg->fri_arity_bits || fpl != fs.final_poly_len() || nqueries != cfg->num_query_rounds)
    return fail(SBN_ERR_MALFORMED_PROOF, "proof shape does not match the table / config");
  if ((is_exp_air(as.kind) || as.kind == SBN_AIR_FLAGS || as.kind == SBN_AIR_FLAGS_U64) && ((u64)exp_rows_per_instance(as.kind) * as.num_io) != ((u64)1 << degree_bits)) return fail(SBN_ERR_MALFORMED_PROOF, "degree_bits does not match num_io");
  const u32 lde_bits = (u32)degree_bits + cfg->rate_bits;
  if (lde_bits < cfg->cap_height + fs.total_arity()) return fail(SBN_ERR_MALFORMED_PROOF, "degree too small for the FRI parameters");
  const size_t capn = (size_t)1 << cfg->cap_height;
  auto read_cap = [&](std::vector<Digest4>& c) { c.resize(capn); for (auto& d : c) d = rd.d(); };
  std::vector<Digest4> trace_cap, z_cap, q_cap;
  read_cap(trace_cap); if (nz) read_cap(z_cap); read_cap(q_cap);
  auto read_ext = [&](std::vector<E2>& v, size_t k) { v.resize(k); for (auto& e : v) e = rd.e(); };
  std::vector<E2> local, next, zs, zs_next, quot;
  read_ext(local, ncol); read_ext(next, ncol); read_ext(zs, nz); read_ext(zs_next, nz); read_ext(quot, nq);
  std::vector<std::vector<Digest4>> fri_caps(nlayers);
  for (auto& c : fri_caps) read_cap(c);
  if (!rd.ok) return fail(SBN_ERR_MALFORMED_PROOF, "truncated proof or non-canonical element");
  struct Initial { std::vector<F> evals; std::vector<Digest4> sib; };
  struct Step { std::vector<E2> evals; std::vector<Digest4> sib; };
  struct Round { std::vector<Initial> init; std::vector<Step> steps; };
  std::vector<size_t> widths; widths.push_back(ncol); if (nz) widths.push_back(nz); widths.push_back(nq);
  std::vector<Round> rounds(nqueries);
  for (auto& r : rounds) {
    for (size_t wd : widths) {
      Initial in; in.evals.resize(wd); for (auto& v : in.evals) v = rd.f();
      in.sib.resize(lde_bits - cfg->cap_height); for (auto& d : in.sib) d = rd.d();
      r.init.push_back(std::move(in));
    }
    u32 bits = lde_bits;
    for (u32 ab : fs.arity_bits) {
      bits -= ab;
      Step s; s.evals.resize((size_t)1 << ab); for (auto& e : s.evals) e = rd.e();
      s.sib.resize(bits - cfg->cap_height); for (auto& d : s.sib) d = rd.d();
      r.steps.push_back(std::move(s));
    }
    if (!rd.ok) return fail(SBN_ERR_MALFORMED_PROOF, "truncated proof or non-canonical element");
  }
  std::vector<E2> final_poly; read_ext(final_poly, fpl);
  F pow_witness = rd.f();
  std::vector<F> pi(npi); for (auto& v : pi) v = rd.f();
  if (!rd.ok || rd.pos != words.size()) return fail(SBN_ERR_MALFORMED_PROOF, "truncated proof, trailing words or non-canonical element");

  // get_challenges (starky get_challenges.rs)
  Challenger ch;
  for (auto& d : trace_cap) for (int i = 0; i < 4; i++) ch.observe(d.e[i]);
  F gam[2][2] = {};
  if (nz) {
    for (int s = 0; s < 2; s++) for (u32 c = 0; c < cfg->num_challenges; c++) { (void)ch.challenge(); gam[s][c] = ch.challenge(); }
    for (auto& d : z_cap) for (int i = 0; i < 4; i++) ch.observe(d.e[i]);
  }
  F alphas[SBN_NCH];
  for (int j = 0; j < SBN_NCH; j++) alphas[j] = ch.challenge();
  for (auto& d : q_cap) for (int i = 0; i < 4; i++) ch.observe(d.e[i]);
  E2 zeta = ch.ext_challenge();
  for (auto& e : local) ch.observe(e);
  for (auto& e : zs) ch.observe(e);
  for (auto& e : quot) ch.observe(e);
  for (auto& e : next) ch.observe(e);
  for (auto& e : zs_next) ch.observe(e);
  E2 fri_alpha = ch.ext_challenge();
  std::vector<E2> betas;
  for (auto& c : fri_caps) { for (auto& d : c) for (int i = 0; i < 4; i++) ch.observe(d.e[i]); betas.push_back(ch.ext_challenge()); }
  for (auto& e : final_poly) ch.observe(e);
  ch.observe(pow_witness);
  F pow_response = ch.challenge();
  std::vector<size_t> indices(nqueries);
  for (auto& x : indices) x = (size_t)(ch.challenge().v % ((u64)1 << lde_bits));

  // vanishing polynomial at zeta (verifier.rs verify_stark_proof_with_challenges)
  F g = f_root_of_unity((u32)degree_bits);
  E2 zeta_pow_deg = e2_exp_pow2(zeta, (u32)degree_bits);
  E2 z_h_zeta = zeta_pow_deg - F(1);
  F nn = F((u64)1 << degree_bits);
  Cons<E2> cs;
  std::vector<E2> apow[SBN_NCH];
  for (int j = 0; j < SBN_NCH; j++) {
    apow[j].resize(apow_len(as.nconstraints, as.nzs));
    E2 a{F(1), F(0)}, al(alphas[j]);
    for (size_t k = 0; k < apow[j].size(); k++) { apow[j][k] = a; a = a * al; }
    cs.alpha[j] = al; cs.apow[j] = apow[j].data();
  }
  cs.start((int)(as.nconstraints + 2 * nz));   // the AIR's constraints, then nz first-row and nz transition permutation checks
  cs.z_last = zeta - f_inv(g);
  cs.l_first = z_h_zeta * e2_inv((zeta - F(1)) * nn);        // eval_l_0_and_l_last
  cs.l_last = z_h_zeta * e2_inv((zeta * g - F(1)) * nn);
  HostRow row{local.data(), next.data()};
  HostZRow zrow{zs.data(), zs_next.data()};
  E2 g0(gam[0][0]), g1(gam[1][1]);
  if (as.kind == SBN_AIR_G1_OP) {
    g1op_eval(cs, row);
    permutation_checks(cs, row, zrow, G1OpShape(), (int)nz, g0, g1);
  } else if (is_op_air(as.kind)) {
    const OpShape sh(as.kind);
    if (as.kind == SBN_AIR_FQ12_MUL) op_eval<8>(cs, row, sh); else op_eval<7>(cs, row, sh);
    permutation_checks(cs, row, zrow, sh, (int)nz, g0, g1);
  } else if (as.kind == SBN_AIR_LOOKUP) {
    lookup_eval(cs, row);
    permutation_checks(cs, row, zrow, LookupShape(), (int)nz, g0, g1);
  } else if (as.kind == SBN_AIR_FLAGS) {
    flag_eval(cs, row, FlagShape((int)as.num_io));   // no permutation pairs: nz = 0, no Z cap, no permutation challenges
  } else if (as.kind == SBN_AIR_FLAGS_U64) {
    flag_u64_eval(cs, row, FlagU64Shape((int)as.num_io));
  } else {
    ExpShape sh = exp_shape(as);
    std::vector<E2> epi(npi); for (size_t i = 0; i < npi; i++) epi[i] = E2(pi[i]);
    static thread_local ExpPiConsts<E2> pic;
    const E2* app[SBN_NCH] = {apow[0].data(), apow[1].data()};
    exp_pi_consts<E2>(sh, app, epi.data(), pic);
    if (sh.E == 1) exp_eval<1>(cs, row, sh, &pic); else if (sh.E == 2) exp_eval<2>(cs, row, sh, &pic); else if (sh.E == 0) exp_eval<0>(cs, row, sh, &pic); else if (sh.E == 13) exp_eval<13>(cs, row, sh, &pic); else exp_eval<12>(cs, row, sh, &pic);
    permutation_checks(cs, row, zrow, sh, (int)nz, g0, g1);
  }
  for (u32 i = 0; i < cfg->num_challenges; i++) {
    // quotient_degree_factor = 2 chunks per challenge: t(zeta) = t0 + zeta^N t1
    E2 t = quot[2 * i] + zeta_pow_deg * quot[2 * i + 1];
    if (cs.rem != 0) return fail(SBN_ERR_VERIFY_FAILED, "internal: constraint count mismatch (%d left)", cs.rem);
    if (cs.result((int)i) != z_h_zeta * t) return fail(SBN_ERR_VERIFY_FAILED, "mismatch between evaluation and opening of quotient polynomial");
  }

  // verify_fri_proof
  {
    u32 lz = pow_response.v ? (u32)__builtin_clzll(pow_response.v) : 64;
    if (lz < cfg->proof_of_work_bits) return fail(SBN_ERR_VERIFY_FAILED, "invalid proof of work");
  }
  // batch zeta: trace ++ zs ++ quotient ; batch g*zeta: trace ++ zs
  E2 zeta_next = zeta * g;
  auto reduce = [&](const std::vector<const std::vector<E2>*>& parts) {
    E2 acc{F(0), F(0)};
    for (size_t p = parts.size(); p-- > 0;) for (size_t j = parts[p]->size(); j-- > 0;) acc = acc * fri_alpha + (*parts[p])[j];
    return acc;
  };
  E2 red0 = reduce({&local, &zs, &quot}), red1 = reduce({&next, &zs_next});
  E2 shift1 = e2_pow(fri_alpha, ncol + nz);
  std::vector<std::vector<Digest4>> init_caps; init_caps.push_back(trace_cap); if (nz) init_caps.push_back(z_cap); init_caps.push_back(q_cap);
  for (size_t q = 0; q < nqueries; q++) {
    size_t x_index = indices[q];
    const Round& r = rounds[q];
    for (size_t t = 0; t < init_caps.size(); t++)
      if (!merkle_verify(r.init[t].evals.data(), r.init[t].evals.size(), x_index, init_caps[t], r.init[t].sib))
        return fail(SBN_ERR_VERIFY_FAILED, "invalid Merkle proof (initial oracle %zu, query %zu)", t, q);
    F subgroup_x = F(GL_GEN) * f_pow(f_root_of_unity(lde_bits), bitrev32((u32)x_index, lde_bits));
    // fri_combine_initial
    E2 e0{F(0), F(0)}, e1(F(0), F(0));
    for (size_t t = init_caps.size(); t-- > 0;) for (size_t j = r.init[t].evals.size(); j-- > 0;) e0 = e0 * fri_alpha + r.init[t].evals[j];
    for (size_t t = init_caps.size() - 1; t-- > 0;) for (size_t j = r.init[t].evals.size(); j-- > 0;) e1 = e1 * fri_alpha + r.init[t].evals[j];
    E2 sum = (e0 - red0) * e2_inv(E2(subgroup_x) - zeta);
    sum = sum * shift1 + (e1 - red1) * e2_inv(E2(subgroup_x) - zeta_next);
    // plonky2 0.1.x fri_combine_initial: "Multiply the final polynomial by `X`" (PR #436): `sum * subgroup_x`
    E2 old_eval = fri_times_x(*cfg) ? sum * subgroup_x : sum;
    for (size_t i = 0; i < fs.arity_bits.size(); i++) {
      u32 ab = fs.arity_bits[i];
      size_t arity = (size_t)1 << ab, coset_index = x_index >> ab, within = x_index & (arity - 1);
      const Step& s = r.steps[i];
      if (s.evals[within] != old_eval) return fail(SBN_ERR_VERIFY_FAILED, "FRI fold consistency check failed (query %zu, layer %zu)", q, i);
      old_eval = compute_evaluation(subgroup_x, within, ab, s.evals, betas[i]);
      std::vector<F> flat(2 * arity);
      for (size_t j = 0; j < arity; j++) { flat[2 * j] = s.evals[j].a; flat[2 * j + 1] = s.evals[j].b; }
      if (!merkle_verify(flat.data(), flat.size(), coset_index, fri_caps[i], s.sib))
        return fail(SBN_ERR_VERIFY_FAILED, "invalid Merkle proof (FRI layer %zu, query %zu)", i, q);
      subgroup_x = f_exp_pow2(subgroup_x, ab);
      x_index = coset_index;
    }
    E2 fe{F(0), F(0)};
    for (size_t j = final_poly.size(); j-- > 0;) fe = fe * E2(subgroup_x) + final_poly[j];
    if (fe != old_eval) return fail(SBN_ERR_VERIFY_FAILED, "final polynomial evaluation is invalid (query %zu)", q);
  }
  return SBN_OK;
}

// Parity hook (include/sbn.h): the table's AIR constraints -- the regrouped templates of air.cuh, here over the base
// field as in the quotient kernel -- folded into the two Horner accumulators on ONE (local, next) row pair.  No
// permutation checks.  Host only; tests compare it with the oracle's constraint-by-constraint evaluation on random rows.
namespace {
struct HostRowF {
  const u64* lv; const u64* nv;
  F l(int c) const { return F(lv[c]); }
  F n(int c) const { return F(nv[c]); }
};
}  // namespace
extern "C" int sbn_eval_constraints_host(const sbn_air_desc* air, const uint64_t* local_row, const uint64_t* next_row, const uint64_t* public_inputs,
                                         size_t n_pi, const uint64_t* alphas, uint64_t z_last, uint64_t l_first, uint64_t l_last, uint64_t* acc_out) {
  if (!air || !local_row || !next_row || !alphas || !acc_out) return fail(SBN_ERR_BAD_ARG, "null argument");
  AirShape as;
  if (!air_shape(air, nullptr, as)) return fail(SBN_ERR_BAD_ARG, "unknown air kind / num_io");
  if (n_pi != as.npi || (n_pi && !public_inputs)) return fail(SBN_ERR_BAD_ARG, "expected %zu public inputs", as.npi);
  for (size_t c = 0; c < as.ncols; c++) if (local_row[c] >= GLP || next_row[c] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "row word %zu is not canonical", c);
  for (size_t i = 0; i < n_pi; i++) if (public_inputs[i] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "public input %zu is not canonical", i);
  if (z_last >= GLP || l_first >= GLP || l_last >= GLP || alphas[0] >= GLP || alphas[1] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "non-canonical scalar");
  Cons<F> cs;
  std::vector<F> apow[SBN_NCH];
  for (int j = 0; j < SBN_NCH; j++) {
    apow[j].resize(apow_len(as.nconstraints, as.nzs));
    F a(1), al(alphas[j]);
    for (size_t k = 0; k < apow[j].size(); k++) { apow[j][k] = a; a = a * al; }
    cs.alpha[j] = al; cs.apow[j] = apow[j].data();
  }
  cs.start((int)as.nconstraints);
  cs.z_last = F(z_last); cs.l_first = F(l_first); cs.l_last = F(l_last);
  HostRowF row{local_row, next_row};
  if (as.kind == SBN_AIR_G1_OP) {
    g1op_eval(cs, row);
  } else if (is_op_air(as.kind)) {
    const OpShape sh(as.kind);
    if (as.kind == SBN_AIR_FQ12_MUL) op_eval<8>(cs, row, sh); else op_eval<7>(cs, row, sh);
  } else if (as.kind == SBN_AIR_LOOKUP) {
    lookup_eval(cs, row);
  } else if (as.kind == SBN_AIR_FLAGS) {
    flag_eval(cs, row, FlagShape((int)as.num_io));
  } else if (as.kind == SBN_AIR_FLAGS_U64) {
    flag_u64_eval(cs, row, FlagU64Shape((int)as.num_io));
  } else {
    ExpShape sh = exp_shape(as);
    std::vector<F> fpi(n_pi); for (size_t i = 0; i < n_pi; i++) fpi[i] = F(public_inputs[i]);
    static thread_local ExpPiConsts<F> pic;
    const F* app[SBN_NCH] = {apow[0].data(), apow[1].data()};
    exp_pi_consts<F>(sh, app, fpi.data(), pic);
    if (sh.E == 1) exp_eval<1>(cs, row, sh, &pic); else if (sh.E == 2) exp_eval<2>(cs, row, sh, &pic); else if (sh.E == 0) exp_eval<0>(cs, row, sh, &pic);
    else if (sh.E == 13) exp_eval<13>(cs, row, sh, &pic); else exp_eval<12>(cs, row, sh, &pic);
  }
  if (cs.rem != 0) return fail(SBN_ERR_HIP, "internal: the evaluator emitted %d constraints fewer than the table declares", cs.rem);
  for (int j = 0; j < SBN_NCH; j++) acc_out[j] = cs.result(j).v;
  return SBN_OK;
}
