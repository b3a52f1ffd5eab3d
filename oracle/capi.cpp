// ORACLE (test infrastructure, NOT product code): C entry points for the ctypes test harness.
// Built by oracle/Makefile into oracle/_build/liboracle.so.  Nothing under starky_bn254_amd/ may
// load this library.
#include <omp.h>
#include "airs.hpp"
#include <cstdlib>
#include <cstdio>
#include <chrono>

using namespace orc;

static std::unique_ptr<Air> make_air(int kind, size_t num_io) {
  if (kind == 1) return std::unique_ptr<Air>(new G1OpAir());
  if (kind == 2) return std::unique_ptr<Air>(new G1ExpAir(num_io));
  if (kind == 3) return std::unique_ptr<Air>(new G2ExpAir(num_io));
  if (kind == 4) return std::unique_ptr<Air>(new Fq12ExpAir(num_io));
  if (kind == 5) return std::unique_ptr<Air>(new FqExpAir(num_io));
  if (kind == 6) return std::unique_ptr<Air>(new Fq12ExpU64Air(num_io));
  if (kind == 7) return std::unique_ptr<Air>(new ModularAir());
  if (kind == 8) return std::unique_ptr<Air>(new Fq12MulAir());
  if (kind == 9) return std::unique_ptr<Air>(new LookupAir());
  if (kind == 10) return std::unique_ptr<Air>(new FlagAir(num_io));
  if (kind == 11) return std::unique_ptr<Air>(new FlagU64Air(num_io));
  return nullptr;
}
static U256 u256_from_u32(const uint32_t* w) {
  U256 x;
  for (int i = 0; i < 4; i++) x.l[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
  return x;
}

extern "C" {

void orc_poseidon_round_constants(uint64_t* out) { memcpy(out, poseidon_consts().rc, 360 * 8); }
void orc_poseidon_permute_fast(uint64_t* st) {  // sparse partial rounds (what the Merkle hashing uses); must equal the definition
  PState s; for (int i = 0; i < 12; i++) s[i] = GF(st[i]);
  poseidon_permute_fast(s);
  for (int i = 0; i < 12; i++) st[i] = s[i].v;
}
void orc_poseidon_permute(uint64_t* st) {
  PState s; for (int i = 0; i < 12; i++) s[i] = GF(st[i]);
  poseidon_permute(s);
  for (int i = 0; i < 12; i++) st[i] = s[i].v;
}
// eight permutations at once (poseidon_x8.hpp); states: [8][12]; returns 0, or -1 when the CPU lacks AVX-512F/DQ
int orc_poseidon_permute_x8(uint64_t* states) {
#if defined(__x86_64__)
  if (!px8::available()) return -1;
  px8::permute8((uint64_t(*)[12])states);
  return 0;
#else
  (void)states; return -1;
#endif
}
void orc_hash_no_pad(const uint64_t* in, size_t n, uint64_t* out) {
  std::vector<GF> v(n); for (size_t i = 0; i < n; i++) v[i] = GF(in[i]);
  Digest d = hash_or_noop(v.data(), n);
  for (int i = 0; i < 4; i++) out[i] = d.e[i].v;
}
void orc_two_to_one(const uint64_t* l, const uint64_t* r, uint64_t* out) {
  Digest a, b; for (int i = 0; i < 4; i++) { a.e[i] = GF(l[i]); b.e[i] = GF(r[i]); }
  Digest d = two_to_one(a, b);
  for (int i = 0; i < 4; i++) out[i] = d.e[i].v;
}
uint64_t orc_gf_mul(uint64_t a, uint64_t b) { return (GF(a) * GF(b)).v; }
uint64_t orc_gf_inv(uint64_t a) { return gf_inv(GF(a)).v; }
// Challenger transcript probe: observe `n` elements then draw `m` challenges.
void orc_challenger_probe(const uint64_t* in, size_t n, uint64_t* out, size_t m) {
  Challenger ch; for (size_t i = 0; i < n; i++) ch.observe(GF(in[i]));
  for (size_t i = 0; i < m; i++) out[i] = ch.get_challenge().v;
}

// values -> coefficients (in place), natural order.
void orc_ifft(uint64_t* v, size_t n) {
  std::vector<GF> a(n); for (size_t i = 0; i < n; i++) a[i] = GF(v[i]);
  ntt_inplace(a.data(), n, true);
  for (size_t i = 0; i < n; i++) v[i] = a[i].v;
}
// coeffs (n) -> coset LDE values (n << rate_bits), natural order, shift 7.
void orc_coset_lde(const uint64_t* c, size_t n, unsigned rate_bits, uint64_t* out) {
  std::vector<GF> a(n); for (size_t i = 0; i < n; i++) a[i] = GF(c[i]);
  std::vector<GF> r = coset_lde(a, rate_bits, GF(GL_GENERATOR));
  for (size_t i = 0; i < r.size(); i++) out[i] = r[i].v;
}
// PolynomialBatch::from_values on a column-major matrix; returns the Merkle cap (2^cap_height x 4),
// optionally the coefficient matrix (col-major) and the LDE matrix (col-major, natural order).
int orc_commit_values(const uint64_t* cols, size_t ncols, size_t n, unsigned rate_bits, unsigned cap_height,
                      uint64_t* cap_out, uint64_t* coeffs_out, uint64_t* lde_out) {
  std::vector<std::vector<GF>> v(ncols, std::vector<GF>(n));
  for (size_t c = 0; c < ncols; c++) for (size_t i = 0; i < n; i++) v[c][i] = GF(cols[c * n + i]);
  PolynomialBatch b = PolynomialBatch::from_values(v, rate_bits, cap_height);
  const auto& cap = b.tree.cap();
  for (size_t i = 0; i < cap.size(); i++) for (int k = 0; k < 4; k++) cap_out[4 * i + k] = cap[i].e[k].v;
  if (coeffs_out) for (size_t c = 0; c < ncols; c++) for (size_t i = 0; i < n; i++) coeffs_out[c * n + i] = b.polynomials[c][i].v;
  if (lde_out) {
    size_t m = n << rate_bits;
    for (size_t i = 0; i < m; i++) { const auto& row = b.get_lde_values(i, 1); for (size_t c = 0; c < ncols; c++) lde_out[c * m + i] = row[c].v; }
  }
  return 0;
}

size_t orc_air_num_columns(int kind, size_t num_io) { auto a = make_air(kind, num_io); return a ? a->num_columns() : 0; }
size_t orc_air_num_public_inputs(int kind, size_t num_io) { auto a = make_air(kind, num_io); return a ? a->num_public_inputs() : 0; }
size_t orc_air_num_permutation_zs(int kind, size_t num_io) { auto a = make_air(kind, num_io); StarkConfig c; return a ? a->num_permutation_batches(c) : 0; }

// generate_modular_op / generate_modular_zero KAT probe: pol_input[31] (i64) ->
// out[16] out_aux_red[16] quot_abs[17] aux_lo[31] aux_hi[31] sign  (112 words)
void orc_modular_witness(const int64_t* pol_input, int zero_mode, int64_t* out) {
  Arr<int64_t, 31> p; for (int i = 0; i < 31; i++) p[i] = pol_input[i];
  ModWitness w = generate_modular_witness(p, zero_mode != 0);
  int k = 0;
  for (int i = 0; i < 16; i++) out[k++] = w.output[i];
  for (int i = 0; i < 16; i++) out[k++] = w.out_aux_red[i];
  for (int i = 0; i < 17; i++) out[k++] = w.quot_abs[i];
  for (int i = 0; i < 31; i++) out[k++] = w.aux_lo[i];
  for (int i = 0; i < 31; i++) out[k++] = w.aux_hi[i];
  out[k++] = w.quot_sign;
}
// Fq probe: a*b mod p and a^-1 mod p on 8xu32 LE limbs.
void orc_fq_mul(const uint32_t* a, const uint32_t* b, uint32_t* out) {
  U256 r = fq_to_u256(fq_mul(fq_from_u256(u256_from_u32(a)), fq_from_u256(u256_from_u32(b))));
  for (int i = 0; i < 8; i++) out[i] = (uint32_t)(r.l[i / 2] >> (32 * (i % 2)));
}
void orc_fq_inv(const uint32_t* a, uint32_t* out) {
  U256 r = fq_to_u256(fq_inv(fq_from_u256(u256_from_u32(a))));
  for (int i = 0; i < 8; i++) out[i] = (uint32_t)(r.l[i / 2] >> (32 * (i % 2)));
}
// permuted_cols (lookup.rs:60-111)
void orc_permuted_cols(const uint64_t* inputs, const uint64_t* table, size_t n, uint64_t* sorted_out, uint64_t* perm_table_out) {
  std::vector<GF> in(n), tb(n), so, pt;
  for (size_t i = 0; i < n; i++) { in[i] = GF(inputs[i]); tb[i] = GF(table[i]); }
  permuted_cols(in, tb, so, pt);
  for (size_t i = 0; i < n; i++) { sorted_out[i] = so[i].v; perm_table_out[i] = pt[i].v; }
}
// flags table for one exponent: 512 rows x 14 cols, row-major.
void orc_flags_table(const uint32_t* limbs, uint64_t* out) {
  GF lv[NUM_FLAGS_COLS], nv[NUM_FLAGS_COLS];
  generate_flags_first_row(lv, 0, limbs);
  for (int c = 0; c < NUM_FLAGS_COLS; c++) out[c] = lv[c].v;
  for (size_t i = 0; i + 1 < 512; i++) {
    for (auto& x : nv) x = GF();
    generate_flags_next_row(lv, nv, i, 0);
    for (int c = 0; c < NUM_FLAGS_COLS; c++) { out[(i + 1) * NUM_FLAGS_COLS + c] = nv[c].v; lv[c] = nv[c]; }
  }
}

// ios: num_io x 40 u32 = x.x[8] x.y[8] off.x[8] off.y[8] exp_val[8].  trace_out: col-major [ncols][512*num_io].
int orc_g1exp_generate_trace(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out) {
  G1ExpAir air(num_io);
  std::vector<G1ExpIONative> in(num_io);
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* w = ios + 40 * k;
    in[k].x.x = u256_from_u32(w); in[k].x.y = u256_from_u32(w + 8);
    in[k].offset.x = u256_from_u32(w + 16); in[k].offset.y = u256_from_u32(w + 24);
    for (int i = 0; i < 8; i++) in[k].exp_val[i] = w[32 + i];
  }
  auto cols = air.generate_trace(in);
  size_t n = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < n; i++) trace_out[c * n + i] = cols[c][i].v;
  auto pi = air.generate_public_inputs(in);
  for (size_t i = 0; i < pi.size(); i++) pi_out[i] = pi[i].v;
  return 0;
}
// ios: num_io x 72 u32 = x.x.c0 x.x.c1 x.y.c0 x.y.c1 off.x.c0 off.x.c1 off.y.c0 off.y.c1 exp_val (8 u32 each)
int orc_g2exp_generate_trace(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out) {
  G2ExpAir air(num_io);
  std::vector<G2ExpIONative> in(num_io);
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* w = ios + 72 * k;
    auto rd2 = [&](int off) { U256x2 v; v.c0 = u256_from_u32(w + off); v.c1 = u256_from_u32(w + off + 8); return v; };
    in[k].x.x = rd2(0); in[k].x.y = rd2(16); in[k].offset.x = rd2(32); in[k].offset.y = rd2(48);
    for (int i = 0; i < 8; i++) in[k].exp_val[i] = w[64 + i];
  }
  auto cols = air.generate_trace(in);
  size_t n = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < n; i++) trace_out[c * n + i] = cols[c][i].v;
  auto pi = air.generate_public_inputs(in);
  for (size_t i = 0; i < pi.size(); i++) pi_out[i] = pi[i].v;
  return 0;
}
// ios: num_io x 200 u32 = x[12] (8 u32 each, flat basis) offset[12] exp_val[8]
int orc_fq12exp_generate_trace(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out) {
  Fq12ExpAir air(num_io);
  std::vector<Fq12ExpIONative> in(num_io);
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* w = ios + 200 * k;
    for (int c = 0; c < 12; c++) { in[k].x[c] = u256_from_u32(w + 8 * c); in[k].offset[c] = u256_from_u32(w + 96 + 8 * c); }
    for (int i = 0; i < 8; i++) in[k].exp_val[i] = w[192 + i];
  }
  auto cols = air.generate_trace(in);
  size_t n = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < n; i++) trace_out[c * n + i] = cols[c][i].v;
  auto pi = air.generate_public_inputs(in);
  for (size_t i = 0; i < pi.size(); i++) pi_out[i] = pi[i].v;
  return 0;
}
// ios: num_io x 194 u32 = x[12] (8 u32 each, flat basis) offset[12] exp_val (lo, hi)
int orc_fq12expu64_generate_trace(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out) {
  Fq12ExpU64Air air(num_io);
  std::vector<Fq12ExpU64IONative> in(num_io);
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* w = ios + 194 * k;
    for (int c = 0; c < 12; c++) { in[k].x[c] = u256_from_u32(w + 8 * c); in[k].offset[c] = u256_from_u32(w + 96 + 8 * c); }
    in[k].exp_val = (uint64_t)w[192] | ((uint64_t)w[193] << 32);
  }
  auto cols = air.generate_trace(in);
  size_t n = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < n; i++) trace_out[c * n + i] = cols[c][i].v;
  auto pi = air.generate_public_inputs(in);
  for (size_t i = 0; i < pi.size(); i++) pi_out[i] = pi[i].v;
  return 0;
}
// ios: num_io x 24 u32 = x[8] offset[8] exp_val[8]
int orc_fqexp_generate_trace(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out) {
  FqExpAir air(num_io);
  std::vector<FqExpIONative> in(num_io);
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* w = ios + 24 * k;
    in[k].x = u256_from_u32(w); in[k].offset = u256_from_u32(w + 8);
    for (int i = 0; i < 8; i++) in[k].exp_val[i] = w[16 + i];
  }
  auto cols = air.generate_trace(in);
  size_t n = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < n; i++) trace_out[c * n + i] = cols[c][i].v;
  auto pi = air.generate_public_inputs(in);
  for (size_t i = 0; i < pi.size(); i++) pi_out[i] = pi[i].v;
  return 0;
}
// pts: rows x 32 u32 = a.x a.y b.x b.y
int orc_g1op_generate_trace(const uint32_t* pts, size_t rows, uint64_t* trace_out) {
  G1OpAir air;
  std::vector<G1Affine> a(rows), b(rows);
  for (size_t r = 0; r < rows; r++) {
    const uint32_t* w = pts + 32 * r;
    a[r].x = u256_from_u32(w); a[r].y = u256_from_u32(w + 8); b[r].x = u256_from_u32(w + 16); b[r].y = u256_from_u32(w + 24);
  }
  auto cols = air.generate_trace(a, b);
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}

// ops: rows x 16 u32 = a b (each < p)
int orc_modular_generate_trace(const uint32_t* ops, size_t rows, uint64_t* trace_out) {
  ModularAir air;
  std::vector<U256> a(rows), b(rows);
  for (size_t r = 0; r < rows; r++) { a[r] = u256_from_u32(ops + 16 * r); b[r] = u256_from_u32(ops + 16 * r + 8); }
  auto cols = air.generate_trace(a, b);
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}
// ops: rows x 192 u32 = x[12] y[12] (coefficients < p, flat basis of fq12_to_columns)
int orc_fq12mul_generate_trace(const uint32_t* ops, size_t rows, uint64_t* trace_out) {
  Fq12MulAir air;
  std::vector<std::array<U256, 12>> x(rows), y(rows);
  for (size_t r = 0; r < rows; r++)
    for (int i = 0; i < 12; i++) { x[r][i] = u256_from_u32(ops + 192 * r + 8 * i); y[r][i] = u256_from_u32(ops + 192 * r + 96 + 8 * i); }
  auto cols = air.generate_trace(x, y);
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}

// MyStark (lookup.rs:136-213): inputs / table of `rows` values each -> the 4-column trace
int orc_lookup_generate_trace(const uint64_t* inputs, const uint64_t* table, size_t rows, uint64_t* trace_out) {
  LookupAir air;
  std::vector<GF> in(rows), tab(rows);
  for (size_t i = 0; i < rows; i++) { in[i] = GF(inputs[i]); tab[i] = GF(table[i]); }
  auto cols = air.generate_trace(in, tab);
  for (size_t c = 0; c < 4; c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}

// FlagStark (flags.rs:392-440): limbs = num_io x 8 u32 -> the [17 + 4 num_io][512 num_io] trace
int orc_flags_generate_trace(const uint32_t* limbs, size_t num_io, uint64_t* trace_out) {
  FlagAir air(num_io);
  std::vector<std::array<uint32_t, NUM_INPUT_LIMBS>> in(num_io);
  for (size_t k = 0; k < num_io; k++) for (int i = 0; i < NUM_INPUT_LIMBS; i++) in[k][i] = limbs[8 * k + i];
  auto cols = air.generate_trace(in);
  const size_t rows = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}

int orc_flags_u64_generate_trace(const uint64_t* exps, size_t num_io, uint64_t* trace_out) {
  FlagU64Air air(num_io);
  auto cols = air.generate_trace(std::vector<uint64_t>(exps, exps + num_io));
  const size_t rows = cols[0].size();
  for (size_t c = 0; c < cols.size(); c++) for (size_t i = 0; i < rows; i++) trace_out[c * rows + i] = cols[c][i].v;
  return 0;
}

// Switch of this restatement (oracle/fri.hpp FriConfig::final_poly_times_x): 1 = plonky2 0.1.x "multiply the final
// polynomial by X" step (default), 0 = the later upstream form without it.
static int g_final_poly_times_x = 1;
void orc_set_final_poly_times_x(int on) { g_final_poly_times_x = on ? 1 : 0; }

// prove(): trace col-major [ncols][1<<degree_bits]; returns malloc'd canonical proof words.
int orc_prove(int kind, size_t num_io, const uint64_t* trace, unsigned degree_bits, const uint64_t* pi, size_t npi,
              uint64_t** proof_out, size_t* nwords_out, double* seconds_out) {
  auto air = make_air(kind, num_io);
  if (!air) return -1;
  size_t n = (size_t)1 << degree_bits, ncols = air->num_columns();
  if (npi != air->num_public_inputs()) return -2;
  std::vector<std::vector<GF>> cols(ncols, std::vector<GF>(n));
  for (size_t c = 0; c < ncols; c++) for (size_t i = 0; i < n; i++) { if (trace[c * n + i] >= GL_P) return -3; cols[c][i] = GF(trace[c * n + i]); }
  std::vector<GF> pis(npi); for (size_t i = 0; i < npi; i++) pis[i] = GF(pi[i]);
  StarkConfig cfg;
  cfg.fri.final_poly_times_x = g_final_poly_times_x != 0;
  auto t0 = std::chrono::steady_clock::now();
  StarkProofWithPublicInputs p = prove(*air, cfg, cols, pis);
  auto t1 = std::chrono::steady_clock::now();
  if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
  std::vector<u64> w = serialize_proof(p, cfg);
  *proof_out = (uint64_t*)malloc(w.size() * 8);
  memcpy(*proof_out, w.data(), w.size() * 8);
  *nwords_out = w.size();
  return 0;
}
void orc_free(void* p) { free(p); }
// OpenMP threads of the following calls (tools/oracle_scaling.py); 0 = leave as is.  Returns the maximum available.
int orc_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
// Stage k of the last orc_prove: name (static storage until the next prove) and wall seconds; returns the stage count.
int orc_last_stage_seconds(int k, const char** name, double* seconds) {
  static std::vector<std::pair<std::string, double>> v;   // stages, then the phases inside the commitments ("commit: ...")
  if (k < 0) { v = stage_log(); v.insert(v.end(), phase_log().begin(), phase_log().end()); }
  if (k >= 0 && (size_t)k < v.size()) { if (name) *name = v[k].first.c_str(); if (seconds) *seconds = v[k].second; }
  return (int)v.size();
}

// 0 = accepted; negative = rejected (message in *why).
int orc_verify(int kind, size_t num_io, const uint64_t* proof, size_t nwords, const char** why) {
  static const char* w0 = ""; if (why) *why = w0;
  auto air = make_air(kind, num_io);
  if (!air) { if (why) *why = "unknown air"; return -1; }
  StarkProofWithPublicInputs p;
  if (!deserialize_proof(proof, nwords, p)) { if (why) *why = "malformed proof bytes"; return -2; }
  StarkConfig cfg;
  cfg.fri.final_poly_times_x = g_final_poly_times_x != 0;
  const char* reason = "";
  if (!verify(*air, cfg, p, &reason)) { if (why) *why = reason; return -3; }
  return 0;
}

// Constraint probe: evaluate the AIR's constraints (no permutation checks) on given local/next rows
// with base-field alphas; returns the accumulators.  Used to test "a valid trace satisfies every
// constraint on the trace domain".
int orc_eval_constraints(int kind, size_t num_io, const uint64_t* lv, const uint64_t* nv, const uint64_t* pi,
                         const uint64_t* alphas, size_t nalpha, uint64_t z_last, uint64_t l_first, uint64_t l_last, uint64_t* acc_out) {
  auto air = make_air(kind, num_io);
  if (!air) return -1;
  size_t ncols = air->num_columns(), npi = air->num_public_inputs();
  std::vector<GF> l(ncols), n(ncols), p(npi), al(nalpha);
  for (size_t i = 0; i < ncols; i++) { l[i] = GF(lv[i]); n[i] = GF(nv[i]); }
  for (size_t i = 0; i < npi; i++) p[i] = GF(pi[i]);
  for (size_t i = 0; i < nalpha; i++) al[i] = GF(alphas[i]);
  Consumer<GF> c(al, GF(z_last), GF(l_first), GF(l_last));
  air->eval(l.data(), n.data(), p.data(), c);
  for (size_t i = 0; i < nalpha; i++) acc_out[i] = c.acc[i].v;
  return 0;
}

}  // extern "C"
