// Host-side pieces shared by the prover and verifier: error reporting, the Fiat-Shamir transcript
// (plonky2 iop/challenger.rs `Challenger`, duplex/overwrite mode), FRI parameters
// (fri/reduction_strategies.rs ConstantArityBits) and the table shapes.
#pragma once
#include "../../include/sbn.h"
#include "poseidon.cuh"
#include "air.cuh"
#include <string>
#include <vector>
#include <cstdarg>
#include <cstdio>

namespace sbn {

extern thread_local std::string g_last_error;
static inline int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_last_error = buf;
  return code;
}

static constexpr u64 PROOF_MAGIC = 0x31564f5250424e53ULL;  // "SNBPROV1"

struct Challenger {
  F st[12];
  std::vector<F> in, out;
  Challenger() { for (auto& x : st) x = F(0); }
  void duplex() {
    for (size_t i = 0; i < in.size(); i++) st[i] = in[i];
    in.clear();
    poseidon_permute(st);
    out.assign(st, st + P_RATE);
  }
  void observe(F e) { out.clear(); in.push_back(e); if (in.size() == (size_t)P_RATE) duplex(); }
  void observe(E2 e) { observe(e.a); observe(e.b); }
  void observe_words(const u64* w, size_t n) { for (size_t i = 0; i < n; i++) observe(F(w[i])); }
  F challenge() {
    if (!in.empty() || out.empty()) duplex();
    F r = out.back(); out.pop_back(); return r;
  }
  E2 ext_challenge() { F a = challenge(); F b = challenge(); return E2(a, b); }
};

struct FriShape {
  std::vector<u32> arity_bits;  // per reduction layer
  u32 degree_bits, rate_bits, cap_height;
  u32 lde_bits() const { return degree_bits + rate_bits; }
  u32 total_arity() const { u32 s = 0; for (auto a : arity_bits) s += a; return s; }
  size_t final_poly_len() const { return (size_t)1 << (degree_bits - total_arity()); }
};
static inline FriShape fri_shape(const sbn_config& c, u32 degree_bits) {
  FriShape f; f.degree_bits = degree_bits; f.rate_bits = c.rate_bits; f.cap_height = c.cap_height;
  u32 d = degree_bits;
  while (d > c.fri_final_poly_bits && d + c.rate_bits - c.fri_arity_bits >= c.cap_height) { f.arity_bits.push_back(c.fri_arity_bits); d -= c.fri_arity_bits; }
  return f;
}

struct AirShape {
  int kind; u32 num_io;
  size_t ncols, npi, npairs, nzs, nconstraints;
};
static inline int exp_e(int kind) { return kind == SBN_AIR_FQ12_EXP ? 12 : (kind == SBN_AIR_FQ12_EXP_U64 ? 13 : (kind == SBN_AIR_G2_EXP ? 2 : (kind == SBN_AIR_FQ_EXP ? 0 : 1))); }
static inline bool air_shape(const sbn_air_desc* air, const sbn_config* cfg, AirShape& s) {
  if (!air) return false;
  s.kind = air->kind; s.num_io = air->num_io;
  u32 nch = cfg ? cfg->num_challenges : 2;
  if (air->kind == SBN_AIR_G1_OP) {
    s.ncols = G1OpShape::NUM_COLS; s.npi = 0; s.npairs = G1OpShape::NUM_PAIRS; s.nconstraints = G1OpShape::NUM_CONSTRAINTS;
  } else if (air->kind == SBN_AIR_MODULAR || air->kind == SBN_AIR_FQ12_MUL) {
    OpShape sh(air->kind);
    s.ncols = sh.num_cols(); s.npi = 0; s.npairs = sh.num_pairs(); s.nconstraints = sh.num_constraints();
  } else if (air->kind == SBN_AIR_LOOKUP) {
    s.ncols = LookupShape::NUM_COLS; s.npi = 0; s.npairs = LookupShape::NUM_PAIRS; s.nconstraints = LookupShape::NUM_CONSTRAINTS;
  } else if (air->kind == SBN_AIR_FLAGS) {
    if (air->num_io == 0 || air->num_io > (u32)G1EXP_MAX_IO || (air->num_io & (air->num_io - 1))) return false;
    FlagShape sh((int)air->num_io);
    s.ncols = sh.num_cols(); s.npi = 0; s.npairs = 0; s.nconstraints = sh.num_constraints();
  } else if (air->kind == SBN_AIR_FLAGS_U64) {
    if (air->num_io < 4 || air->num_io > 4 * (u32)G1EXP_MAX_IO || (air->num_io & (air->num_io - 1))) return false;
    FlagU64Shape sh((int)air->num_io);
    s.ncols = sh.num_cols(); s.npi = 0; s.npairs = 0; s.nconstraints = sh.num_constraints();
  } else if (air->kind == SBN_AIR_G1_EXP || air->kind == SBN_AIR_G2_EXP || air->kind == SBN_AIR_FQ12_EXP || air->kind == SBN_AIR_FQ_EXP || air->kind == SBN_AIR_FQ12_EXP_U64) {
    if (air->num_io == 0 || air->num_io > (u32)G1EXP_MAX_IO || (air->num_io & (air->num_io - 1))) return false;
    ExpShape sh(exp_e(air->kind), (int)air->num_io);
    s.ncols = sh.num_cols; s.npi = sh.num_pi; s.npairs = sh.num_pairs(); s.nconstraints = sh.num_constraints();
  } else return false;
  // num_permutation_batches = ceil(pairs*num_challenges / (constraint_degree-1)), constraint_degree = 3
  s.nzs = (s.npairs * nch + 1) / 2;
  return true;
}
static inline bool is_op_air(int kind) { return kind == SBN_AIR_MODULAR || kind == SBN_AIR_FQ12_MUL; }   // OpShape tables (air.cuh)
static inline bool is_exp_air(int kind) { return kind == SBN_AIR_G1_EXP || kind == SBN_AIR_G2_EXP || kind == SBN_AIR_FQ12_EXP || kind == SBN_AIR_FQ_EXP || kind == SBN_AIR_FQ12_EXP_U64; }
static inline size_t exp_rows_per_instance(int kind) { return (kind == SBN_AIR_FQ12_EXP_U64 || kind == SBN_AIR_FLAGS_U64) ? 128 : 512; }   // (FLAGS: 512)
// u32 words of one instance in the `ios` arrays of include/sbn.h (x, offset, exp_val)
static inline size_t exp_io_words(int kind) {
  switch (kind) {
    case SBN_AIR_G1_EXP: return 40; case SBN_AIR_G2_EXP: return 72; case SBN_AIR_FQ12_EXP: return 200; case SBN_AIR_FQ_EXP: return 24;
    case SBN_AIR_FQ12_EXP_U64: return 194; default: return 0;
  }
}
static inline ExpShape exp_shape(const AirShape& a) { return ExpShape(exp_e(a.kind), (int)a.num_io); }
static inline bool config_supported(const sbn_config* c) {
  return c && c->num_challenges == SBN_NCH && c->rate_bits == 1 && c->cap_height >= 1 && c->cap_height <= 8 &&
         c->fri_arity_bits >= 1 && c->fri_arity_bits <= 4 && c->num_query_rounds >= 1 && c->num_query_rounds <= 512 &&
         c->proof_of_work_bits <= 32 && c->fri_variant <= SBN_FRI_PLAIN;
}
// sbn_config.fri_variant: 0 selects the default (the 0.1.x line: final polynomial times X)
static inline u32 fri_times_x(const sbn_config& c) { return c.fri_variant == SBN_FRI_PLAIN ? 0u : 1u; }

// tracegen.hip: Jacobian curve chains of every G1ExpStark instance on host threads (layout: bn254w.cuh g1_chains)
int tracegen_host_chains(int E, const uint32_t* ios, size_t K, u64* ja, u64* jb, int form = 0);   // form: tracegen.hip
// tracegen.hip: square-and-multiply chains of every Fq12ExpStark / Fq12ExpU64Stark instance, standard form, [K][steps+1][12][4] each
int tracegen_host_chains_fq12(const uint32_t* ios, size_t iow, int steps, size_t K, u64* ca, u64* cb);
// the same for FqExpStark: [K][257][4] each
int tracegen_host_chains_fq(const uint32_t* ios, size_t K, u64* ca, u64* cb);
// tracegen.hip: threads of the host worker pool (the caller's included): SBN_HOST_THREADS, else one per visible CPU (<= 64)
unsigned tracegen_host_threads();
// tracegen.hip: the curve chains run eight instances per AVX-512 IFMA register on this CPU (0.15 ms per group of eight)
bool tracegen_host_chains_vectorized();
// prover.hip: the device sbn_set_device / sbn_set_thread_device selected for the calling thread (else the process default)
int current_device();

}  // namespace sbn

struct sbn_proof {
  std::vector<u64> words;
  u32 degree_bits;
};
