// C ABI glue (include/sbn.h): configuration, table shapes, proof object, one-shot prove().
#include "host_common.hpp"
#include <cstring>

using namespace sbn;

extern "C" {

const char* sbn_version(void) { return "starky-bn254-amd 0.1 (gfx950)"; }
const char* sbn_last_error(void) { return g_last_error.c_str(); }

// starky config.rs `StarkConfig::standard_fast_config` (the reference passes the column / public-input
// counts at run time, src/curves/g1/exp.rs:250-253; here they come from the table descriptor).
void sbn_standard_fast_config(sbn_config* c) {
  if (!c) return;
  c->security_bits = 100; c->num_challenges = 2; c->rate_bits = 1; c->cap_height = 4; c->proof_of_work_bits = 16;
  c->fri_arity_bits = 4; c->fri_final_poly_bits = 5; c->num_query_rounds = 84;
}

size_t sbn_air_num_columns(const sbn_air_desc* air) { AirShape s; return air_shape(air, nullptr, s) ? s.ncols : 0; }
size_t sbn_air_num_public_inputs(const sbn_air_desc* air) { AirShape s; return air_shape(air, nullptr, s) ? s.npi : 0; }
size_t sbn_air_num_permutation_zs(const sbn_air_desc* air, const sbn_config* cfg) { AirShape s; return air_shape(air, cfg, s) ? s.nzs : 0; }
size_t sbn_air_num_constraints(const sbn_air_desc* air) { AirShape s; return air_shape(air, nullptr, s) ? s.nconstraints : 0; }

size_t sbn_proof_num_words(const sbn_proof* p) { return p ? p->words.size() : 0; }
const uint64_t* sbn_proof_words(const sbn_proof* p) { return p ? p->words.data() : nullptr; }
size_t sbn_proof_serialize(const sbn_proof* p, uint8_t* buf, size_t cap) {
  if (!p) return 0;
  size_t need = p->words.size() * 8;
  if (buf && cap >= need) memcpy(buf, p->words.data(), need);  // little-endian host
  return need;
}
uint32_t sbn_proof_degree_bits(const sbn_proof* p) { return p ? p->degree_bits : 0; }
void sbn_proof_free(sbn_proof* p) { delete p; }

int sbn_prove(const sbn_air_desc* air, const sbn_config* cfg, const uint64_t* trace, uint32_t degree_bits, const uint64_t* pi, size_t n_pi,
              sbn_proof** out) {
  if (!out) return fail(SBN_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  sbn_prover* P = nullptr;
  int rc = sbn_prover_create(air, cfg, degree_bits, &P);
  if (rc) return rc;
  rc = sbn_prover_load_trace(P, trace, pi, n_pi);
  if (!rc) rc = sbn_prover_prove(P, out);
  sbn_prover_destroy(P);
  return rc;
}

}  // extern "C"
