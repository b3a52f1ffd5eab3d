// C ABI glue (include/sbn.h): configuration, table shapes, proof object, one-shot prove().
#include "host_common.hpp"
#include <cstring>
#include <atomic>
#include <mutex>
#include <thread>
#include <vector>

using namespace sbn;

extern "C" {

const char* sbn_version(void) { return "starky-bn254-amd 0.3 (gfx950)"; }
int sbn_abi_version(void) { return SBN_ABI_VERSION; }
const char* sbn_last_error(void) { return g_last_error.c_str(); }

// starky config.rs `StarkConfig::standard_fast_config` (the reference passes the column / public-input
// counts at run time, src/curves/g1/exp.rs:250-253; here they come from the table descriptor).
void sbn_standard_fast_config(sbn_config* c) {
  if (!c) return;
  c->security_bits = 100; c->num_challenges = 2; c->rate_bits = 1; c->cap_height = 4; c->proof_of_work_bits = 16;
  c->fri_arity_bits = 4; c->fri_final_poly_bits = 5; c->num_query_rounds = 84;
  c->fri_variant = SBN_FRI_TIMES_X;
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

// ---- batch mode (BASELINE config[2]: a batch of independent proofs per GPU) -------------------------------------------
// `inflight` prover contexts on the current GPU, one host thread each; every unit = one instance list of the table,
// witness generated on the device, then proved.  While one proof sits in a latency-bound tail or waits for the host
// transcript, the kernels of the others fill the GPU (30.7 instead of 26.5 proofs/s for G1ExpStark(128)).
struct sbn_batch_prover { std::vector<sbn_prover*> provers; int kind = 0; };

int sbn_batch_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, uint32_t inflight, sbn_batch_prover** out) {
  if (!out || inflight == 0 || inflight > 16) return fail(SBN_ERR_BAD_ARG, "bad arguments (1 <= inflight <= 16)");
  *out = nullptr;
  if (!air) return fail(SBN_ERR_BAD_ARG, "null argument");
  sbn_batch_prover* B = new sbn_batch_prover();
  B->kind = air->kind;
  for (uint32_t i = 0; i < inflight; i++) {
    sbn_prover* P = nullptr;
    int rc = sbn_prover_create(air, cfg, degree_bits, &P);
    if (rc) { for (auto q : B->provers) sbn_prover_destroy(q); delete B; return rc; }
    B->provers.push_back(P);
  }
  *out = B;
  return SBN_OK;
}
void sbn_batch_prover_destroy(sbn_batch_prover* B) {
  if (!B) return;
  for (auto q : B->provers) sbn_prover_destroy(q);
  delete B;
}
int sbn_batch_prover_prove_ios(sbn_batch_prover* B, const uint32_t* ios, size_t ios_words_per_unit, size_t num_io, size_t count, sbn_proof** proofs_out) {
  if (!B || !ios || !proofs_out) return fail(SBN_ERR_BAD_ARG, "null argument");
  for (size_t i = 0; i < count; i++) proofs_out[i] = nullptr;
  if (exp_io_words(B->kind) == 0 || ios_words_per_unit != exp_io_words(B->kind) * num_io)
    return fail(SBN_ERR_BAD_ARG, "ios_words_per_unit must be %zu u32 words per instance times num_io", exp_io_words(B->kind));
  std::atomic<size_t> next(0);
  std::atomic<int> first_rc(0);
  std::mutex m; std::string msg;
  auto work = [&](sbn_prover* P) {
    for (size_t u; (u = next.fetch_add(1)) < count && first_rc.load() == 0;) {
      int rc = sbn_prover_generate_trace(P, ios + u * ios_words_per_unit, num_io, nullptr);
      if (!rc) rc = sbn_prover_prove(P, &proofs_out[u]);
      if (rc) { std::lock_guard<std::mutex> g(m); if (first_rc.load() == 0) { first_rc = rc; msg = g_last_error; } }
    }
  };
  std::vector<std::thread> th;
  for (size_t i = 1; i < B->provers.size() && i < count; i++) th.emplace_back(work, B->provers[i]);
  work(B->provers[0]);
  for (auto& t : th) t.join();
  if (first_rc.load()) {
    for (size_t i = 0; i < count; i++) { delete proofs_out[i]; proofs_out[i] = nullptr; }
    return fail(first_rc.load(), "%s", msg.c_str());
  }
  return SBN_OK;
}

}  // extern "C"
