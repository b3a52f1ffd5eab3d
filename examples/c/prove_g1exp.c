/* The C ABI from plain C (no Python, no torch): G1ExpStark(128) -- the reference's test_g1_exp_raw workload
 * (src/curves/g1/exp.rs:784-845) -- from an instance list to a verified proof on one MI355X.
 *
 *   gcc -O2 -I../../include prove_g1exp.c -L../../starky_bn254_amd -lsbn254 -Wl,-rpath,$PWD/../../starky_bn254_amd -o prove_g1exp
 *   ./prove_g1exp ios.bin        ios.bin = 128 x 40 little-endian u32: x.x x.y offset.x offset.y exp_val (include/sbn.h)
 *
 * Prints the proof's size and a checksum of its words; exits non-zero on any failure. */
#include "sbn.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: %s ios.bin\n", argv[0]); return 2; }
  static uint32_t ios[128 * 40];
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(ios, sizeof ios, 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  fclose(f);
  if (sbn_abi_version() != SBN_ABI_VERSION) { fprintf(stderr, "header / library ABI mismatch\n"); return 1; }
  sbn_air_desc air = {SBN_AIR_G1_EXP, 128};
  sbn_config cfg;
  sbn_standard_fast_config(&cfg);                                  /* stark.config()                           exp.rs:811 */
  sbn_prover* prover = NULL;
  sbn_proof* proof = NULL;
  size_t n_pi = sbn_air_num_public_inputs(&air);
  uint64_t* pi = malloc(n_pi * sizeof *pi);
  int rc = sbn_prover_create(&air, &cfg, 16, &prover);
  if (!rc) rc = sbn_prover_generate_trace(prover, ios, 128, pi);   /* generate_trace + generate_public_inputs  exp.rs:816-817, on the device */
  if (!rc) rc = sbn_prover_prove(prover, &proof);                  /* prove::<F, C, _, D>(...)                 exp.rs:818-825 */
  if (rc) { fprintf(stderr, "failed (%d): %s\n", rc, sbn_last_error()); return 1; }
  size_t bytes = sbn_proof_serialize(proof, NULL, 0);
  uint8_t* buf = malloc(bytes);
  sbn_proof_serialize(proof, buf, bytes);
  rc = sbn_verify(&air, &cfg, buf, bytes);                         /* verify_stark_proof(...)                  exp.rs:826 */
  if (rc) { fprintf(stderr, "verify failed (%d): %s\n", rc, sbn_last_error()); return 1; }
  uint64_t sum = 0;
  const uint64_t* w = sbn_proof_words(proof);
  for (size_t i = 0; i < sbn_proof_num_words(proof); i++) sum = sum * 0x100000001b3ULL ^ w[i];
  printf("G1ExpStark(128): degree_bits %u, %zu public inputs, proof %zu words, checksum %016llx, verified\n", sbn_proof_degree_bits(proof), n_pi,
         sbn_proof_num_words(proof), (unsigned long long)sum);
  float ms[32];
  int k = sbn_prover_stage_times(prover, ms, 32);
  for (int i = 0; i < k && i < 10; i++) printf("  %-18s %7.3f ms\n", sbn_prover_stage_name(i), ms[i]);
  sbn_proof_free(proof);
  sbn_prover_destroy(prover);
  free(buf); free(pi);
  return 0;
}
