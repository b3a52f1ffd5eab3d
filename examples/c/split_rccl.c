/* One proof split over the GPUs of a node from plain C: one process per GPU, the library's RCCL transport, the 128-byte
 * unique id handed from rank 0 to the others through a file (any channel does: a socket, MPI, ...).
 * BASELINE config[4] is `split_rccl <rank> <world> <id file> fq12_ios.bin 512` on 8 GPUs (reference workload
 * src/fields/fq12/exp.rs:638-696); with world = 1 it runs on a one-GPU box.
 *
 *   gcc -O2 -I../../include split_rccl.c -L../../starky_bn254_amd -lsbn254 -Wl,-rpath,$PWD/../../starky_bn254_amd -o split_rccl
 *   for r in 0 1 ... ; do ./split_rccl $r $WORLD /tmp/sbn_id ios.bin 16 & done        (rank r uses GPU r)
 *
 * ios.bin = num_io x 200 little-endian u32: x[12] offset[12] (flat-basis Fq12 coefficients, 8 limbs each) exp_val[8]. */
#include "sbn.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static int die(const char* what, int rc) { fprintf(stderr, "%s failed (%d): %s\n", what, rc, sbn_last_error()); return 1; }

int main(int argc, char** argv) {
  if (argc < 6) { fprintf(stderr, "usage: %s <rank> <world> <id file> <ios.bin> <num_io>\n", argv[0]); return 2; }
  const uint32_t rank = (uint32_t)atoi(argv[1]), world = (uint32_t)atoi(argv[2]), num_io = (uint32_t)atoi(argv[5]);
  uint32_t bits = 9; while ((1u << bits) < 512 * num_io) bits++;
  uint32_t* ios = malloc((size_t)num_io * 200 * sizeof *ios);
  FILE* f = fopen(argv[4], "rb");
  if (!f || fread(ios, (size_t)num_io * 200 * sizeof *ios, 1, f) != 1) { fprintf(stderr, "cannot read %s\n", argv[4]); return 2; }
  fclose(f);
  int rc = sbn_set_device((int)(rank % (uint32_t)sbn_device_count()));
  if (rc) return die("sbn_set_device", rc);
  /* the RCCL unique id: rank 0 draws it and publishes it (write to a temporary name, then rename: readers never see half of it) */
  uint8_t id[128];
  if (rank == 0) {
    if ((rc = sbn_rccl_unique_id(id))) return die("sbn_rccl_unique_id", rc);
    char tmp[512]; snprintf(tmp, sizeof tmp, "%s.tmp", argv[3]);
    f = fopen(tmp, "wb");
    if (!f || fwrite(id, 128, 1, f) != 1) { fprintf(stderr, "cannot write %s\n", tmp); return 2; }
    fclose(f); rename(tmp, argv[3]);
  } else {
    for (int tries = 0; !(f = fopen(argv[3], "rb")); tries++) { if (tries > 600) { fprintf(stderr, "no id file\n"); return 2; } usleep(100000); }
    if (fread(id, 128, 1, f) != 1) return 2;
    fclose(f);
  }
  sbn_air_desc air = {SBN_AIR_FQ12_EXP, num_io};
  sbn_config cfg; sbn_standard_fast_config(&cfg);
  uint64_t sb = 0, rb = 0;
  if ((rc = sbn_split_exchange_bytes(&air, &cfg, bits, world, &sb, &rb))) return die("sbn_split_exchange_bytes", rc);
  sbn_comm comm;
  if ((rc = sbn_rccl_comm_create(id, rank, world, sb, rb, &comm))) return die("sbn_rccl_comm_create", rc);
  if ((rc = sbn_comm_selftest(&comm))) return die("sbn_comm_selftest", rc);            /* optional: the transport moves bytes where they belong */
  sbn_split_prover* prover = NULL; sbn_proof* proof = NULL;
  uint64_t* pi = malloc(sbn_air_num_public_inputs(&air) * sizeof *pi);
  if ((rc = sbn_split_prover_create(&air, &cfg, bits, &comm, &prover))) return die("sbn_split_prover_create", rc);
  if ((rc = sbn_split_prover_generate_trace(prover, ios, num_io, pi))) return die("sbn_split_prover_generate_trace", rc);
  if ((rc = sbn_split_prover_prove(prover, &proof))) return die("sbn_split_prover_prove", rc);   /* the same words on every rank */
  size_t bytes = sbn_proof_serialize(proof, NULL, 0);
  uint8_t* buf = malloc(bytes);
  sbn_proof_serialize(proof, buf, bytes);
  if ((rc = sbn_verify(&air, &cfg, buf, bytes))) return die("sbn_verify", rc);
  uint64_t sum = 0; const uint64_t* w = sbn_proof_words(proof);
  for (size_t i = 0; i < sbn_proof_num_words(proof); i++) sum = sum * 0x100000001b3ULL ^ w[i];
  printf("rank %u of %u: Fq12ExpStark(%u), 2^%u rows, proof %zu words, checksum %016llx, verified\n", rank, world, num_io, bits, sbn_proof_num_words(proof),
         (unsigned long long)sum);
  sbn_proof_free(proof); sbn_split_prover_destroy(prover); sbn_rccl_comm_destroy(&comm);
  free(buf); free(pi); free(ios);
  return 0;
}
