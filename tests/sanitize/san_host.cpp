// AddressSanitizer / UBSan run of the PRODUCT's host-side code (host witness generators of tracegen.hip, the verifier and
// the host constraint evaluator of verifier.hip), built host-instrumented by `make -C starky_bn254_amd/csrc san`
// (hipcc -fsanitize=address,undefined -fno-gpu-sanitize; no GPU is touched: nothing here launches a kernel).  The C ABI
// takes raw pointers and sizes, so this is where an off-by-one in a column map or a proof reader would show.
//   san_host <dir>: inputs and the oracle's traces / proofs as written by tests/test_sanitizers.py and san_oracle.
#include "../../include/sbn.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace sbn { thread_local std::string g_last_error; }   // defined in prover.hip in the library proper

template <class T> static std::vector<T> read_all(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<T> v((size_t)n / sizeof(T));
  if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}

static int run(const std::string& dir, const char* name, int kind, size_t words_per_row, int (*gen)(const uint32_t*, size_t, uint64_t*)) {
  const std::vector<uint32_t> in = read_all<uint32_t>(dir + "/" + (kind == SBN_AIR_MODULAR ? "modular_ops.bin" : "g1op_pts.bin"));
  const std::vector<uint64_t> want = read_all<uint64_t>(dir + "/" + name + "_trace.bin");
  std::vector<uint64_t> proof = read_all<uint64_t>(dir + "/" + name + "_proof.bin");
  sbn_air_desc air{kind, 0};
  sbn_config cfg; memset(&cfg, 0, sizeof cfg);
  // (sbn_standard_fast_config lives in capi.hip; the values are StarkConfig::standard_fast_config's)
  cfg.security_bits = 100; cfg.num_challenges = 2; cfg.rate_bits = 1; cfg.cap_height = 4; cfg.proof_of_work_bits = 16;
  cfg.fri_arity_bits = 4; cfg.fri_final_poly_bits = 5; cfg.num_query_rounds = 84; cfg.fri_variant = SBN_FRI_DEFAULT;
  const size_t rows = in.size() / words_per_row, ncols = want.size() / rows;
  std::vector<uint64_t> trace(ncols * rows);
  if (gen(in.data(), rows, trace.data())) { fprintf(stderr, "%s: generate_trace failed\n", name); return 1; }
  if (trace != want) { fprintf(stderr, "%s: host witness differs from the oracle's\n", name); return 1; }
  if (sbn_verify(&air, &cfg, (const uint8_t*)proof.data(), proof.size() * 8)) { fprintf(stderr, "%s: oracle proof rejected\n", name); return 1; }
  // every truncation point of the header and the first sections, and a flipped word in each region, must be REJECTED
  // cleanly (no read past the buffer)
  for (size_t cut : {(size_t)0, (size_t)1, (size_t)11, (size_t)12, (size_t)75, proof.size() / 3, proof.size() - 1})
    if (!sbn_verify(&air, &cfg, (const uint8_t*)proof.data(), cut * 8)) { fprintf(stderr, "%s: truncated proof (%zu words) accepted\n", name, cut); return 1; }
  if (!sbn_verify(&air, &cfg, (const uint8_t*)proof.data(), proof.size() * 8 - 3)) { fprintf(stderr, "%s: ragged length accepted\n", name); return 1; }
  for (size_t at : {(size_t)1, (size_t)2, (size_t)8, (size_t)10, (size_t)11, (size_t)20, proof.size() / 2, proof.size() - 2}) {
    std::vector<uint64_t> bad = proof;
    bad[at] ^= 5;
    if (!sbn_verify(&air, &cfg, (const uint8_t*)bad.data(), bad.size() * 8)) { fprintf(stderr, "%s: tampered word %zu accepted\n", name, at); return 1; }
  }
  // the regrouped constraint evaluator on consecutive rows of the trace: vanishes on the trace domain
  std::vector<uint64_t> lv(ncols), nv(ncols);
  uint64_t alphas[2] = {0x123456789abcdefULL, 0xfedcba987654321ULL}, acc[2];
  for (size_t r : {(size_t)0, (size_t)1, rows / 2}) {
    for (size_t c = 0; c < ncols; c++) { lv[c] = trace[c * rows + r]; nv[c] = trace[c * rows + (r + 1) % rows]; }
    if (sbn_eval_constraints_host(&air, lv.data(), nv.data(), nullptr, 0, alphas, 1, r == 0, 0, acc)) { fprintf(stderr, "%s: eval failed\n", name); return 1; }
    if (acc[0] || acc[1]) { fprintf(stderr, "%s: constraints do not vanish on row %zu\n", name, r); return 1; }
  }
  printf("%s: host witness == oracle, verifier accepts / rejects as it must, constraints vanish\n", name);
  return 0;
}

// the two unit-test tables' generators (lookup.rs:151-166 on its fixed input, flags.rs:392-440 on one input) and their evaluators
static int unit_tables() {
  const uint64_t inputs[8] = {6, 3, 1, 1, 0, 0, 0, 0}, table[8] = {0, 1, 2, 3, 4, 5, 6, 7}, want2[8] = {0, 0, 0, 0, 1, 1, 3, 6}, want3[8] = {0, 2, 4, 5, 1, 7, 3, 6};
  uint64_t t[32];
  if (sbn_generate_trace_lookup(inputs, table, 8, t) || memcmp(t + 16, want2, sizeof want2) || memcmp(t + 24, want3, sizeof want3)) { fprintf(stderr, "lookup witness\n"); return 1; }
  const uint32_t limbs[8] = {0x89abcdefu, 1, 2, 3, 4, 5, 6, 0xffffffffu};
  std::vector<uint64_t> f((size_t)21 * 512);
  if (sbn_generate_trace_flags(limbs, 1, f.data())) { fprintf(stderr, "flags witness\n"); return 1; }
  sbn_air_desc lk{SBN_AIR_LOOKUP, 0}, fl{SBN_AIR_FLAGS, 1};
  uint64_t alphas[2] = {3, 5}, acc[2], lv[21], nv[21];
  for (size_t r : {(size_t)0, (size_t)62, (size_t)510}) {
    for (int c = 0; c < 21; c++) { lv[c] = f[(size_t)c * 512 + r]; nv[c] = f[(size_t)c * 512 + r + 1]; }
    if (sbn_eval_constraints_host(&fl, lv, nv, nullptr, 0, alphas, 1, r == 0, 0, acc) || acc[0] || acc[1]) { fprintf(stderr, "flags constraints, row %zu\n", r); return 1; }
  }
  for (size_t r = 0; r < 7; r++) {
    for (int c = 0; c < 4; c++) { lv[c] = t[c * 8 + r]; nv[c] = t[c * 8 + r + 1]; }
    if (sbn_eval_constraints_host(&lk, lv, nv, nullptr, 0, alphas, 1, 0, 0, acc) || acc[0] || acc[1]) { fprintf(stderr, "lookup constraints, row %zu\n", r); return 1; }
  }
  printf("unit-test tables: witnesses and constraints ok\n");
  return 0;
}

// the curve chains of the device witness (csrc/bn254w.cuh exp_chains) one instance at a time and eight per AVX-512 IFMA register
// (csrc/chains_ifma.hpp): the same words, for whole and ragged groups -- with every store and shift of the vector form under ASan / UBSan
static int curve_chains() {
  // G1 points with small coordinates: (1, 2) is on y^2 = x^3 + 3; offsets and more points by repeated doubling are not needed -- the
  // chains accept any pair of affine points that is not degenerate for the given exponent
  const uint64_t P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
  for (int count : {8, 11}) {
    std::vector<uint32_t> ios((size_t)count * 40, 0);
    for (int k = 0; k < count; k++) {
      uint32_t* io = ios.data() + 40 * k;
      io[0] = 1; io[8] = 2;                                  // x = (1, 2)
      io[16] = 1;                                            // offset = (1, -2)
      uint64_t ny[4]; unsigned __int128 br = 0;
      for (int i = 0; i < 4; i++) { unsigned __int128 t = (unsigned __int128)P[i] - (i == 0 ? 2 : 0) - (uint64_t)br; ny[i] = (uint64_t)t; br = (t >> 64) & 1; }
      for (int i = 0; i < 4; i++) { io[24 + 2 * i] = (uint32_t)ny[i]; io[24 + 2 * i + 1] = (uint32_t)(ny[i] >> 32); }
      for (int i = 0; i < 8; i++) io[32 + i] = 0x9e3779b9u * (uint32_t)(k * 8 + i + 1) | 2u;   // exponent: bit 0 clear (x + (-x) would be degenerate), bit 1 set
      io[32] &= ~1u;
    }
    const size_t words = (size_t)count * 257 * 3 * 4;
    std::vector<uint64_t> ja1(words), jb1(words), ja2(words, 7), jb2(words, 9);
    if (sbn_host_curve_chains(1, ios.data(), (size_t)count, ja1.data(), jb1.data(), 1)) { fprintf(stderr, "scalar chains failed: %s\n", sbn::g_last_error.c_str()); return 1; }
    const int rc = sbn_host_curve_chains(1, ios.data(), (size_t)count, ja2.data(), jb2.data(), 2);
    if (rc == SBN_ERR_UNSUPPORTED) { printf("curve chains: this CPU has no AVX-512 IFMA, scalar form only\n"); return 0; }
    if (rc) { fprintf(stderr, "IFMA chains failed: %s\n", sbn::g_last_error.c_str()); return 1; }
    if (ja1 != ja2 || jb1 != jb2) { fprintf(stderr, "IFMA chains differ from the scalar chains (%d instances)\n", count); return 1; }
  }
  printf("curve chains: eight instances per IFMA register == one at a time\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: san_host <dir>\n"); return 2; }
  if (unit_tables()) return 1;
  if (curve_chains()) return 1;
  int rc = run(argv[1], "modular", SBN_AIR_MODULAR, 16, sbn_generate_trace_modular);
  rc |= run(argv[1], "g1op", SBN_AIR_G1_OP, 32, sbn_generate_trace_g1_op);
  return rc;
}
