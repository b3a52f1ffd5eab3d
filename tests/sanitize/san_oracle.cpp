// AddressSanitizer / UBSan run of the ORACLE's host code (test infrastructure; SURVEY.md section 5 asks for a sanitizer build
// of the CPU side).  Built by `make -C oracle san` from oracle/capi.cpp with -fsanitize=address,undefined; started by
// tests/test_sanitizers.py, which writes the seeded inputs and reads the traces / proofs back.
//   san_oracle <dir>: reads <dir>/modular_ops.bin (rows x 16 u32), <dir>/g1op_pts.bin (rows x 32 u32); for each table:
//   generate_trace -> prove -> verify -> a tampered proof is rejected; writes <table>_trace.bin, <table>_proof.bin.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

extern "C" {
size_t orc_air_num_columns(int kind, size_t num_io);
int orc_modular_generate_trace(const uint32_t* ops, size_t rows, uint64_t* trace_out);
int orc_g1op_generate_trace(const uint32_t* pts, size_t rows, uint64_t* trace_out);
int orc_prove(int kind, size_t num_io, const uint64_t* trace, unsigned degree_bits, const uint64_t* pi, size_t npi, uint64_t** proof_out, size_t* nwords_out,
              double* seconds_out);
int orc_verify(int kind, size_t num_io, const uint64_t* proof, size_t nwords, const char** why);
void orc_free(void* p);
}

template <class T> static std::vector<T> read_all(const std::string& path) {
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) { fprintf(stderr, "cannot open %s\n", path.c_str()); exit(2); }
  fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
  std::vector<T> v((size_t)n / sizeof(T));
  if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(2);
  fclose(f);
  return v;
}
template <class T> static void write_all(const std::string& path, const T* p, size_t n) {
  FILE* f = fopen(path.c_str(), "wb");
  if (!f || fwrite(p, sizeof(T), n, f) != n) { fprintf(stderr, "cannot write %s\n", path.c_str()); exit(2); }
  fclose(f);
}

static int run(const std::string& dir, const char* name, int kind, const std::vector<uint32_t>& in, size_t words_per_row,
               int (*gen)(const uint32_t*, size_t, uint64_t*)) {
  const size_t rows = in.size() / words_per_row, ncols = orc_air_num_columns(kind, 0);
  unsigned bits = 0; while (((size_t)1 << bits) < rows) bits++;
  std::vector<uint64_t> trace(ncols * rows);
  if (gen(in.data(), rows, trace.data())) return 1;
  uint64_t* proof = nullptr; size_t nw = 0; double secs = 0;
  if (orc_prove(kind, 0, trace.data(), bits, nullptr, 0, &proof, &nw, &secs)) return 1;
  const char* why = "";
  if (orc_verify(kind, 0, proof, nw, &why)) { fprintf(stderr, "%s: own proof rejected: %s\n", name, why); return 1; }
  std::vector<uint64_t> bad(proof, proof + nw);
  bad[12 + 3] ^= 1;   // a word of the trace cap
  if (!orc_verify(kind, 0, bad.data(), nw, &why)) { fprintf(stderr, "%s: tampered proof accepted\n", name); return 1; }
  bad.assign(proof, proof + nw / 2);   // truncated
  if (!orc_verify(kind, 0, bad.data(), bad.size(), &why)) { fprintf(stderr, "%s: truncated proof accepted\n", name); return 1; }
  write_all(dir + "/" + name + "_trace.bin", trace.data(), trace.size());
  write_all(dir + "/" + name + "_proof.bin", proof, nw);
  orc_free(proof);
  printf("%s: %zu rows x %zu columns, proof %zu words, %.2f s under the sanitizers\n", name, rows, ncols, nw, secs);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage: san_oracle <dir>\n"); return 2; }
  const std::string dir = argv[1];
  int rc = run(dir, "modular", 7, read_all<uint32_t>(dir + "/modular_ops.bin"), 16, orc_modular_generate_trace);
  rc |= run(dir, "g1op", 1, read_all<uint32_t>(dir + "/g1op_pts.bin"), 32, orc_g1op_generate_trace);
  return rc;
}
