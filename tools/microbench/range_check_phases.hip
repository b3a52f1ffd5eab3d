// Micro-benchmark: where the time of tg::range_check_kernel goes (1.2 ms per G1 witness, 381 workgroups of
// 156 KB LDS = two rounds on 256 CUs).  The kernel is compiled here with SBN_RC_PROFILE: workgroup 0 stamps the 100 MHz clock at
// its phase boundaries.  Columns: uniform 16-bit limbs with `zero_frac` of the rows zero (rows without an operation hold zeros in
// every gadget column) and 14-bit columns (aux_hi); 381 columns x 65,536 rows like G1ExpStark(128).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc range_check_phases.hip -o range_check_phases
#define SBN_RC_PROFILE 1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "kernels_tracegen.cuh"

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
  const int ncol = argc > 1 ? atoi(argv[1]) : 381;
  const size_t n = 65536;
  // trace layout of the kernel: target columns first_col .., outputs at start_lookups + 1 + 2k (sorted) and + 2 + 2k (permuted table)
  const int first_col = 0, start_lookups = ncol;
  const size_t cols = (size_t)ncol + 1 + 2 * (size_t)ncol;
  std::vector<u64> h(cols * n, 0);
  std::mt19937_64 rng(5);
  for (int c = 0; c < ncol; c++) {
    const int kind = c % 3;   // 0: uniform 16-bit, 1: 30% zero rows + uniform, 2: 14-bit
    for (size_t i = 0; i < n; i++) {
      u64 v = rng() & 0xffff;
      if (kind == 1 && (rng() % 10) < 3) v = 0;
      if (kind == 2) v &= 0x3fff;
      h[(size_t)c * n + i] = v;
    }
  }
  u64* d; int* d_err;
  HIPC(hipMalloc(&d, cols * n * sizeof(u64))); HIPC(hipMalloc(&d_err, sizeof(int)));
  HIPC(hipMemcpy(d, h.data(), cols * n * sizeof(u64), hipMemcpyHostToDevice));
  HIPC(hipMemset(d_err, 0, sizeof(int)));
  HIPC(hipFuncSetAttribute((const void*)tg::range_check_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tg::RC_LDS_BYTES));
  hipFuncAttributes fa; HIPC(hipFuncGetAttributes(&fa, (const void*)tg::range_check_kernel<false>));
  printf("# range_check_kernel<false>: %d VGPRs, %d B scratch, %zu B LDS; %d columns x %zu rows\n", fa.numRegs, (int)fa.localSizeBytes, (size_t)tg::RC_LDS_BYTES, ncol, n);
  hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  static const char* names[] = {"zero LDS", "histogram (column read, LDS atomics)", "prefix counts", "min tree + in-segment pool (registers)", "left-over walk",
                                "strided pass (sorted copy, first occurrences)", "(barrier)", "heavy values + 65535 run"};
  for (int old_form = 0; old_form < 2; old_form++) {
    for (int rep = 0; rep < 2; rep++) {
      HIPC(hipEventRecord(e0));
      hipLaunchKernelGGL(tg::range_check_kernel<false>, dim3(ncol), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, 0, d, n, first_col, start_lookups, d_err, (const unsigned int*)nullptr, old_form);
      HIPC(hipEventRecord(e1)); HIPC(hipEventSynchronize(e1));
    }
    float ms; HIPC(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long prof[16];
    HIPC(hipMemcpyFromSymbol(prof, HIP_SYMBOL(tg::g_rc_prof), sizeof prof));
    printf("%s form: %.3f ms for the launch; workgroup 0 (column kind 0), microseconds per phase:\n", old_form ? "one-pass (round 3)" : "two-pass", ms);
    if (old_form) {
      for (int k = 0; k < 4; k++) printf("   %-44s %8.2f\n", names[k], (prof[k + 1] - prof[k]) / 100.0);
      printf("   %-44s %8.2f\n", "main loop (every absent value searches)", (prof[7] - prof[4]) / 100.0);
      printf("   %-44s %8.2f\n", names[7], (prof[8] - prof[7]) / 100.0);
    } else
      for (int k = 0; k < 8; k++) printf("   %-44s %8.2f\n", names[k], (prof[k + 1] - prof[k]) / 100.0);
    printf("   %-44s %8.2f\n", "total (workgroup 0)", (prof[8] - prof[0]) / 100.0);
  }
  int err = 0; HIPC(hipMemcpy(&err, d_err, sizeof(int), hipMemcpyDeviceToHost));
  printf("# err word %d\n", err);
  return 0;
}
