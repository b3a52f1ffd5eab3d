// Micro-benchmark: the Poseidon permutation of the leaf sponge (poseidon_permute_fast, the body of leaf_absorb_kernel)
// alone on the chip at a CONTROLLED 1, 2, 3 (and 4 where registers allow) waves per SIMD.  Question:
// how many cycles does one wave-instruction of this stream cost its SIMD, and does a partner wave halve it?
//
// One workgroup = 256 lanes = one wave per SIMD; grid = #CUs x W workgroups, each asking for 160 KB / W of LDS so that
// exactly W are resident per CU.  Every lane runs PERMS dependent permutations (the sponge's own dependency pattern).
// Output: cycles per permutation per wave (s_memtime), permutations/s of the whole chip, and -- with the instruction
// count of one permutation taken from the kernel's ISA (argv[1], default 14700; tools/count_perm_instrs.sh prints it) --
// cycles per wave-instruction per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc sponge_rate.hip -o sponge_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "poseidon.cuh"

#define PERMS 16
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void sponge_kernel(u64* out, u64* cyc, u32 s) {
#if defined(__HIP_DEVICE_COMPILE__)
  extern __shared__ u32 lds_[];
  if (s == 0xdeadbeef) lds_[threadIdx.x] = s;
  F st[12];
  for (int i = 0; i < 12; i++) st[i] = F((u64)(threadIdx.x + blockIdx.x * 256) * 12 + i + s);
  u64 t0 = __builtin_readcyclecounter();
  for (int it = 0; it < PERMS; it++) poseidon_permute_fast(st);
  u64 t1 = __builtin_readcyclecounter();
  u64 x = 0;
  for (int i = 0; i < 12; i++) x ^= st[i].v;
  out[blockIdx.x * 256 + threadIdx.x] = x;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;
#endif
}

int main(int argc, char** argv) {
  const double instr_per_perm = argc > 1 ? atof(argv[1]) : 14700.0;
  hipDeviceProp_t prop; HIPC(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u64 *d, *dc;
  HIPC(hipMalloc(&d, (size_t)cus * 8 * 256 * 8));
  HIPC(hipMalloc(&dc, (size_t)cus * 8 * 4 * 8));
  hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  hipFuncAttributes fa; HIPC(hipFuncGetAttributes(&fa, (const void*)sponge_kernel));
  printf("# device %s, %d CUs; sponge_kernel: %d VGPRs (arch+acc), %d B scratch; %.0f wave-instructions per permutation assumed\n", prop.gcnArchName, cus,
         fa.numRegs, (int)fa.localSizeBytes, instr_per_perm);
  printf("%-4s %16s %16s %22s %20s\n", "W", "cycles/perm/wave", "us/perm/wave", "cycles/instr/SIMD", "Gperm/s (chip)");
  std::vector<u64> hc((size_t)cus * 8 * 4);
  for (int w = 1; w <= 4; w++) {
    const int blocks = cus * w;
    const size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)255;
    HIPC(hipFuncSetAttribute((const void*)sponge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = 0; HIPC(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, sponge_kernel, 256, lds));
    if (occ < w) { printf("%-4d not resident (registers allow %d workgroups per CU)\n", w, occ); continue; }
    hipLaunchKernelGGL(sponge_kernel, dim3(blocks), dim3(256), lds, 0, d, dc, 1u);
    HIPC(hipDeviceSynchronize());
    HIPC(hipEventRecord(e0));
    hipLaunchKernelGGL(sponge_kernel, dim3(blocks), dim3(256), lds, 0, d, dc, 1u);
    HIPC(hipEventRecord(e1)); HIPC(hipEventSynchronize(e1));
    float ms; HIPC(hipEventElapsedTime(&ms, e0, e1));
    HIPC(hipMemcpy(hc.data(), dc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
    double sum = 0; for (int i = 0; i < blocks * 4; i++) sum += (double)hc[i];
    const double cyc_perm = sum / (blocks * 4) / PERMS;
    printf("%-4d %16.0f %16.2f %22.2f %20.3f\n", w, cyc_perm, ms * 1e3 / PERMS, cyc_perm / (instr_per_perm * w),
           (double)blocks * 256 * PERMS / (ms * 1e-3) / 1e9);
  }
  return 0;
}
