// Micro-benchmark: does a v_mfma_i32_4x4x4i8 (16 blocks: a lane-local 4x4 int8 matrix-vector product with a matrix shared by
// the wave) cost VALU issue time on gfx950?  A wave runs ITER iterations of a block of NV independent v_mad_u64_u32 on 8
// chains with NM matrix instructions (8 independent accumulator quads) spread between them, at W = 1, 2, 3 waves per SIMD.
// If the matrix pipe runs beside the VALU pipe, the time of (NV, NM) stays that of (NV, 0) until NM * 8 cycles > NV * 4.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 mfma_mix.hip -o mfma_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint64_t u64; typedef uint32_t u32;
typedef int v4i __attribute__((ext_vector_type(4)));
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
#define ITER 2048

template <int NV, int NM>
__global__ __launch_bounds__(256) void mix_kernel(u64* out, u32 seed) {
  extern __shared__ u32 lds_[];
  if (seed == 0xdeadbeef) lds_[threadIdx.x] = seed;
  u64 y[8];
  for (int i = 0; i < 8; i++) y[i] = (u64)threadIdx.x * 977 + i + seed;
  v4i acc[8];
  for (int i = 0; i < 8; i++) acc[i] = v4i{(int)threadIdx.x, i, 2, 3};
  int a = (int)(threadIdx.x * 0x01010101u + seed), b = (int)(threadIdx.x * 0x00030507u ^ seed);
  const u32 m = threadIdx.x | 1;
  for (int it = 0; it < ITER; it++) {
#pragma unroll
    for (int k = 0; k < (NV > NM ? NV : NM); k++) {
      if (k < NV) y[k & 7] = (u64)((u32)y[k & 7]) * m + y[(k + 1) & 7];      // v_mad_u64_u32, chains of length NV/8
      if (k < NM) acc[k & 7] = __builtin_amdgcn_mfma_i32_4x4x4i8(a, b, acc[k & 7], 0, 0, 0);
    }
  }
  u64 x = 0;
  for (int i = 0; i < 8; i++) x ^= y[i] ^ (u64)(u32)(acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3]);
  out[blockIdx.x * 256 + threadIdx.x] = x;
}

template <int NV, int NM> int run(int cus, u64* d) {
  hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  printf("NV=%2d NM=%2d:", NV, NM);
  for (int w = 1; w <= 3; w++) {
    const int blocks = cus * w;
    const size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)255;
    HIPC(hipFuncSetAttribute((const void*)mix_kernel<NV, NM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((mix_kernel<NV, NM>), dim3(blocks), dim3(256), lds, 0, d, 1u);
    HIPC(hipDeviceSynchronize());
    HIPC(hipEventRecord(e0));
    hipLaunchKernelGGL((mix_kernel<NV, NM>), dim3(blocks), dim3(256), lds, 0, d, 1u);
    HIPC(hipEventRecord(e1)); HIPC(hipEventSynchronize(e1));
    float ms; HIPC(hipEventElapsedTime(&ms, e0, e1));
    printf("  W=%d %8.2f ns/iter/wave-slot", w, ms * 1e6 / ITER / w);
  }
  printf("\n");
  return 0;
}

int main() {
  hipDeviceProp_t prop; HIPC(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  u64* d; HIPC(hipMalloc(&d, (size_t)cus * 4 * 256 * 8));
  printf("# %s, %d CUs; time per loop iteration of one wave divided by waves per SIMD (= SIMD time per wave-iteration)\n", prop.gcnArchName, cus);
  run<32, 0>(cus, d); run<32, 4>(cus, d); run<32, 8>(cus, d); run<32, 16>(cus, d); run<32, 32>(cus, d);
  run<0, 32>(cus, d); run<0, 8>(cus, d); run<16, 16>(cus, d); run<8, 0>(cus, d); run<8, 8>(cus, d);
  return 0;
}
