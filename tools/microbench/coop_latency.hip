// Latency of ONE Poseidon permutation per wave in three shapes (dependent chain of `iters` permutations, one wave):
//   thread : one lane per permutation (poseidon_permute_fast)
//   dpp16  : 16 lanes per permutation, DPP row rotations (poseidon_permute_coop16)
//   bperm16: 16 lanes per permutation, ds_bpermute (__shfl width 16)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../starky_bn254_amd/csrc coop_latency.hip -o coop_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#include "poseidon.cuh"

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u64 coop16_bperm(u64 v, u32 lane) {
  const u64* rc = POSEIDON_EFF_DEV;
  constexpr u32 C[12] = {17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20};
  const u32 j = lane < 12 ? lane : 0;
  for (int r = 0; r < 30; r++) {
    v = pw::add_c(v, rc[12 * r + j]);
    const bool full = r < 4 || r >= 26;
    u64 sb = pw::sbox(v);
    v = (full || lane == 0) ? sb : v;
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    u64 al = 0, ah = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) {
      u32 src = j + i; src = src >= 12 ? src - 12 : src;
      u32 xl = __shfl(lo, (int)src, 16), xh = __shfl(hi, (int)src, 16);
      al += (u64)xl * C[i]; ah += (u64)xh * C[i];
    }
    if (lane == 0) { al += (u64)lo * 8u; ah += (u64)hi * 8u; }
    u32 h0 = (u32)ah, h1 = (u32)(ah >> 32);
    u64 t = al + ((u64)h0 << 32);
    u32 c = t < al;
    u64 add = (u64)(h1 + c) * GLEPS;
    u64 o = t + add;
    if (o < add) o += GLEPS;
    v = o;
  }
  return v >= GLP ? v - GLP : v;
}

#define DEV_ONLY(...) __VA_ARGS__
#else
#define DEV_ONLY(...)
#endif

__global__ void k_thread(u64* out, int iters) {
  DEV_ONLY(F s[12];
  for (int i = 0; i < 12; i++) s[i] = F(threadIdx.x * 12 + i);
  for (int it = 0; it < iters; it++) poseidon_permute_fast(s);
  out[threadIdx.x + blockIdx.x * blockDim.x] = s[0].v;)
}
__global__ void k_dpp(u64* out, int iters) {
  DEV_ONLY(const u32 lane = threadIdx.x & 15; const u32 e = lane < 12 ? lane : lane - 12;
  u64 v = (threadIdx.x / 16) * 12 + e;
  for (int it = 0; it < iters; it++) v = poseidon_permute_coop16(v, lane);
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;)
}
__global__ void k_bperm(u64* out, int iters) {
  DEV_ONLY(const u32 lane = threadIdx.x & 15;
  u64 v = (threadIdx.x / 16) * 12 + (lane < 12 ? lane : 0);
  for (int it = 0; it < iters; it++) v = coop16_bperm(v, lane);
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;)
}

__global__ void __launch_bounds__(1024) k_dpp1024(u64* out, int iters) {
  DEV_ONLY(const u32 lane = threadIdx.x & 15; const u32 e = lane < 12 ? lane : lane - 12;
  u64 v = (threadIdx.x / 16) * 12 + e;
  for (int it = 0; it < iters; it++) v = poseidon_permute_coop16(v, lane);
  out[threadIdx.x + blockIdx.x * blockDim.x] = v;)
}
__global__ void __launch_bounds__(1024) k_thread1024(u64* out, int iters) {
  DEV_ONLY(F s[12];
  for (int i = 0; i < 12; i++) s[i] = F(threadIdx.x * 12 + i);
  if (threadIdx.x < 256) for (int it = 0; it < iters; it++) poseidon_permute_fast(s);
  out[threadIdx.x + blockIdx.x * blockDim.x] = s[0].v;)
}
template <typename K> static float run(K k, int blocks, int threads, int iters, u64* d) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, 2);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, d, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / iters;
}
int main() {
  u64* d; hipMalloc(&d, 1 << 24);
  const int iters = 50;
  for (int waves : {1, 2, 4, 8}) {
    int threads = 64 * (waves > 4 ? 4 : waves), blocks = waves > 4 ? waves / 4 : 1;
    printf("waves on one CU = %d (block %d x %d): thread %.1f us/perm  dpp16 %.1f us/perm  bperm16 %.1f us/perm\n", waves, blocks, threads,
           run(k_thread, blocks, threads, iters, d), run(k_dpp, blocks, threads, iters, d), run(k_bperm, blocks, threads, iters, d));
  }
  printf("block 1024 (launch_bounds 1024): dpp16 %.1f us/perm, thread (256 active lanes) %.1f us/perm; 256 such blocks: dpp16 %.1f, thread %.1f\n",
         run(k_dpp1024, 1, 1024, iters, d), run(k_thread1024, 1, 1024, iters, d), run(k_dpp1024, 256, 1024, iters, d), run(k_thread1024, 256, 1024, iters, d));
  u64 h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("check %llx\n", (unsigned long long)h[0]);
  return 0;
}
