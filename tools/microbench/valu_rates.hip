// Micro-benchmark: issue cost of the integer VALU instructions the Goldilocks/Poseidon kernels use.
// Each kernel runs ITER iterations of 8 independent chains of one instruction; waves = 4 per SIMD.
// Prints ns per wave-instruction per SIMD relative to v_add_u32.   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define K(name, body)                                                             \
  __global__ __launch_bounds__(256) void name(uint64_t* out, uint32_t s) {        \
    uint64_t a0 = threadIdx.x + s, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13, a6 = a0 + 17, a7 = a0 + 19; \
    uint32_t b = s | 1, c = s + 3;                                                \
    for (int i = 0; i < ITER; i++) { body }                                       \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7; \
  }
#define R8(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define MAD64(x) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(x) : "v"(b), "v"(c) : "vcc");
#define LSHLADD64(x) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(x) : "v"(a7));
#define ADD32(x) { uint32_t t = (uint32_t)x; asm volatile("v_add_u32 %0, %0, %1" : "+v"(t) : "v"(b)); x = t; }
#define MAD24(x) { uint32_t t = (uint32_t)x; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(t) : "v"(b), "v"(c)); x = t; }
#define MULLO(x) { uint32_t t = (uint32_t)x; asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(t) : "v"(b)); x = t; }
#define MULHI(x) { uint32_t t = (uint32_t)x; asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(t) : "v"(b)); x = t; }
#define LSHLADD32(x) { uint32_t t = (uint32_t)x; asm volatile("v_lshl_add_u32 %0, %0, 5, %1" : "+v"(t) : "v"(b)); x = t; }
#define ADD3(x) { uint32_t t = (uint32_t)x; asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(t) : "v"(b), "v"(c)); x = t; }
#define CNDMASK(x) { uint32_t t = (uint32_t)x; asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(t) : "v"(b) : ); x = t; }
#define ADDCO(x) { uint32_t t = (uint32_t)x; asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(t) : "v"(b) : "vcc"); x = t; }
K(k_add32, R8(ADD32)) K(k_mad64, R8(MAD64)) K(k_lshladd64, R8(LSHLADD64)) K(k_mad24, R8(MAD24)) K(k_mullo, R8(MULLO)) K(k_mulhi, R8(MULHI))
K(k_lshladd32, R8(LSHLADD32)) K(k_add3, R8(ADD3)) K(k_cndmask, R8(CNDMASK)) K(k_addco, R8(ADDCO))
int main() {
  uint64_t* d; hipMalloc(&d, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int blocks = 256 * 4;  // 4 x 256-thread blocks per CU = 4 waves per SIMD
  struct { const char* n; void (*f)(uint64_t*, uint32_t); } ks[] = {{"v_add_u32", k_add32}, {"v_mad_u64_u32", k_mad64}, {"v_lshl_add_u64", k_lshladd64},
    {"v_mad_u32_u24", k_mad24}, {"v_mul_lo_u32", k_mullo}, {"v_mul_hi_u32", k_mulhi}, {"v_lshl_add_u32", k_lshladd32}, {"v_add3_u32", k_add3},
    {"v_cndmask_b32", k_cndmask}, {"v_add_co_u32", k_addco}};
  double base = 0;
  for (auto& k : ks) {
    hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, d, 1u);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // wave-instructions per SIMD: 4 waves x ITER x 8
    double ns_per = ms * 1e6 / (4.0 * ITER * 8);
    if (!base) base = ns_per;
    printf("%-16s %8.3f ms  %6.3f ns per wave-instr per SIMD  (x%.2f vs v_add_u32; ~%.1f cycles @2.4GHz)\n", k.n, ms, ns_per, ns_per / base, ns_per * 2.4);
  }
  return 0;
}
