// Micro-benchmark: issue cost of the VALU / SALU instructions the Goldilocks / Poseidon / NTT kernels use, at a
// CONTROLLED occupancy of 1, 2, 3, 4 and 8 waves per SIMD (the question of VERDICT r1 item 5: does a wave64 integer
// instruction issue in 2 cycles once a partner wave is resident, as v_fma_f32 does on the SIMD-32 of CDNA4?).
//
// Every kernel runs ITER iterations of ONE asm block holding UNROLL x 8 copies of one instruction on 8 independent
// register chains (one block, so the compiler's hazard recognizer cannot pad it with s_nop: between two asm statements
// that define SGPRs it inserts one).  "dep" rows put all copies on ONE chain (latency instead of throughput).
// A workgroup is 256 lanes (one wave per SIMD); the grid is (#CUs x W) workgroups and every workgroup asks for
// 160 KB / W of LDS, so exactly W workgroups (W waves per SIMD) are resident per CU for the whole run.  Each wave
// brackets its loop with s_memtime (shader clock); the table prints
//     cycles per wave-instruction per SIMD = mean wave cycles / (instructions per wave x W)
// i.e. the reciprocal throughput of one SIMD, and in brackets the same from the wall clock in ns (ratio = clock).
//
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 valu_rates.hip -o valu_rates      Run: ./valu_rates [filter]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#define ITER 512
#define UNROLL 8

#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)

// operands: %0..%7 = 32-bit chains x0..x7, %8..%15 = 64-bit chains y0..y7, %16 = b (vgpr), %17 = c (vgpr),
//           %18 = m0 (sgpr pair, written), %19 = m1 (sgpr pair, read), %20 = sc (sgpr, 32 bit)
#define K(name, STR)                                                                                                    \
  __global__ __launch_bounds__(256) void name(uint64_t* out, uint64_t* cyc, uint32_t s) {                              \
    extern __shared__ uint32_t lds_[];                                                                                  \
    uint32_t x0 = threadIdx.x + s, x1 = x0 * 3, x2 = x0 * 5, x3 = x0 * 7, x4 = x0 + 11, x5 = x0 + 13, x6 = x0 + 17, x7 = x0 + 19; \
    uint64_t y0 = x0 * 0x100000001ull, y1 = y0 * 3, y2 = y0 * 5, y3 = y0 * 7, y4 = y0 + 11, y5 = y0 + 13, y6 = y0 + 17, y7 = y0 + 19; \
    uint32_t b = (threadIdx.x * 2 + s) | 1, c = threadIdx.x + s + 3, sc = s + 5;                                        \
    uint64_t m0 = 0x5555555555555555ull * s, m1 = ~m0;                                                                  \
    if (s == 0xdeadbeef) lds_[threadIdx.x] = s;                                                                         \
    uint64_t t0 = __builtin_readcyclecounter();                                                                         \
    for (int i = 0; i < ITER; i++) {                                                                                    \
      asm volatile(STR STR STR STR STR STR STR STR                                                                      \
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7), "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), \
                     "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7), "+v"(b), "+v"(c), "+s"(m0), "+s"(m1), "+s"(sc)             \
                   : : "vcc", "scc");                                                                                   \
    }                                                                                                                   \
    uint64_t t1 = __builtin_readcyclecounter();                                                                         \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ y0 ^ y1 ^ y2 ^ y3 ^ y4 ^ y5 ^ y6 ^ y7 ^ m0 ^ m1 ^ sc ^ b ^ c; \
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + threadIdx.x / 64] = t1 - t0;                                     \
  }
K(k0, "v_fma_f32 %0, %0, %16, %17\n" "v_fma_f32 %1, %1, %16, %17\n" "v_fma_f32 %2, %2, %16, %17\n" "v_fma_f32 %3, %3, %16, %17\n" "v_fma_f32 %4, %4, %16, %17\n" "v_fma_f32 %5, %5, %16, %17\n" "v_fma_f32 %6, %6, %16, %17\n" "v_fma_f32 %7, %7, %16, %17\n" )
K(k1, "v_add_u32 %0, %0, %16\n" "v_add_u32 %1, %1, %16\n" "v_add_u32 %2, %2, %16\n" "v_add_u32 %3, %3, %16\n" "v_add_u32 %4, %4, %16\n" "v_add_u32 %5, %5, %16\n" "v_add_u32 %6, %6, %16\n" "v_add_u32 %7, %7, %16\n" )
K(k2, "v_mov_b32 %0, %16\n" "v_mov_b32 %1, %16\n" "v_mov_b32 %2, %16\n" "v_mov_b32 %3, %16\n" "v_mov_b32 %4, %16\n" "v_mov_b32 %5, %16\n" "v_mov_b32 %6, %16\n" "v_mov_b32 %7, %16\n" )
K(k3, "v_xor_b32 %0, %0, %16\n" "v_xor_b32 %1, %1, %16\n" "v_xor_b32 %2, %2, %16\n" "v_xor_b32 %3, %3, %16\n" "v_xor_b32 %4, %4, %16\n" "v_xor_b32 %5, %5, %16\n" "v_xor_b32 %6, %6, %16\n" "v_xor_b32 %7, %7, %16\n" )
K(k4, "v_add3_u32 %0, %0, %16, %17\n" "v_add3_u32 %1, %1, %16, %17\n" "v_add3_u32 %2, %2, %16, %17\n" "v_add3_u32 %3, %3, %16, %17\n" "v_add3_u32 %4, %4, %16, %17\n" "v_add3_u32 %5, %5, %16, %17\n" "v_add3_u32 %6, %6, %16, %17\n" "v_add3_u32 %7, %7, %16, %17\n" )
K(k5, "v_lshl_add_u32 %0, %0, 5, %16\n" "v_lshl_add_u32 %1, %1, 5, %16\n" "v_lshl_add_u32 %2, %2, 5, %16\n" "v_lshl_add_u32 %3, %3, 5, %16\n" "v_lshl_add_u32 %4, %4, 5, %16\n" "v_lshl_add_u32 %5, %5, 5, %16\n" "v_lshl_add_u32 %6, %6, 5, %16\n" "v_lshl_add_u32 %7, %7, 5, %16\n" )
K(k6, "v_alignbit_b32 %0, %0, %16, 7\n" "v_alignbit_b32 %1, %1, %16, 7\n" "v_alignbit_b32 %2, %2, %16, 7\n" "v_alignbit_b32 %3, %3, %16, 7\n" "v_alignbit_b32 %4, %4, %16, 7\n" "v_alignbit_b32 %5, %5, %16, 7\n" "v_alignbit_b32 %6, %6, %16, 7\n" "v_alignbit_b32 %7, %7, %16, 7\n" )
K(k7, "v_mad_u32_u24 %0, %16, %17, %0\n" "v_mad_u32_u24 %1, %16, %17, %1\n" "v_mad_u32_u24 %2, %16, %17, %2\n" "v_mad_u32_u24 %3, %16, %17, %3\n" "v_mad_u32_u24 %4, %16, %17, %4\n" "v_mad_u32_u24 %5, %16, %17, %5\n" "v_mad_u32_u24 %6, %16, %17, %6\n" "v_mad_u32_u24 %7, %16, %17, %7\n" )
K(k8, "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %1, %1, %16\n" "v_mul_lo_u32 %2, %2, %16\n" "v_mul_lo_u32 %3, %3, %16\n" "v_mul_lo_u32 %4, %4, %16\n" "v_mul_lo_u32 %5, %5, %16\n" "v_mul_lo_u32 %6, %6, %16\n" "v_mul_lo_u32 %7, %7, %16\n" )
K(k9, "v_mul_hi_u32 %0, %0, %16\n" "v_mul_hi_u32 %1, %1, %16\n" "v_mul_hi_u32 %2, %2, %16\n" "v_mul_hi_u32 %3, %3, %16\n" "v_mul_hi_u32 %4, %4, %16\n" "v_mul_hi_u32 %5, %5, %16\n" "v_mul_hi_u32 %6, %6, %16\n" "v_mul_hi_u32 %7, %7, %16\n" )
K(k10, "v_add_co_u32 %0, vcc, %0, %16\n" "v_add_co_u32 %1, vcc, %1, %16\n" "v_add_co_u32 %2, vcc, %2, %16\n" "v_add_co_u32 %3, vcc, %3, %16\n" "v_add_co_u32 %4, vcc, %4, %16\n" "v_add_co_u32 %5, vcc, %5, %16\n" "v_add_co_u32 %6, vcc, %6, %16\n" "v_add_co_u32 %7, vcc, %7, %16\n" )
K(k11, "v_addc_co_u32 %0, vcc, %0, %16, vcc\n" "v_addc_co_u32 %1, vcc, %1, %16, vcc\n" "v_addc_co_u32 %2, vcc, %2, %16, vcc\n" "v_addc_co_u32 %3, vcc, %3, %16, vcc\n" "v_addc_co_u32 %4, vcc, %4, %16, vcc\n" "v_addc_co_u32 %5, vcc, %5, %16, vcc\n" "v_addc_co_u32 %6, vcc, %6, %16, vcc\n" "v_addc_co_u32 %7, vcc, %7, %16, vcc\n" )
K(k12, "v_sub_co_u32 %0, vcc, %0, %16\n" "v_sub_co_u32 %1, vcc, %1, %16\n" "v_sub_co_u32 %2, vcc, %2, %16\n" "v_sub_co_u32 %3, vcc, %3, %16\n" "v_sub_co_u32 %4, vcc, %4, %16\n" "v_sub_co_u32 %5, vcc, %5, %16\n" "v_sub_co_u32 %6, vcc, %6, %16\n" "v_sub_co_u32 %7, vcc, %7, %16\n" )
K(k13, "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %1, %18, %1, %16\n" "v_add_co_u32 %2, %18, %2, %16\n" "v_add_co_u32 %3, %18, %3, %16\n" "v_add_co_u32 %4, %18, %4, %16\n" "v_add_co_u32 %5, %18, %5, %16\n" "v_add_co_u32 %6, %18, %6, %16\n" "v_add_co_u32 %7, %18, %7, %16\n" )
K(k14, "v_addc_co_u32 %0, %18, %0, %16, %19\n" "v_addc_co_u32 %1, %18, %1, %16, %19\n" "v_addc_co_u32 %2, %18, %2, %16, %19\n" "v_addc_co_u32 %3, %18, %3, %16, %19\n" "v_addc_co_u32 %4, %18, %4, %16, %19\n" "v_addc_co_u32 %5, %18, %5, %16, %19\n" "v_addc_co_u32 %6, %18, %6, %16, %19\n" "v_addc_co_u32 %7, %18, %7, %16, %19\n" )
K(k15, "v_cndmask_b32 %0, %0, %16, vcc\n" "v_cndmask_b32 %1, %1, %16, vcc\n" "v_cndmask_b32 %2, %2, %16, vcc\n" "v_cndmask_b32 %3, %3, %16, vcc\n" "v_cndmask_b32 %4, %4, %16, vcc\n" "v_cndmask_b32 %5, %5, %16, vcc\n" "v_cndmask_b32 %6, %6, %16, vcc\n" "v_cndmask_b32 %7, %7, %16, vcc\n" )
K(k16, "v_cndmask_b32 %0, %0, %16, %19\n" "v_cndmask_b32 %1, %1, %16, %19\n" "v_cndmask_b32 %2, %2, %16, %19\n" "v_cndmask_b32 %3, %3, %16, %19\n" "v_cndmask_b32 %4, %4, %16, %19\n" "v_cndmask_b32 %5, %5, %16, %19\n" "v_cndmask_b32 %6, %6, %16, %19\n" "v_cndmask_b32 %7, %7, %16, %19\n" )
K(k17, "v_cmp_lt_u32 vcc, %0, %16\n" "v_cmp_lt_u32 vcc, %1, %16\n" "v_cmp_lt_u32 vcc, %2, %16\n" "v_cmp_lt_u32 vcc, %3, %16\n" "v_cmp_lt_u32 vcc, %4, %16\n" "v_cmp_lt_u32 vcc, %5, %16\n" "v_cmp_lt_u32 vcc, %6, %16\n" "v_cmp_lt_u32 vcc, %7, %16\n" )
K(k18, "v_cmp_lt_u64 vcc, %8, %15\n" "v_cmp_lt_u64 vcc, %9, %15\n" "v_cmp_lt_u64 vcc, %10, %15\n" "v_cmp_lt_u64 vcc, %11, %15\n" "v_cmp_lt_u64 vcc, %12, %15\n" "v_cmp_lt_u64 vcc, %13, %15\n" "v_cmp_lt_u64 vcc, %14, %15\n" "v_cmp_lt_u64 vcc, %15, %15\n" )
K(k19, "v_cmp_lt_u64 %18, %8, %15\n" "v_cmp_lt_u64 %18, %9, %15\n" "v_cmp_lt_u64 %18, %10, %15\n" "v_cmp_lt_u64 %18, %11, %15\n" "v_cmp_lt_u64 %18, %12, %15\n" "v_cmp_lt_u64 %18, %13, %15\n" "v_cmp_lt_u64 %18, %14, %15\n" "v_cmp_lt_u64 %18, %15, %15\n" )
K(k20, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %9, vcc, %16, %17, %9\n" "v_mad_u64_u32 %10, vcc, %16, %17, %10\n" "v_mad_u64_u32 %11, vcc, %16, %17, %11\n" "v_mad_u64_u32 %12, vcc, %16, %17, %12\n" "v_mad_u64_u32 %13, vcc, %16, %17, %13\n" "v_mad_u64_u32 %14, vcc, %16, %17, %14\n" "v_mad_u64_u32 %15, vcc, %16, %17, %15\n" )
K(k21, "v_mad_u64_u32 %8, vcc, %16, 41, %8\n" "v_mad_u64_u32 %9, vcc, %16, 41, %9\n" "v_mad_u64_u32 %10, vcc, %16, 41, %10\n" "v_mad_u64_u32 %11, vcc, %16, 41, %11\n" "v_mad_u64_u32 %12, vcc, %16, 41, %12\n" "v_mad_u64_u32 %13, vcc, %16, 41, %13\n" "v_mad_u64_u32 %14, vcc, %16, 41, %14\n" "v_mad_u64_u32 %15, vcc, %16, 41, %15\n" )
K(k22, "v_mad_u64_u32 %8, vcc, %16, %20, %8\n" "v_mad_u64_u32 %9, vcc, %16, %20, %9\n" "v_mad_u64_u32 %10, vcc, %16, %20, %10\n" "v_mad_u64_u32 %11, vcc, %16, %20, %11\n" "v_mad_u64_u32 %12, vcc, %16, %20, %12\n" "v_mad_u64_u32 %13, vcc, %16, %20, %13\n" "v_mad_u64_u32 %14, vcc, %16, %20, %14\n" "v_mad_u64_u32 %15, vcc, %16, %20, %15\n" )
K(k23, "v_mad_u64_u32 %8, %18, %16, %17, %8\n" "v_mad_u64_u32 %9, %18, %16, %17, %9\n" "v_mad_u64_u32 %10, %18, %16, %17, %10\n" "v_mad_u64_u32 %11, %18, %16, %17, %11\n" "v_mad_u64_u32 %12, %18, %16, %17, %12\n" "v_mad_u64_u32 %13, %18, %16, %17, %13\n" "v_mad_u64_u32 %14, %18, %16, %17, %14\n" "v_mad_u64_u32 %15, %18, %16, %17, %15\n" )
K(k24, "v_mad_u64_u32 %8, vcc, %16, -1, %8\n" "v_mad_u64_u32 %9, vcc, %16, -1, %9\n" "v_mad_u64_u32 %10, vcc, %16, -1, %10\n" "v_mad_u64_u32 %11, vcc, %16, -1, %11\n" "v_mad_u64_u32 %12, vcc, %16, -1, %12\n" "v_mad_u64_u32 %13, vcc, %16, -1, %13\n" "v_mad_u64_u32 %14, vcc, %16, -1, %14\n" "v_mad_u64_u32 %15, vcc, %16, -1, %15\n" )
K(k25, "v_mad_u64_u32 %8, vcc, %16, %17, 0\n" "v_mad_u64_u32 %9, vcc, %16, %17, 0\n" "v_mad_u64_u32 %10, vcc, %16, %17, 0\n" "v_mad_u64_u32 %11, vcc, %16, %17, 0\n" "v_mad_u64_u32 %12, vcc, %16, %17, 0\n" "v_mad_u64_u32 %13, vcc, %16, %17, 0\n" "v_mad_u64_u32 %14, vcc, %16, %17, 0\n" "v_mad_u64_u32 %15, vcc, %16, %17, 0\n" )
K(k26, "v_lshl_add_u64 %8, %8, 3, %15\n" "v_lshl_add_u64 %9, %9, 3, %15\n" "v_lshl_add_u64 %10, %10, 3, %15\n" "v_lshl_add_u64 %11, %11, 3, %15\n" "v_lshl_add_u64 %12, %12, 3, %15\n" "v_lshl_add_u64 %13, %13, 3, %15\n" "v_lshl_add_u64 %14, %14, 3, %15\n" "v_lshl_add_u64 %15, %15, 3, %15\n" )
K(k27, "v_lshrrev_b64 %8, 3, %8\n" "v_lshrrev_b64 %9, 3, %9\n" "v_lshrrev_b64 %10, 3, %10\n" "v_lshrrev_b64 %11, 3, %11\n" "v_lshrrev_b64 %12, 3, %12\n" "v_lshrrev_b64 %13, 3, %13\n" "v_lshrrev_b64 %14, 3, %14\n" "v_lshrrev_b64 %15, 3, %15\n" )
K(k28, "v_mov_b32_dpp %0, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %1, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %2, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %3, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %4, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %5, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %6, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" "v_mov_b32_dpp %7, %16 row_ror:3 row_mask:0xf bank_mask:0xf\n" )
K(k29, "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" "s_nop 0\n" )
K(k30, "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" "s_mov_b32 %20, 17\n" )
K(k31, "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" "s_add_u32 %20, %20, 17\n" )
K(k32, "v_add_co_u32 %0, %18, %0, %16\n s_nop 1\n v_addc_co_u32 %0, %18, %0, %16, %18\n" "v_add_co_u32 %1, %18, %1, %16\n s_nop 1\n v_addc_co_u32 %1, %18, %1, %16, %18\n" "v_add_co_u32 %2, %18, %2, %16\n s_nop 1\n v_addc_co_u32 %2, %18, %2, %16, %18\n" "v_add_co_u32 %3, %18, %3, %16\n s_nop 1\n v_addc_co_u32 %3, %18, %3, %16, %18\n" "v_add_co_u32 %4, %18, %4, %16\n s_nop 1\n v_addc_co_u32 %4, %18, %4, %16, %18\n" "v_add_co_u32 %5, %18, %5, %16\n s_nop 1\n v_addc_co_u32 %5, %18, %5, %16, %18\n" "v_add_co_u32 %6, %18, %6, %16\n s_nop 1\n v_addc_co_u32 %6, %18, %6, %16, %18\n" "v_add_co_u32 %7, %18, %7, %16\n s_nop 1\n v_addc_co_u32 %7, %18, %7, %16, %18\n" )
K(k33, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n v_add_co_u32 %0, %18, %0, %16\n" "v_mad_u64_u32 %9, vcc, %16, %17, %9\n v_add_co_u32 %1, %18, %1, %16\n" "v_mad_u64_u32 %10, vcc, %16, %17, %10\n v_add_co_u32 %2, %18, %2, %16\n" "v_mad_u64_u32 %11, vcc, %16, %17, %11\n v_add_co_u32 %3, %18, %3, %16\n" "v_mad_u64_u32 %12, vcc, %16, %17, %12\n v_add_co_u32 %4, %18, %4, %16\n" "v_mad_u64_u32 %13, vcc, %16, %17, %13\n v_add_co_u32 %5, %18, %5, %16\n" "v_mad_u64_u32 %14, vcc, %16, %17, %14\n v_add_co_u32 %6, %18, %6, %16\n" "v_mad_u64_u32 %15, vcc, %16, %17, %15\n v_add_co_u32 %7, %18, %7, %16\n" )
K(k34, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n v_add_u32 %0, %0, %16\n" "v_mad_u64_u32 %9, vcc, %16, %17, %9\n v_add_u32 %1, %1, %16\n" "v_mad_u64_u32 %10, vcc, %16, %17, %10\n v_add_u32 %2, %2, %16\n" "v_mad_u64_u32 %11, vcc, %16, %17, %11\n v_add_u32 %3, %3, %16\n" "v_mad_u64_u32 %12, vcc, %16, %17, %12\n v_add_u32 %4, %4, %16\n" "v_mad_u64_u32 %13, vcc, %16, %17, %13\n v_add_u32 %5, %5, %16\n" "v_mad_u64_u32 %14, vcc, %16, %17, %14\n v_add_u32 %6, %6, %16\n" "v_mad_u64_u32 %15, vcc, %16, %17, %15\n v_add_u32 %7, %7, %16\n" )
K(k35, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n s_nop 0\n" "v_mad_u64_u32 %9, vcc, %16, %17, %9\n s_nop 0\n" "v_mad_u64_u32 %10, vcc, %16, %17, %10\n s_nop 0\n" "v_mad_u64_u32 %11, vcc, %16, %17, %11\n s_nop 0\n" "v_mad_u64_u32 %12, vcc, %16, %17, %12\n s_nop 0\n" "v_mad_u64_u32 %13, vcc, %16, %17, %13\n s_nop 0\n" "v_mad_u64_u32 %14, vcc, %16, %17, %14\n s_nop 0\n" "v_mad_u64_u32 %15, vcc, %16, %17, %15\n s_nop 0\n" )
K(k36, "v_add_u32 %0, %0, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %1, %1, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %2, %2, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %3, %3, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %4, %4, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %5, %5, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %6, %6, %16\n s_mov_b32 %20, 17\n" "v_add_u32 %7, %7, %16\n s_mov_b32 %20, 17\n" )
K(k37, "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" "v_add_u32 %0, %0, %16\n" )
K(k38, "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_mad_u64_u32 %8, vcc, %16, %17, %8\n" )
K(k39, "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" "v_mul_lo_u32 %0, %0, %16\n" )
K(k40, "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" "v_add_co_u32 %0, %18, %0, %16\n" )

K(k41, "v_cmp_lt_u32 vcc, %0, %16\n v_cndmask_b32 %0, %0, %17, vcc\n" "v_cmp_lt_u32 vcc, %1, %16\n v_cndmask_b32 %1, %1, %17, vcc\n" "v_cmp_lt_u32 vcc, %2, %16\n v_cndmask_b32 %2, %2, %17, vcc\n" "v_cmp_lt_u32 vcc, %3, %16\n v_cndmask_b32 %3, %3, %17, vcc\n" "v_cmp_lt_u32 vcc, %4, %16\n v_cndmask_b32 %4, %4, %17, vcc\n" "v_cmp_lt_u32 vcc, %5, %16\n v_cndmask_b32 %5, %5, %17, vcc\n" "v_cmp_lt_u32 vcc, %6, %16\n v_cndmask_b32 %6, %6, %17, vcc\n" "v_cmp_lt_u32 vcc, %7, %16\n v_cndmask_b32 %7, %7, %17, vcc\n" )
K(k42, "v_cmp_lt_u64 vcc, %8, %15\n v_cndmask_b32 %0, %0, %17, vcc\n v_cndmask_b32 %16, %16, %0, vcc\n" "v_cmp_lt_u64 vcc, %9, %15\n v_cndmask_b32 %1, %1, %17, vcc\n v_cndmask_b32 %16, %16, %1, vcc\n" "v_cmp_lt_u64 vcc, %10, %15\n v_cndmask_b32 %2, %2, %17, vcc\n v_cndmask_b32 %16, %16, %2, vcc\n" "v_cmp_lt_u64 vcc, %11, %15\n v_cndmask_b32 %3, %3, %17, vcc\n v_cndmask_b32 %16, %16, %3, vcc\n" "v_cmp_lt_u64 vcc, %12, %15\n v_cndmask_b32 %4, %4, %17, vcc\n v_cndmask_b32 %16, %16, %4, vcc\n" "v_cmp_lt_u64 vcc, %13, %15\n v_cndmask_b32 %5, %5, %17, vcc\n v_cndmask_b32 %16, %16, %5, vcc\n" "v_cmp_lt_u64 vcc, %14, %15\n v_cndmask_b32 %6, %6, %17, vcc\n v_cndmask_b32 %16, %16, %6, vcc\n" "v_cmp_lt_u64 vcc, %15, %15\n v_cndmask_b32 %7, %7, %17, vcc\n v_cndmask_b32 %16, %16, %7, vcc\n" )
K(k43, "v_cndmask_b32_e64 %0, %0, %16, vcc\n" "v_cndmask_b32_e64 %1, %1, %16, vcc\n" "v_cndmask_b32_e64 %2, %2, %16, vcc\n" "v_cndmask_b32_e64 %3, %3, %16, vcc\n" "v_cndmask_b32_e64 %4, %4, %16, vcc\n" "v_cndmask_b32_e64 %5, %5, %16, vcc\n" "v_cndmask_b32_e64 %6, %6, %16, vcc\n" "v_cndmask_b32_e64 %7, %7, %16, vcc\n" )
K(k44, "v_cmp_lt_u32 %18, %0, %16\n v_cndmask_b32 %0, %0, %17, %18\n" "v_cmp_lt_u32 %18, %1, %16\n v_cndmask_b32 %1, %1, %17, %18\n" "v_cmp_lt_u32 %18, %2, %16\n v_cndmask_b32 %2, %2, %17, %18\n" "v_cmp_lt_u32 %18, %3, %16\n v_cndmask_b32 %3, %3, %17, %18\n" "v_cmp_lt_u32 %18, %4, %16\n v_cndmask_b32 %4, %4, %17, %18\n" "v_cmp_lt_u32 %18, %5, %16\n v_cndmask_b32 %5, %5, %17, %18\n" "v_cmp_lt_u32 %18, %6, %16\n v_cndmask_b32 %6, %6, %17, %18\n" "v_cmp_lt_u32 %18, %7, %16\n v_cndmask_b32 %7, %7, %17, %18\n" )
K(k45, "v_cmp_lt_u32 %18, %0, %16\n s_nop 1\n v_cndmask_b32 %0, %0, %17, %18\n" "v_cmp_lt_u32 %18, %1, %16\n s_nop 1\n v_cndmask_b32 %1, %1, %17, %18\n" "v_cmp_lt_u32 %18, %2, %16\n s_nop 1\n v_cndmask_b32 %2, %2, %17, %18\n" "v_cmp_lt_u32 %18, %3, %16\n s_nop 1\n v_cndmask_b32 %3, %3, %17, %18\n" "v_cmp_lt_u32 %18, %4, %16\n s_nop 1\n v_cndmask_b32 %4, %4, %17, %18\n" "v_cmp_lt_u32 %18, %5, %16\n s_nop 1\n v_cndmask_b32 %5, %5, %17, %18\n" "v_cmp_lt_u32 %18, %6, %16\n s_nop 1\n v_cndmask_b32 %6, %6, %17, %18\n" "v_cmp_lt_u32 %18, %7, %16\n s_nop 1\n v_cndmask_b32 %7, %7, %17, %18\n" )
K(k46, "v_add_co_u32 %0, vcc, %0, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %1, vcc, %1, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %2, vcc, %2, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %3, vcc, %3, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %4, vcc, %4, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %5, vcc, %5, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %6, vcc, %6, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" "v_add_co_u32 %7, vcc, %7, %16\n v_addc_co_u32 %17, vcc, %17, %16, vcc\n" )
K(k47, "v_and_b32 %0, %0, %16\n" "v_and_b32 %1, %1, %16\n" "v_and_b32 %2, %2, %16\n" "v_and_b32 %3, %3, %16\n" "v_and_b32 %4, %4, %16\n" "v_and_b32 %5, %5, %16\n" "v_and_b32 %6, %6, %16\n" "v_and_b32 %7, %7, %16\n" )
K(k48, "v_lshlrev_b32 %0, 3, %0\n" "v_lshlrev_b32 %1, 3, %1\n" "v_lshlrev_b32 %2, 3, %2\n" "v_lshlrev_b32 %3, 3, %3\n" "v_lshlrev_b32 %4, 3, %4\n" "v_lshlrev_b32 %5, 3, %5\n" "v_lshlrev_b32 %6, 3, %6\n" "v_lshlrev_b32 %7, 3, %7\n" )
K(k49, "v_sub_u32 %0, %0, %16\n" "v_sub_u32 %1, %1, %16\n" "v_sub_u32 %2, %2, %16\n" "v_sub_u32 %3, %3, %16\n" "v_sub_u32 %4, %4, %16\n" "v_sub_u32 %5, %5, %16\n" "v_sub_u32 %6, %6, %16\n" "v_sub_u32 %7, %7, %16\n" )
K(k50, "v_add_u32 %0, %0, %16\n v_xor_b32 %0, %0, %17\n v_add_u32 %0, %0, %17\n v_mad_u64_u32 %8, vcc, %16, %17, %8\n" "v_add_u32 %1, %1, %16\n v_xor_b32 %1, %1, %17\n v_add_u32 %1, %1, %17\n v_mad_u64_u32 %9, vcc, %16, %17, %9\n" "v_add_u32 %2, %2, %16\n v_xor_b32 %2, %2, %17\n v_add_u32 %2, %2, %17\n v_mad_u64_u32 %10, vcc, %16, %17, %10\n" "v_add_u32 %3, %3, %16\n v_xor_b32 %3, %3, %17\n v_add_u32 %3, %3, %17\n v_mad_u64_u32 %11, vcc, %16, %17, %11\n" "v_add_u32 %4, %4, %16\n v_xor_b32 %4, %4, %17\n v_add_u32 %4, %4, %17\n v_mad_u64_u32 %12, vcc, %16, %17, %12\n" "v_add_u32 %5, %5, %16\n v_xor_b32 %5, %5, %17\n v_add_u32 %5, %5, %17\n v_mad_u64_u32 %13, vcc, %16, %17, %13\n" "v_add_u32 %6, %6, %16\n v_xor_b32 %6, %6, %17\n v_add_u32 %6, %6, %17\n v_mad_u64_u32 %14, vcc, %16, %17, %14\n" "v_add_u32 %7, %7, %16\n v_xor_b32 %7, %7, %17\n v_add_u32 %7, %7, %17\n v_mad_u64_u32 %15, vcc, %16, %17, %15\n" )

struct Kern { const char* name; void (*f)(uint64_t*, uint64_t*, uint32_t); int instr_per_copy; };

int main(int argc, char** argv) {
  const char* filter = argc > 1 ? argv[1] : nullptr;
  uint64_t *d, *dc;
  hipDeviceProp_t prop; HIPC(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const int max_blocks = cus * 8;
  HIPC(hipMalloc(&d, (size_t)max_blocks * 256 * 8));
  HIPC(hipMalloc(&dc, (size_t)max_blocks * 4 * 8));
  hipEvent_t e0, e1; HIPC(hipEventCreate(&e0)); HIPC(hipEventCreate(&e1));
  printf("# device %s, %d CUs, clockRate %d kHz; per wave: ITER %d x UNROLL %d x 8 copies\n", prop.gcnArchName, cus, prop.clockRate, ITER, UNROLL);
  printf("# cells: shader-clock cycles per wave-instruction per SIMD (wall-clock ns per wave-instruction per SIMD) at W waves per SIMD\n");
  Kern ks[] = {{"v_fma_f32 (control)", k0, 1},
    {"v_add_u32", k1, 1},
    {"v_mov_b32", k2, 1},
    {"v_xor_b32", k3, 1},
    {"v_add3_u32", k4, 1},
    {"v_lshl_add_u32", k5, 1},
    {"v_alignbit_b32", k6, 1},
    {"v_mad_u32_u24", k7, 1},
    {"v_mul_lo_u32", k8, 1},
    {"v_mul_hi_u32", k9, 1},
    {"v_add_co_u32 ->vcc", k10, 1},
    {"v_addc_co_u32 vcc->vcc", k11, 1},
    {"v_sub_co_u32 ->vcc", k12, 1},
    {"v_add_co_u32 ->sgpr (VOP3)", k13, 1},
    {"v_addc_co_u32 sgpr->sgpr' (VOP3)", k14, 1},
    {"v_cndmask_b32 vcc", k15, 1},
    {"v_cndmask_b32 sgpr (VOP3)", k16, 1},
    {"v_cmp_lt_u32 ->vcc", k17, 1},
    {"v_cmp_lt_u64 ->vcc", k18, 1},
    {"v_cmp_lt_u64 ->sgpr", k19, 1},
    {"v_mad_u64_u32 v,v (+vcc)", k20, 1},
    {"v_mad_u64_u32 v,inline 41", k21, 1},
    {"v_mad_u64_u32 v,sgpr", k22, 1},
    {"v_mad_u64_u32 carry->sgpr", k23, 1},
    {"v_mad_u64_u32 v,-1", k24, 1},
    {"v_mad_u64_u32 addend 0", k25, 1},
    {"v_lshl_add_u64", k26, 1},
    {"v_lshrrev_b64", k27, 1},
    {"v_mov_b32 dpp row_ror", k28, 1},
    {"s_nop 0", k29, 1},
    {"s_mov_b32", k30, 1},
    {"s_add_u32", k31, 1},
    {"add_co->sgpr; s_nop 1; addc<-sgpr (3 instr)", k32, 3},
    {"v_mad_u64_u32 + v_add_co->sgpr (2 instr)", k33, 2},
    {"v_mad_u64_u32 + v_add_u32 (2 instr)", k34, 2},
    {"v_mad_u64_u32 + s_nop 0 (2 instr)", k35, 2},
    {"v_add_u32 + s_mov_b32 (2 instr)", k36, 2},
    {"v_cmp_lt_u32->vcc + v_cndmask_e32 vcc (2 instr)", k41, 2},
    {"v_cmp_lt_u64->vcc + 2x v_cndmask_e32 vcc (3 instr)", k42, 3},
    {"v_cndmask_b32_e64 mask=vcc", k43, 1},
    {"v_cmp_lt_u32->sgpr + v_cndmask_e64 sgpr (2 instr)", k44, 2},
    {"v_cmp_lt_u32->sgpr + s_nop 1 + v_cndmask_e64 (3 instr)", k45, 3},
    {"v_add_co->vcc + v_addc vcc (2 instr)", k46, 2},
    {"v_and_b32", k47, 1},
    {"v_lshlrev_b32", k48, 1},
    {"v_sub_u32", k49, 1},
    {"v_add_u32 x3 + v_mad_u64_u32 (4 instr)", k50, 4},
    {"dep: v_add_u32", k37, 1},
    {"dep: v_mad_u64_u32 (addend chain)", k38, 1},
    {"dep: v_mul_lo_u32", k39, 1},
    {"dep: v_add_co_u32 ->sgpr", k40, 1}};
  const int Ws[] = {1, 2, 3, 4, 8};
  printf("%-46s", "instruction");
  for (int w : Ws) printf("  W=%d cyc   (ns)  ", w);
  printf("\n");
  std::vector<uint64_t> hc((size_t)max_blocks * 4);
  for (auto& k : ks) {
    if (filter && !strstr(k.name, filter)) continue;
    printf("%-46s", k.name);
    for (int w : Ws) {
      const int blocks = cus * w;
      const size_t lds = (size_t)(160 * 1024 / w) & ~(size_t)255;
      HIPC(hipFuncSetAttribute((const void*)k.f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), lds, 0, d, dc, 1u);
      HIPC(hipDeviceSynchronize());
      HIPC(hipEventRecord(e0));
      hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), lds, 0, d, dc, 1u);
      HIPC(hipEventRecord(e1)); HIPC(hipEventSynchronize(e1));
      float ms; HIPC(hipEventElapsedTime(&ms, e0, e1));
      HIPC(hipMemcpy(hc.data(), dc, (size_t)blocks * 4 * 8, hipMemcpyDeviceToHost));
      double sum = 0; for (int i = 0; i < blocks * 4; i++) sum += (double)hc[i];
      const double wave_cycles = sum / (blocks * 4);
      const double n_instr = (double)ITER * UNROLL * 8 * k.instr_per_copy;
      printf("  %6.2f (%5.2f)  ", wave_cycles / (n_instr * w), ms * 1e6 / (n_instr * w));
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
