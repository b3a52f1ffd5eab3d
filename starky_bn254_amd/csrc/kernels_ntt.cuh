// K1: the transform kernels of the prover path (gfx950) and the weak Goldilocks arithmetic (namespace nw) they share with the other
// kernels.  Included by kernels.cuh (declarations; prover.hip launches the kernels) and by ntt.hip (SBN_NTT_KERNELS_HERE: the
// definitions).  The kernels live in their own translation unit because they are compiled with the max-ILP machine scheduler
// (Makefile: -mllvm -amdgpu-sched-strategy=max-ilp): a DFT-16 is ~650 single-instruction asm primitives whose carries sit in SGPR pairs,
// the default scheduler emits them as one dependent chain through ONE pair with a wait state after almost every instruction (0.6 s_nop
// per VALU instruction), the max-ILP one interleaves independent butterflies over many pairs (fast pass: 1,820 -> 959 s_nop, same
// VALU count, 56 -> 60 VGPRs).  The same flag on the whole of prover.hip makes the quotient and witness kernels spill, hence the split.
#pragma once
#include "gl.cuh"

// =================================================================================================
// K1  batched NTT pass  (replaces plonky2_field fft.rs / PolynomialValues::ifft / coset_fft; P1,P2,P3,P5)
// An n-point transform per column is n = R*S: pass A does R-point NTTs along the stride-S axis for a
// tile of T consecutive inner indices, multiplies by the inter-pass twiddles and stores in place
// layout; pass B does the S-point NTTs along the contiguous axis and stores transposed.  Both passes
// stage an [R][T] tile through LDS (radix-2 DIF, bit-reversed on store), so every global access is a
// run of T*8 = 128 contiguous bytes per row of the tile.
// Algorithmic bytes per pass: 8*n read + 8*n written per column (zero-padded inputs read n_in).
// =================================================================================================
struct NttPassParams {
  const u64* in; u64* out;
  size_t in_col_stride, out_col_stride;  // elements between columns
  u32 log_r;        // log2 of NTT length in this pass
  u32 log_t;        // log2 of tile width
  u32 log_n;        // log2 of the full transform length
  size_t in_sr, in_st, out_sr, out_st;   // element strides of (r, t) in input / output
  size_t n_in;      // input indices >= n_in read as zero (zero-padded LDE)
  u32 r_fast_load;  // 1: r is the contiguous input axis (pass B)
  u32 twiddle;      // 1: multiply output (k, t) by w_n^(k*t_global) (pass A)
  const u64* tw;    // root powers w_M^e, e < M/2, of the table's size M = 2^tw_log (inverse table for iNTT)
  u32 tw_log;
  const u64* pre;   // optional per-index input scale (coset shift powers), indexed by input index
  const u64* post;  // optional per-index output scale, indexed by output index
  u64 scale;        // scalar output scale (1/n for inverse); 1 = none
  // fast pass only.  split = 1: this launch is one 1,024-point pass over an input whose rows 512.. are zero (the coset LDE
  // of a 2^18-row table: 2^19 = 1,024 x 512), run as TWO 512-point passes on grid.z: DFT_1024(x)[2j + v] = DFT_512(x_r
  // w_1024^(v r))[j].  Half v reads the same 512 rows, scaled by pre (v = 0) or pre2 (v = 1: pre2[i] = pre[i] w_1024^(row of
  // i)), and writes the output rows of parity v.  The 64-values-per-lane form of a register-resident 1,024-point pass needs
  // twice the registers of the 512-point one and loses its place beside the sponge waves; this form keeps
  // the 512-point kernel's footprint and reads the (cache-resident) input twice.
  const u64* pre2;
  u32 split;
  // fast pass only.  xcd_order = 1 (default; SBN_NTT_XCD=0 turns it off): the grid is (columns, tiles) instead of (tiles,
  // columns): consecutive workgroups -- which the dispatcher deals round-robin over the 8 XCDs -- then work on the SAME tile
  // of different columns, i.e. read the same twiddle / coset-scale table entries while each XCD keeps a fixed set of columns
  // (blockIdx.x % 8 when the column count is a multiple of 8).  Measured: 1 % off the proof, five A/B pairs of
  // five; what it does NOT do is keep a pass's output in L2 for the next launch (SBN_NTT_SUB experiment, same section).
  u32 xcd_order;
};

GL_HD F tw_lookup(const u64* tw, u32 tw_log, u64 e_of_order, u32 order_log) {
  // returns w_{2^order_log}^e using the half table of w_{2^tw_log}
  u64 idx = (e_of_order << (tw_log - order_log)) & ((1ULL << tw_log) - 1);
  u64 half = 1ULL << (tw_log - 1);
  if (idx >= half) return -F(tw[idx - half]);
  return F(tw[idx]);
}

// The same root from a FULL table (w^i, i < 2^tw_log): two 32-bit index instructions and the load, where the half table costs
// a 64-bit compare, a negation and two selects (~17 VALU instructions per root, two roots per element and pass).
__device__ __forceinline__ u64 tw_full(const u64* __restrict__ tw, u32 tw_log, u32 e_of_order, u32 order_log) {
  return tw[(e_of_order << (tw_log - order_log)) & ((1u << tw_log) - 1u)];
}

static constexpr int NTT_THREADS = 256;

#ifdef SBN_NTT_KERNELS_HERE
__global__ __launch_bounds__(NTT_THREADS) void ntt_pass_kernel(NttPassParams p) {
  extern __shared__ u64 lds[];
  const u32 R = 1u << p.log_r, T = 1u << p.log_t;
  const u32 LDT = T + 1;  // padded row
  const size_t col = blockIdx.y;
  const size_t t0 = (size_t)blockIdx.x << p.log_t;
  const u64* in = p.in + col * p.in_col_stride;
  u64* out = p.out + col * p.out_col_stride;
  const u32 total = R << p.log_t;
  // ---- load tile
  for (u32 e = threadIdx.x; e < total; e += NTT_THREADS) {
    u32 r, t;
    if (p.r_fast_load) { r = e & (R - 1); t = e >> p.log_r; } else { t = e & (T - 1); r = e >> p.log_t; }
    size_t gi = (size_t)r * p.in_sr + (t0 + t) * p.in_st;
    u64 v = 0;
    if (gi < p.n_in) {
      v = in[gi];
      if (p.pre) v = (F(v) * F(p.pre[gi])).v;
    }
    lds[r * LDT + t] = v;
  }
  __syncthreads();
  // ---- radix-2 DIF along r
  for (u32 s = 0; s < p.log_r; s++) {
    u32 half = R >> (s + 1);
    u32 nb = total >> 1;
    for (u32 e = threadIdx.x; e < nb; e += NTT_THREADS) {
      u32 t = e & (T - 1), b = e >> p.log_t;
      u32 j = b & (half - 1), blk = b / half;
      u32 i0 = blk * 2 * half + j, i1 = i0 + half;
      F u{lds[i0 * LDT + t]}, v(lds[i1 * LDT + t]);
      F w = tw_lookup(p.tw, p.tw_log, (u64)j << s, p.log_r);
      lds[i0 * LDT + t] = (u + v).v;
      lds[i1 * LDT + t] = ((u - v) * w).v;
    }
    __syncthreads();
  }
  // ---- store (LDS row q holds frequency bitrev(q))
  for (u32 e = threadIdx.x; e < total; e += NTT_THREADS) {
    u32 t = e & (T - 1), k = e >> p.log_t;
    u32 q = bitrev32(k, p.log_r);
    F v{lds[q * LDT + t]};
    if (p.twiddle) v = v * tw_lookup(p.tw, p.tw_log, (u64)k * (t0 + t), p.log_n);
    if (p.scale != 1) v = v * F(p.scale);
    size_t go = (size_t)k * p.out_sr + (t0 + t) * p.out_st;
    if (p.post) v = v * F(p.post[go]);
    out[go] = v.v;
  }
}
#else
__global__ void ntt_pass_kernel(NttPassParams p);
#endif

// -------------------------------------------------------------------------------------------------------------
// K1' fast pass for R = 256 / 512 (the sizes the 2^16..2^18-row tables use): a 256-point NTT is two rounds of
// 16-point DFTs held in REGISTERS.  In Goldilocks 2 has order 192, so rho = 2^12 is a primitive 16th root of unity
// and every twiddle inside a 16-point DFT is a power of two: the butterflies need only shifts and one 128-bit
// reduction, no multiplications.  The reference's root is omega_16 = rho^13 (omega_16^-1 = rho^3), so
// DFT_omega[k] = DFT_rho[13 k mod 16]: the same values in a permuted order, which is absorbed into the store index
// (k = 5 m for the forward transform, 11 m for the inverse, m = index under rho).  One general twiddle
// omega_256^(g k1) per element sits between the two rounds, which exchange data through LDS once.  R = 512 adds one
// leading radix-2 stage (free for the zero-padded LDE, whose upper half is zero).  Values stay "weak" (any u64
// congruent mod p) until the final store.  ~200 instructions per element per pass instead of ~450.
// -------------------------------------------------------------------------------------------------------------
namespace nw {
// Weak Goldilocks arithmetic (any u64 congruent to the value) from SINGLE-INSTRUCTION asm primitives: carries and borrows
// stay in SGPR pairs (v_add_co / v_addc / v_sub_co / v_subb with explicit carry operands) and are repaid by
// v_mad_u64_u32 x, -1 (x + c (2^32 - 1)) instead of the 64-bit compare + two selects the compiler emits for
// `if (s < a) s += EPS`.  The compiler still allocates every register, schedules, and pads the SGPR write -> read wait states
// itself (it sees each primitive's operands), so there is no fixed register window.  A 16-point DFT drops from 942 VALU
// instructions (+213 s_nop) to 656, a general multiply from 33 to 21 (round 3: 13, gl_mul_weak).  Set SBN_NTT_CXX_ARITH at compile time for the plain
// C++ forms (A/B measurements).
#if !defined(SBN_NTT_CXX_ARITH)
using namespace gp;   // single-instruction primitives (gl.cuh)
// x + carry * (2^32 - 1), twice (the first repayment can wrap again only from the top 2^32 values).  Round 4 measured what the
// second repayment costs and whether an exact form can avoid it (profiles/r4_ntt_repay_ab.txt): leaving it out (wrong once in ~2^33
// operations) makes a G1 proof 0.59 ms shorter; a wave-uniform branch on its carry mask gives nothing back; ORing the mask into a flag
// and re-running a flagged tile exactly is 5.8 ms SLOWER (64 scalar ORs per DFT-16 on the CU's one scalar unit, which the sponge needs).
__device__ __forceinline__ u64 repay_carry(u64 s, u64 k) { u64 k2, k3; u64 t = madm1(sel01(k), s, k2); return madm1(sel01(k2), t, k3); }
__device__ __forceinline__ u64 add(u64 a, u64 b) {
  u64 k1, k2; u32 lo = addco(lo32(a), lo32(b), k1); u32 hi = addc(hi32(a), hi32(b), k1, k2);
  return repay_carry(pack(lo, hi), k2);
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {   // a - b - borrow * (2^32 - 1), twice
  u64 k1, k2, k3, k4, k5, k6;
  u32 lo = subco(lo32(a), lo32(b), k1); u32 hi = subb(hi32(a), hi32(b), k1, k2);
  u32 l2 = subco(lo, selm1(k2), k3); u32 h2 = subb0(hi, k3, k4);
  u32 l3 = subco(l2, selm1(k4), k5); u32 h3 = subb0(h2, k5, k6);
  return pack(l3, h3);
}
// (hi : lo) mod p for hi < 2^32:  lo + hi * (2^32 - 1)
__device__ __forceinline__ u64 red32(u64 lo, u32 hi) { u64 k; u64 t = madm1(hi, lo, k); u64 k2; return madm1(sel01(k), t, k2); }
// (hi : lo) mod p, weak:  lo - hi_hi + hi_lo * (2^32 - 1); the borrow comes off the multiplier (gl_mul_weak, gl.cuh)
__device__ __forceinline__ u64 red(u64 lo, u64 hi) {
  u64 k1, k2, k3, k4, k5, k6;
  const u32 l = subco(lo32(lo), hi32(hi), k1), h = subb0(hi32(lo), k1, k2);
  const u32 t = subb0(lo32(hi), k2, k3);
  const u64 r = madm1(t, pack(l, h), k4);
  const u32 rl = addc(lo32(r), selm1(k4), k3, k5), rh = addc(hi32(r), 0u, k5, k6);
  return pack(rl, rh);
}
template <int E> __device__ __forceinline__ u64 mul_pow2(u64 x) {  // x * 2^E, 0 <= E < 96
  if constexpr (E == 0) return x;
  else if constexpr (E < 32) return red32(x << E, hi32(x) >> (32 - E));
  else if constexpr (E == 32) return red32(pack(0, lo32(x)), hi32(x));
  else if constexpr (E < 64) return red(pack(0, lo32(x) << (E - 32)), x >> (64 - E));
  else return mul_pow2<E - 48>(mul_pow2<48>(x));
}
__device__ __forceinline__ u64 mul(u64 a, u64 b) { return gl_mul_weak(a, b); }
__device__ __forceinline__ u64 canon(u64 x) { return gp::canon(x); }
#else
__device__ __forceinline__ u64 add(u64 a, u64 b) {
  u64 s = a + b;
  u64 c = s < a ? GLEPS : 0;
  s += c;
  if (s < c) s += GLEPS;
  return s;
}
__device__ __forceinline__ u64 sub(u64 a, u64 b) {
  u64 d = a - b;
  u64 c = a < b ? GLEPS : 0;
  u64 r = d - c;
  if (d < c) r -= GLEPS;
  return r;
}
__device__ __forceinline__ u64 red(u64 lo, u64 hi) {  // (hi:lo) mod p, weak
  u32 hi_hi = (u32)(hi >> 32), hi_lo = (u32)hi;
  u64 t0 = lo - hi_hi;
  if (lo < hi_hi) t0 -= GLEPS;
  u64 t1 = ((u64)hi_lo << 32) - hi_lo;
  u64 r = t0 + t1;
  if (r < t1) r += GLEPS;
  return r;
}
template <int E> __device__ __forceinline__ u64 mul_pow2(u64 x) {  // x * 2^E, 0 <= E < 96
  if constexpr (E == 0) return x;
  else if constexpr (E < 64) return red(x << E, x >> (64 - E));
  else return mul_pow2<E - 48>(mul_pow2<48>(x));
}
__device__ __forceinline__ u64 mul(u64 a, u64 b) { return red(a * b, __umul64hi(a, b)); }
__device__ __forceinline__ u64 canon(u64 x) { return x >= GLP ? x - GLP : x; }
#endif

#define NW_BF(i, j, E) { u64 a_ = x[i], b_ = x[j]; x[i] = add(a_, b_); x[j] = mul_pow2<E>(sub(a_, b_)); }
// 16-point DIF DFT with root rho = 2^12; x[p] <- DFT_rho[bitrev4(p)]
__device__ __forceinline__ void dft16_rho(u64* x) {
  NW_BF(0, 8, 0) NW_BF(1, 9, 12) NW_BF(2, 10, 24) NW_BF(3, 11, 36) NW_BF(4, 12, 48) NW_BF(5, 13, 60) NW_BF(6, 14, 72) NW_BF(7, 15, 84)
  NW_BF(0, 4, 0) NW_BF(1, 5, 24) NW_BF(2, 6, 48) NW_BF(3, 7, 72) NW_BF(8, 12, 0) NW_BF(9, 13, 24) NW_BF(10, 14, 48) NW_BF(11, 15, 72)
  NW_BF(0, 2, 0) NW_BF(1, 3, 48) NW_BF(4, 6, 0) NW_BF(5, 7, 48) NW_BF(8, 10, 0) NW_BF(9, 11, 48) NW_BF(12, 14, 0) NW_BF(13, 15, 48)
  NW_BF(0, 1, 0) NW_BF(2, 3, 0) NW_BF(4, 5, 0) NW_BF(6, 7, 0) NW_BF(8, 9, 0) NW_BF(10, 11, 0) NW_BF(12, 13, 0) NW_BF(14, 15, 0)
}
#undef NW_BF
}  // namespace nw

// LOG_B = 0: R = 256, LOG_B = 1: R = 512.  Tile width T = 16, 256 threads, LDS = (R/16)*272*8 bytes.
// `kperm` = 5 (forward) or 11 (inverse): index of DFT_omega given the index under rho.
#ifdef SBN_NTT_KERNELS_HERE
template <int LOG_B>
__global__ __launch_bounds__(256) void ntt_fast_pass_kernel(NttPassParams p, u32 kperm) {
  extern __shared__ u64 lds[];
  constexpr u32 NB = 1u << LOG_B;       // 256-point blocks per tile column
  constexpr u32 R = 256u << LOG_B;
  const size_t col = p.xcd_order ? blockIdx.x : blockIdx.y;
  const size_t t0 = (size_t)(p.xcd_order ? blockIdx.y : blockIdx.x) << 4;
  const u32 half = p.split ? blockIdx.z : 0u;            // split: which parity of the 1,024-point pass's outputs
  const u64* __restrict__ pre = half ? p.pre2 : p.pre;
  const u64* in = p.in + col * p.in_col_stride;
  u64* out = p.out + col * p.out_col_stride;
  // work item = (g, t): g = residue of r mod 16 within a 256-block, t = tile column.  For pass B (r contiguous in
  // memory) lanes run over g first so that each group of 16 lanes reads 128 contiguous bytes.
  u32 g, t;
  if (p.r_fast_load) { g = threadIdx.x & 15; t = threadIdx.x >> 4; } else { t = threadIdx.x & 15; g = threadIdx.x >> 4; }
  const size_t tg = t0 + t;
  // LDS exchange layout, chosen per lane order so that both the round-1 writes and the round-2 reads are
  // bank-conflict free (8-byte slots, 32 per LDS row): lanes (t fastest): [k1][g][t], row stride 272;
  // lanes (g fastest): [k1][t][g], row stride 257.
  const u32 S1 = p.r_fast_load ? 257u : 272u;
  const u32 lane_off = p.r_fast_load ? (t * 16) : t;   // + g*16 or + g
  const u32 g_mul = p.r_fast_load ? 1u : 16u;
  u64 x[NB][16];
  // ---- load (+ coset pre-scale), natural order r = g + 16 q (+ 256 b)
#pragma unroll
  for (u32 b = 0; b < NB; b++)
#pragma unroll
    for (u32 q = 0; q < 16; q++) {
      u32 r = g + 16 * q + 256 * b;
      size_t gi = (size_t)r * p.in_sr + tg * p.in_st;
      u64 v = 0;
      if (gi < p.n_in) {
        v = in[gi];
        if (pre) v = nw::mul(v, pre[gi]);
      }
      x[b][q] = v;
    }
  // ---- leading radix-2 DIF stage for R = 512: (u, v) = (a + b, (a - b) w_512^r); block 0 -> even, block 1 -> odd outputs
  if (LOG_B == 1) {
    // the zero-padded LDE (n_in = half the rows): the upper block is all zeros, so (u, v) = (a, a w) without the add / sub
    const bool upper_zero = p.in_st == 1 && (size_t)256 * p.in_sr >= p.n_in;
#pragma unroll
    for (u32 q = 0; q < 16; q++) {
      u32 r = g + 16 * q;
      u64 a = x[0][q], bb = x[NB - 1][q];
      u64 w = tw_full(p.tw, p.tw_log, r, 9);
      if (upper_zero) { x[NB - 1][q] = nw::mul(a, w); }
      else { x[0][q] = nw::add(a, bb); x[NB - 1][q] = nw::mul(nw::sub(a, bb), w); }
    }
  }
  // ---- round 1: DFT-16 over q, twiddle by w_256^(g k1), exchange through LDS
#pragma unroll
  for (u32 b = 0; b < NB; b++) {
    nw::dft16_rho(x[b]);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      u32 m = __brev(pidx) >> 28;             // index under rho
      u32 k1 = (kperm * m) & 15;              // index under omega_16 (or its inverse)
      u64 v = x[b][pidx];
      if (g && k1) v = nw::mul(v, tw_full(p.tw, p.tw_log, (u32)(g * k1), 8));
      lds[(b * 16 + k1) * S1 + lane_off + g * g_mul] = v;
    }
  }
  __syncthreads();
  // ---- round 2: this thread now owns k1 = its former g index: read Z[g'][k1] for g' = 0..15
  const u32 k1 = g;
#pragma unroll
  for (u32 b = 0; b < NB; b++) {
#pragma unroll
    for (u32 gp = 0; gp < 16; gp++) x[b][gp] = lds[(b * 16 + k1) * S1 + lane_off + gp * g_mul];
    nw::dft16_rho(x[b]);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      u32 m = __brev(pidx) >> 28;
      u32 k2 = (kperm * m) & 15;
      u32 k256 = k1 + 16 * k2;                       // frequency inside the 256-point block
      u32 k = LOG_B == 0 ? k256 : (k256 * NB + b);   // DIF: block b holds outputs k = 2 k256 + b
      if (LOG_B == 1) k = (k << p.split) + half;      // split: this 512-point pass yields the outputs 2 k + half of the 1,024-point one
      u64 v = x[b][pidx];
      if (p.twiddle) v = nw::mul(v, tw_full(p.tw, p.tw_log, (u32)(k * tg), p.log_n));
      if (p.scale != 1) v = nw::mul(v, p.scale);
      size_t go = (size_t)k * p.out_sr + tg * p.out_st;
      if (p.post) v = nw::mul(v, p.post[go]);
      out[go] = nw::canon(v);
    }
  }
  (void)R;
}
#else
template <int LOG_B> __global__ void ntt_fast_pass_kernel(NttPassParams p, u32 kperm);   // <0>, <1> instantiated in ntt.hip
#endif

// -------------------------------------------------------------------------------------------------------------
// K1'' the second pass of the inverse transform FUSED with the first pass of the coset LDE (n = 2^16, 2^17 rows).
// With n = n1 * n2 (n2 = 256) the inverse transform's pass B leaves a lane with the sixteen coefficients
// k = k1 + 16 k2 (k2 = 0..15) of tile column tg, i.e. coefficient indices k * n1 + tg -- and the LDE's first pass (512-point
// over rows r of which only r < 256 are non-zero, stride n1 between rows) wants exactly r = g + 16 q of tile column tg in
// one lane: the same lane with g = k1, q = k2.  So the coefficients are stored (they are needed by the openings and FRI)
// and go on, in registers, through the coset scale, the trivial radix-2 stage and the two DFT-16 rounds of the LDE pass.
// Saves one launch and one read of the coefficients per column chunk, no arithmetic.  The second LDS exchange also
// TRANSPOSES the lane order (g fastest -> t fastest), so the final stores are 128-byte runs like the unfused pass's.
// -------------------------------------------------------------------------------------------------------------
struct NttFusedParams {
  const u64* in; size_t in_col_stride;      // pass-A output of the inverse transform: element (r, tg) at r + tg * 256
  u64* coef; size_t coef_col_stride;        // coefficients, natural order: (k, tg) at k * n1 + tg
  u64* out; size_t out_col_stride;          // pass-A output of the LDE: (k, tg) at k * n1 + tg, k < 512
  size_t n1;                                // 2^ceil(log n / 2) = tile columns of both passes
  u32 log_n;                                // 16 or 17
  const u64* tw_i; const u64* tw_f; u32 tw_log;   // inverse / forward root tables of the LDE size
  const u64* pre;                           // coset scale 7^i
  u64 scale;                                // 1 / n
  const u64* pre2;                          // 2^18-row form only: 7^i w_1024^(row of i), NttPassParams::pre2
};
#ifdef SBN_NTT_KERNELS_HERE
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void ntt_fused_inv_b_lde_a_kernel(NttFusedParams p) {   // 128 VGPRs: two of its waves fit beside the sponge's two on a SIMD
  extern __shared__ u64 lds[];
  const size_t col = blockIdx.x;                       // (columns, tiles) grid order, as the other passes
  const size_t t0 = (size_t)blockIdx.y << 4;
  const u64* in = p.in + col * p.in_col_stride;
  u64* coef = p.coef + col * p.coef_col_stride;
  u64* out = p.out + col * p.out_col_stride;
  const u32 g = threadIdx.x & 15, t = threadIdx.x >> 4;   // part 1: lanes over g first (the input rows are contiguous)
  const size_t tg = t0 + t;
  u64 x[2][16];
  // ---- part 1: pass B of the inverse transform (256-point, omega^-1: index permutation 11)
#pragma unroll
  for (u32 q = 0; q < 16; q++) x[0][q] = in[(size_t)(g + 16 * q) + tg * 256];
  nw::dft16_rho(x[0]);
#pragma unroll
  for (u32 pidx = 0; pidx < 16; pidx++) {
    const u32 m = __brev(pidx) >> 28, k1 = (11u * m) & 15;
    u64 v = x[0][pidx];
    if (g && k1) v = nw::mul(v, tw_full(p.tw_i, p.tw_log, (u32)(g * k1), 8));
    lds[k1 * 257 + t * 16 + g] = v;
  }
  __syncthreads();
#pragma unroll
  for (u32 gp = 0; gp < 16; gp++) x[0][gp] = lds[g * 257 + t * 16 + gp];   // this lane now owns k1 = g
  nw::dft16_rho(x[0]);
  // coefficient k = g + 16 k2, k2 = 11 * brev4(pidx): store it, and keep it as row r = g + 16 q (q = k2) of the LDE pass
  u64 a[16];
#pragma unroll
  for (u32 pidx = 0; pidx < 16; pidx++) {
    const u32 m = __brev(pidx) >> 28, k2 = (11u * m) & 15;
    const u32 k = g + 16 * k2;
    const u64 v = nw::mul(x[0][pidx], p.scale);
    coef[(size_t)k * p.n1 + tg] = nw::canon(v);
    a[k2] = v;
  }
  __syncthreads();                                    // every lane has read its part-1 values: the LDS is free again
  // ---- part 2: pass A of the coset LDE (512-point, rows 256.. zero, omega: index permutation 5)
#pragma unroll
  for (u32 q = 0; q < 16; q++) {
    const u32 r = g + 16 * q;
    const u64 v = nw::mul(a[q], p.pre[(size_t)r * p.n1 + tg]);
    x[0][q] = v;
    x[1][q] = nw::mul(v, tw_full(p.tw_f, p.tw_log, r, 9));     // (u, v) = (a, a w_512^r): the upper half is zero
  }
#pragma unroll
  for (u32 b = 0; b < 2; b++) {
    nw::dft16_rho(x[b]);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      const u32 m = __brev(pidx) >> 28, k1 = (5u * m) & 15;
      u64 v = x[b][pidx];
      if (g && k1) v = nw::mul(v, tw_full(p.tw_f, p.tw_log, (u32)(g * k1), 8));
      lds[(b * 16 + k1) * 272 + g * 17 + t] = v;       // [k1][g][t], g rows padded to 17: both sides (nearly) conflict-free
    }
  }
  __syncthreads();
  // round 2 with the lanes re-dealt t fastest: lane (k1, t2) -- the stores below are 16 consecutive words per row
  const u32 t2 = threadIdx.x & 15, k1o = threadIdx.x >> 4;
  const size_t tg2 = t0 + t2;
#pragma unroll
  for (u32 b = 0; b < 2; b++) {
#pragma unroll
    for (u32 gp = 0; gp < 16; gp++) x[b][gp] = lds[(b * 16 + k1o) * 272 + gp * 17 + t2];
    nw::dft16_rho(x[b]);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      const u32 m = __brev(pidx) >> 28, k2 = (5u * m) & 15;
      const u32 k = (k1o + 16 * k2) * 2 + b;           // DIF: block b holds the outputs 2 k256 + b
      const u64 v = nw::mul(x[b][pidx], tw_full(p.tw_f, p.tw_log, (u32)(k * tg2), p.log_n + 1));
      out[(size_t)k * p.n1 + tg2] = nw::canon(v);
    }
  }
}
#else
__global__ void ntt_fused_inv_b_lde_a_kernel(NttFusedParams p);
#endif

// -------------------------------------------------------------------------------------------------------------
// K1-512: the same fusion for 2^18-row tables (n = 512 x 512; the LDE is 2^19 = 1,024 x 512 points, its first pass two
// 512-point transforms, NttPassParams::split).  The inverse transform's pass B is a 512-point DIF pass: a leading radix-2 stage,
// then two 256-point blocks, so a lane ends with the coefficients k = 2 (g + 16 k2) + b (k2 = 0..15, b = 0, 1) of its tile
// column.  The LDE's pass wants the rows r = k of the same tile column; it is therefore run decimation in TIME over the
// parity of r: E = DFT_256(rows 2 r'), O = DFT_256(rows 2 r' + 1) with r' = g + 16 q -- exactly what the lane holds at
// (b, q = k2) -- and X[j] = E[j] + w_512^j O[j], X[j + 256] = E[j] - w_512^j O[j] at the end, in the lane that owns j in
// both blocks.  Output half h (rows 2 X + h of the 1,024-point pass) scales the rows by pre (h = 0) or pre2 (h = 1) first.
// Half 0 takes the coefficients from registers; half 1 reads the lane's own 32 coefficients back (written a moment ago: L2)
// rather than holding them, which would cost 64 more VGPRs and the place beside the sponge waves.  Saves two launches per
// chunk and one read of the coefficients; same arithmetic as the separate passes.  Capped at 128 VGPRs (two waves beside the
// sponge's two; the compiler spills 14 dwords of lane indices at the start): measured against 143 VGPRs without spills,
// Fq12ExpStark(512) 0.605 against 0.608 s and the sponge launches beside it 8 % faster (profiles/r4_fused512.txt).  The compiler
// fences and the opaque lane indices below keep it from hoisting ~100 table / output addresses out of the phases they belong to
// (without them: 256 VGPRs + 10 AGPRs, or ~190 spilled dwords at 128).
// -------------------------------------------------------------------------------------------------------------
#ifdef SBN_NTT_KERNELS_HERE
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void ntt_fused512_inv_b_lde_a_kernel(NttFusedParams p) {
  extern __shared__ u64 lds[];
  const size_t col = blockIdx.x;
  const size_t t0 = (size_t)blockIdx.y << 4;
  const u64* in = p.in + col * p.in_col_stride;
  u64* coef = p.coef + col * p.coef_col_stride;
  u64* out = p.out + col * p.out_col_stride;
  const u32 g = threadIdx.x & 15, t = threadIdx.x >> 4;   // part 1: lanes over g first (the input rows are contiguous)
  const size_t tg = t0 + t;
  u64 x[2][16];
  // ---- part 1: pass B of the inverse transform (512-point DIF, omega^-1: index permutation 11)
#pragma unroll
  for (u32 q0 = 0; q0 < 16; q0 += 8) {
#pragma unroll
    for (u32 q = q0; q < q0 + 8; q++) {
      const u64 lo = in[(size_t)(g + 16 * q) + tg * 512], hi = in[(size_t)(g + 16 * q + 256) + tg * 512];
      x[0][q] = nw::add(lo, hi);
      x[1][q] = nw::mul(nw::sub(lo, hi), tw_full(p.tw_i, p.tw_log, g + 16 * q, 9));
    }
    asm volatile("" ::: "memory");
  }
#pragma unroll
  for (u32 b = 0; b < 2; b++) {
    nw::dft16_rho(x[b]);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      const u32 m = __brev(pidx) >> 28, k1 = (11u * m) & 15;
      u64 v = x[b][pidx];
      if (g && k1) v = nw::mul(v, tw_full(p.tw_i, p.tw_log, (u32)(g * k1), 8));
      lds[(b * 16 + k1) * 257 + t * 16 + g] = v;
    }
  }
  __syncthreads();
  // coefficient k = 2 (g + 16 k2) + b: stored, and kept (weak form) as row k of the LDE pass
#pragma unroll
  for (u32 b = 0; b < 2; b++) {
    u64 z[16];
#pragma unroll
    for (u32 gp = 0; gp < 16; gp++) z[gp] = lds[(b * 16 + g) * 257 + t * 16 + gp];
    nw::dft16_rho(z);
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      const u32 m = __brev(pidx) >> 28, k2 = (11u * m) & 15;
      const u64 v = nw::mul(z[pidx], p.scale);
      coef[(size_t)((g + 16 * k2) * 2 + b) * 512 + tg] = nw::canon(v);
      x[b][k2] = v;
    }
    asm volatile("" ::: "memory");                     // (block 1's exchange reads stay behind block 0's stores: registers)
  }
  __syncthreads();
  // ---- part 2: the 1,024-point first pass of the coset LDE, output half h = 0, 1 (omega: index permutation 5)
#pragma unroll 1
  for (u32 h = 0; h < 2; h++) {
    u32 t2 = threadIdx.x & 15, k1o = threadIdx.x >> 4;   // lanes of the last round: t fastest (128-byte stores)
    u32 gl = g, tl = t;
    asm volatile("" : "+v"(t2), "+v"(k1o), "+v"(gl), "+v"(tl));   // opaque: ~100 table / output addresses are not hoisted out of the loop (and spilled)
    const size_t tg2 = t0 + t2, tgl = t0 + tl;
    // rows scaled by the coset table of this half, eight at a time (compiler fences: 32 hoisted table loads and their
    // addresses would cost the occupancy); half 1 reads the lane's own coefficients back (L2) instead of holding 32 values
    const u64* __restrict__ pre = h ? p.pre2 : p.pre;
#pragma unroll
    for (u32 b = 0; b < 2; b++)
#pragma unroll
      for (u32 q0 = 0; q0 < 16; q0 += 8) {
#pragma unroll
        for (u32 q = q0; q < q0 + 8; q++) {
          const size_t ci = (size_t)(2 * (gl + 16 * q) + b) * 512 + tgl;
          u64 c = x[b][q];
          if (h) c = coef[ci];
          x[b][q] = nw::mul(c, pre[ci]);
        }
        asm volatile("" ::: "memory");
      }
#pragma unroll
    for (u32 b = 0; b < 2; b++) {
      nw::dft16_rho(x[b]);
#pragma unroll
      for (u32 pidx = 0; pidx < 16; pidx++) {
        const u32 m = __brev(pidx) >> 28, k1 = (5u * m) & 15;
        u64 v = x[b][pidx];
        if (gl && k1) v = nw::mul(v, tw_full(p.tw_f, p.tw_log, (u32)(gl * k1), 8));
        lds[(b * 16 + k1) * 272 + gl * 17 + tl] = v;
      }
    }
    __syncthreads();
#pragma unroll
    for (u32 b = 0; b < 2; b++) {
#pragma unroll
      for (u32 gp = 0; gp < 16; gp++) x[b][gp] = lds[(b * 16 + k1o) * 272 + gp * 17 + t2];
      nw::dft16_rho(x[b]);
    }
#pragma unroll
    for (u32 pidx = 0; pidx < 16; pidx++) {
      const u32 m = __brev(pidx) >> 28, k2 = (5u * m) & 15;
      const u32 j = k1o + 16 * k2;                       // frequency inside the 256-point blocks
      const u64 e = x[0][pidx], o = nw::mul(x[1][pidx], tw_full(p.tw_f, p.tw_log, j, 9));
      const u32 ka = 2 * j + h, kb = 2 * (j + 256) + h;  // rows of the 1,024-point pass
      out[(size_t)ka * 512 + tg2] = nw::canon(nw::mul(nw::add(e, o), tw_full(p.tw_f, p.tw_log, (u32)(ka * tg2), 19)));
      out[(size_t)kb * 512 + tg2] = nw::canon(nw::mul(nw::sub(e, o), tw_full(p.tw_f, p.tw_log, (u32)(kb * tg2), 19)));
    }
    __syncthreads();                                    // the exchange buffer is rewritten by the next half
  }
}
#else
__global__ void ntt_fused512_inv_b_lde_a_kernel(NttFusedParams p);
#endif

