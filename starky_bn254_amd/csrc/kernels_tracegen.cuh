// Device witness generation for G1ExpStark and G2ExpStark (E = 1, 2): replaces generate_trace / generate_public_inputs
// (src/curves/g1/exp.rs:255-327, src/curves/g2/exp.rs:271-342) with kernels that fill the column-major trace directly in HBM, so the 0.9 GB
// trace never crosses PCIe.  The host generator (tracegen.hip) is the bit-exact counterpart; both share bn254w.cuh.
//
// The reference's per-instance loop is sequential in two ways: a = 2^t x is a chain of doublings and
// b += a is a chain of additions, and every affine step needs a field inversion.  Here
//   * both chains of an instance are walked in Jacobian coordinates (no inversion): bn254w.cuh g1_chains, run on host
//     threads by default (512 strictly sequential point operations per instance are 0.1% of the arithmetic and
//     latency-bound on a lane) or by chain_kernel, one lane per instance;
//   * affine_lambda_kernel then inverts all Z's and all slope denominators at once (the inversions of different
//     rows are independent once the chain is known, and the denominators have closed forms in Jacobian coordinates): a lane owns
//     the eight values of two rows and inverts their product (Montgomery's trick; Fermat, a^(p-2), for the one inversion left);
//   * gadget_witness_kernel computes the limb columns and the modular-gadget witnesses, one lane per (row, gadget);
//   * flags / pulse columns are closed forms of the row index;
//   * range_check_kernel: one workgroup per range-checked column, histogram and prefix counts in LDS, and the
//     reference's sorted-merge with its LIFO pool of unused table values (src/utils/lookup.rs:60-111) restated as
//     independent searches over the prefix counts.
#pragma once
#include "bn254w.cuh"

namespace tg {
using namespace bnw;


// a^(p-2) with 4-bit fixed windows (252 squarings + <= 64 + 14 products).
__device__ __noinline__ Fq finv_fermat(const Fq& a) {
  constexpr u64 PL[4] = BNW_PL;
  const u64 e[4] = {PL[0] - 2, PL[1], PL[2], PL[3]};
  Fq tab[16];
  tab[0] = fq_one(); tab[1] = a;
  for (int i = 2; i < 16; i++) tab[i] = mmul(tab[i - 1], a);
  Fq r = tab[(e[3] >> 60) & 15];
  for (int w = 62; w >= 0; w--) {
    r = mmul(r, r); r = mmul(r, r); r = mmul(r, r); r = mmul(r, r);
    const unsigned nib = (unsigned)(e[w >> 4] >> (4 * (w & 15))) & 15;
    if (nib) r = mmul(r, tab[nib]);
  }
  return r;
}


// Inversions are BATCHED inside a lane (Montgomery's trick): a lane multiplies up the TG_INV_BATCH values it owns, inverts the
// product with one Fermat chain (~330 multiplies) and walks back with two multiplies per value, so an inversion costs ~45
// multiplies instead of ~330.  A coordinate in Fq2 is inverted through its norm a^2 + b^2 (an Fq value): 1/(a + bi) = (a - bi) / norm.
static constexpr int TG_INV_BATCH = 8;
__device__ __forceinline__ Fq cnorm(const Co<1>& a) { return a.c[0]; }
__device__ __forceinline__ Fq cnorm(const Co<2>& a) { return fadd(mmul(a.c[0], a.c[0]), mmul(a.c[1], a.c[1])); }
__device__ __forceinline__ Co<1> cinv_from_norm(const Co<1>&, const Fq& ni) { Co<1> r; r.c[0] = ni; return r; }
__device__ __forceinline__ Co<2> cinv_from_norm(const Co<2>& a, const Fq& ni) {
  Co<2> r; r.c[0] = mmul(a.c[0], ni); r.c[1] = fsub(Fq{{0, 0, 0, 0}}, mmul(a.c[1], ni));
  return r;
}
// nrm[j] (all non-zero) -> nrm[j]^-1, j < TG_INV_BATCH
__device__ __forceinline__ void batch_inverse(Fq (&nrm)[TG_INV_BATCH]) {
  Fq pre[TG_INV_BATCH];
  pre[0] = nrm[0];
#pragma unroll
  for (int j = 1; j < TG_INV_BATCH; j++) pre[j] = mmul(pre[j - 1], nrm[j]);
  Fq inv = finv_fermat(pre[TG_INV_BATCH - 1]);
#pragma unroll
  for (int j = TG_INV_BATCH - 1; j > 0; j--) {
    const Fq ni = mmul(inv, pre[j - 1]);
    inv = mmul(inv, nrm[j]);
    nrm[j] = ni;
  }
  nrm[0] = inv;
}

// One lane per instance: exp_chains (bn254w.cuh).  512 strictly sequential point operations per lane: the host
// threads do this faster (prover.hip picks), the kernel keeps the path host-free when asked (SBN_TRACEGEN_DEVICE_CHAIN=1).
template <int E>
__global__ void chain_kernel(const uint32_t* __restrict__ ios, size_t K, u64* __restrict__ ja, u64* __restrict__ jb, int* __restrict__ err) {
  const size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (k >= K) return;
  const int bad = exp_chains<E>(ios + 8 * (4 * E + 1) * k, k, ja, jb);
  if (bad) atomicOr(err, bad);
}

// ---- the curve chains with ONE WAVE per instance (round 3) --------------------------------------------------------------------
// exp_chains is 512 dependent point operations per instance: 13 ms on one lane, 2.2 ms on the host pool.  But a Jacobian doubling
// and an addition are ~25 Fq (4 x that: Fq2) products of which up to 8 (32) are independent, and the doubling of a and the addition
// b += a of one step do not depend on each other.  The host lays one step out as LEVELS of independent micro-operations over Fq
// slots (ChainProgram in prover.hip: every product, sum, difference of jac_double / jac_add, Fq2 products expanded into their four
// Fq products; level = longest dependency path), and a wave executes a level with one micro-operation per lane, operands and
// results in LDS.  Field arithmetic is exact, so the stored Jacobian coordinates are word for word those of exp_chains.
// Micro-operation word: kind | dst << 8 | a << 16 | b << 24 (slot numbers < 256).
enum { CP_NOP = 0, CP_MUL = 1, CP_ADD = 2, CP_SUB = 3, CP_COPY = 4, CP_CHK1 = 5, CP_CHK2 = 6, CP_LANES = 64, CP_MAX_SLOTS = 256 };
struct ChainProgDev {
  const uint32_t* ops[2];     // [bit]: [levels][CP_LANES] micro-operations
  int levels[2];
  unsigned char in_slot[8];   // slots of x.x x.y offset.x offset.y (E each, component-major as in the io words)
  unsigned char one_slot[2];  // first components of a.Z and b.Z (set to 1; the other components to 0)
  unsigned char zero_slot[2]; // second components of a.Z and b.Z (E = 2; 255 = none)
  unsigned char coord[12];    // slots of a.X a.Y a.Z b.X b.Y b.Z, E each
};
template <int E>
__global__ void __launch_bounds__(CP_LANES) chain_coop_kernel(const uint32_t* __restrict__ ios, size_t K, u64* __restrict__ ja, u64* __restrict__ jb,
                                                              int* __restrict__ err, ChainProgDev cp) {
  __shared__ Fq V[CP_MAX_SLOTS];
  __shared__ uint32_t prog[2][24 * CP_LANES];     // both programs (<= 24 levels each), read 256 times: kept in LDS
  const size_t k = blockIdx.x;
  const int lane = threadIdx.x;
  for (int b = 0; b < 2; b++) for (int lv = 0; lv < cp.levels[b]; lv++) prog[b][lv * CP_LANES + lane] = cp.ops[b][lv * CP_LANES + lane];
  const uint32_t* io = ios + 8 * (4 * E + 1) * k;
  if (lane < 4 * E) { u64 t4[4]; u32x8_to_u64x4(io + 8 * lane, t4); V[cp.in_slot[lane]] = to_m(t4); }
  if (lane >= 32 && lane < 34) V[cp.one_slot[lane - 32]] = fq_one();
  if (E == 2 && lane >= 40 && lane < 42) V[cp.zero_slot[lane - 40]] = Fq{{0, 0, 0, 0}};
  __syncthreads();
  const uint32_t* e = io + 32 * E;
  for (int t = 0;; t++) {
    if (lane < 6 * E) {                       // the Jacobian coordinates of step t, Montgomery form (affine_lambda_kernel reads them)
      const int which = lane / (3 * E), cc = (lane % (3 * E)) / E, q = lane % E;
      const Fq v = V[cp.coord[lane]];
      u64* dst = (which ? jb : ja) + jac_at<E>(k, t, cc) + 4 * q;
      for (int i = 0; i < 4; i++) dst[i] = v.l[i];
    }
    if (t == 256) break;
    const int bit = (e[t >> 5] >> (t & 31)) & 1;
    const uint32_t* ops = prog[bit];
    const int nl = cp.levels[bit];
    for (int lv = 0; lv < nl; lv++) {
      const uint32_t op = ops[lv * CP_LANES + lane];
      const int kind = op & 255, d = (op >> 8) & 255, a = (op >> 16) & 255, b = op >> 24;
      if (kind == CP_MUL) V[d] = mmul(V[a], V[b]);
      else if (kind == CP_ADD) V[d] = fadd(V[a], V[b]);
      else if (kind == CP_SUB) V[d] = fsub(V[a], V[b]);
      else if (kind == CP_COPY) V[d] = V[a];
      else if (kind == CP_CHK1) { if (fzero(V[a])) atomicOr(err, TG_ERR_DEGENERATE); }
      else if (kind == CP_CHK2) { if (fzero(V[a]) && fzero(V[b])) atomicOr(err, TG_ERR_DEGENERATE); }
      __syncthreads();
    }
  }
}

// Row r of instance k: a = A[r>>1]; even rows: b = B[r>>1], add if bit; odd rows: b = B[(r>>1)+1], double.
template <int E>
struct LambdaRow {
  size_t k; int t, tb, op; bool dbl;
  __device__ __forceinline__ LambdaRow(const uint32_t* __restrict__ ios, size_t row) {
    k = row >> 9; const int r = (int)(row & 511); t = r >> 1; dbl = r & 1;
    const bool bit = (ios[8 * (4 * E + 1) * k + 32 * E + (t >> 5)] >> (t & 31)) & 1;
    op = dbl ? 2 : (bit ? 1 : 0);
    tb = dbl ? t + 1 : t;
  }
};
// One lane per TG_ROWS rows (row = L + j NL), straight from the JACOBIAN chains: the standard-form values ax ay bx by lam nx ny of
// the row (sv[((v*E + q)*4 + limb) * n + row]) and its operation (0 none, 1 add, 2 double); also the instance output B[256] (u32
// limbs) for the public inputs.  Rounds 2-3 ran two kernels, each waiting for one Fermat chain (~350 dependent products, 0.44 +
// 0.49 ms): affine_kernel inverted every Z, lambda_kernel then the slope denominators.  The denominators have closed forms in
// Jacobian coordinates -- x2 - x1 = (X2 Z1^2 - X1 Z2^2) / (Z1^2 Z2^2), 2 y1 = 2 Y1 / Z1^3 -- so ONE batch per lane inverts Z1, Z2,
// Z_next and D = X2 Z1^2 - X1 Z2^2 (or 2 Y1) of its rows together (Montgomery's trick, one Fermat chain), and
// lambda = (y2 - y1) Z1^2 Z2^2 / D  (or 3 x1^2 Z1^3 / (2 Y1)).  Field elements are exact, so every word written equals the two-kernel
// path's.  D = 0 iff the affine denominator is 0 (the Z's are non-zero): the same instances are refused.
static constexpr int TG_ROWS = TG_INV_BATCH / 4;
template <int E>
__global__ void affine_lambda_kernel(const uint32_t* __restrict__ ios, size_t K, const u64* __restrict__ ja, const u64* __restrict__ jb, size_t n,
                                     u64* __restrict__ sv, unsigned char* __restrict__ row_op, u64* __restrict__ pi_out, int* __restrict__ err) {
  const size_t L = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  const size_t NL = (n + TG_ROWS - 1) / TG_ROWS;
  if (L >= NL) return;
  Fq nrm[TG_INV_BATCH];
  for (int j = 0; j < TG_ROWS; j++) {
    const size_t row = L + (size_t)j * NL;
    for (int i = 0; i < 4; i++) nrm[4 * j + i] = fq_one();
    if (row >= n) continue;
    const LambdaRow<E> R(ios, row);
    const Co<E> Z1 = ldc<E>(ja + jac_at<E>(R.k, R.t, 2)), Z2 = ldc<E>(jb + jac_at<E>(R.k, R.tb, 2));
    if (czero<E>(Z1) || czero<E>(Z2)) { atomicOr(err, TG_ERR_DEGENERATE); continue; }   // (a zero would poison the batch; the proof is refused anyway)
    nrm[4 * j] = cnorm(Z1); nrm[4 * j + 1] = cnorm(Z2);
    if (!R.op) continue;
    const Co<E> Zn = ldc<E>((R.dbl ? ja : jb) + jac_at<E>(R.k, R.t + 1, 2));
    Co<E> D;
    if (R.dbl) { const Co<E> Y1 = ldc<E>(ja + jac_at<E>(R.k, R.t, 1)); D = cadd(Y1, Y1); }
    else D = csub(cmul(ldc<E>(jb + jac_at<E>(R.k, R.tb, 0)), cmul(Z1, Z1)), cmul(ldc<E>(ja + jac_at<E>(R.k, R.t, 0)), cmul(Z2, Z2)));
    if (czero<E>(Zn) || czero<E>(D)) { atomicOr(err, TG_ERR_DEGENERATE); continue; }
    nrm[4 * j + 2] = cnorm(Zn); nrm[4 * j + 3] = cnorm(D);
  }
  batch_inverse(nrm);
  for (int j = 0; j < TG_ROWS; j++) {
    const size_t row = L + (size_t)j * NL;
    if (row >= n) continue;
    const LambdaRow<E> R(ios, row);
    const size_t k = R.k; const int t = R.t, op = R.op; const bool dbl = R.dbl;
    const Co<E> X1 = ldc<E>(ja + jac_at<E>(k, t, 0)), Y1 = ldc<E>(ja + jac_at<E>(k, t, 1)), Z1 = ldc<E>(ja + jac_at<E>(k, t, 2));
    const Co<E> X2 = ldc<E>(jb + jac_at<E>(k, R.tb, 0)), Y2 = ldc<E>(jb + jac_at<E>(k, R.tb, 1)), Z2 = ldc<E>(jb + jac_at<E>(k, R.tb, 2));
    Co<E> v[7];
    const Co<E> zi1 = cinv_from_norm(Z1, nrm[4 * j]), zi1s = cmul(zi1, zi1);
    const Co<E> zi2 = cinv_from_norm(Z2, nrm[4 * j + 1]), zi2s = cmul(zi2, zi2);
    v[0] = cmul(X1, zi1s); v[1] = cmul(Y1, cmul(zi1s, zi1));
    v[2] = cmul(X2, zi2s); v[3] = cmul(Y2, cmul(zi2s, zi2));
    if (op) {
      const Co<E> Z1s = cmul(Z1, Z1);
      Co<E> D, num, scale;
      if (dbl) { D = cadd(Y1, Y1); const Co<E> x2 = cmul(v[0], v[0]); num = cadd(cadd(x2, x2), x2); scale = cmul(Z1s, Z1); }
      else { const Co<E> Z2s = cmul(Z2, Z2); D = csub(cmul(X2, Z1s), cmul(X1, Z2s)); num = csub(v[3], v[1]); scale = cmul(Z1s, Z2s); }
      v[4] = cmul(cmul(num, scale), cinv_from_norm(D, nrm[4 * j + 3]));
      const u64* nsrc = dbl ? ja : jb;
      const Co<E> Xn = ldc<E>(nsrc + jac_at<E>(k, t + 1, 0)), Yn = ldc<E>(nsrc + jac_at<E>(k, t + 1, 1)), Zn = ldc<E>(nsrc + jac_at<E>(k, t + 1, 2));
      const Co<E> zin = cinv_from_norm(Zn, nrm[4 * j + 2]), zins = cmul(zin, zin);
      v[5] = cmul(Xn, zins); v[6] = cmul(Yn, cmul(zins, zin));
    }
    for (int w = 0; w < (op ? 7 : 4); w++)
      for (int q = 0; q < E; q++) {
        u64 s[4]; from_m(v[w].c[q], s);
        for (int i = 0; i < 4; i++) sv[(size_t)((w * E + q) * 4 + i) * n + row] = s[i];
      }
    row_op[row] = (unsigned char)op;
    if ((row & 511) == 511) {  // b at the last row is the output (g1/exp.rs:124-135, g2/exp.rs:139-156)
      for (int c = 0; c < 2; c++)
        for (int q = 0; q < E; q++) {
          u64 s[4]; from_m(v[2 + c].c[q], s);
          for (int i = 0; i < 8; i++) pi_out[16 * E * k + 8 * (c * E + q) + i] = (s[i >> 1] >> (32 * (i & 1))) & 0xffffffffULL;
        }
    }
  }
}

// One lane per (row, modular gadget): lane (row, g), g = blk * E + q, computes gadget blk (0: the zero polynomial of the
// slope, 1: new_x, 2: new_y; generate_g1_add / _double muladd.rs:124-177, 409-460 and their G2 twins) for the Fq2 component
// q -- its limb convolutions (component q of an Fq2 product is two 16 x 16 convolutions), its modular witness -- and writes
// that gadget's columns; the blk 0 lanes also write the limb columns of lambda / new_x / new_y, the blk 1 / 2 lanes those of
// a / b.  Lanes of a wave share g (lane index = g * n + row), so they neither diverge nor scatter their stores.
// Round 1 had one lane per ROW with all three (six) gadgets: 7.4 KB (13.5 KB) of scratch arrays per lane, 3,458 spilled
// registers, 1.98 GB of traffic for a 0.88 GB trace.
template <int E>
__device__ __forceinline__ void prod_comp(const int64_t (*X)[16], const int64_t (*Y)[16], int q, int64_t* out) {
  if (E == 1) { conv16(X[0], Y[0], out); return; }
  int64_t t[31];   // Fq2 limb product (src/fields/fq2.rs:41-58): c0 = x0*y0 - x1*y1, c1 = x0*y1 + x1*y0
  if (q == 0) { conv16(X[0], Y[0], out); conv16(X[E - 1], Y[E - 1], t); for (int k = 0; k < 31; k++) out[k] -= t[k]; }
  else { conv16(X[0], Y[E - 1], out); conv16(X[E - 1], Y[0], t); for (int k = 0; k < 31; k++) out[k] += t[k]; }
}
template <int E>
__global__ void __launch_bounds__(256) gadget_witness_kernel(const u64* __restrict__ sv, const unsigned char* __restrict__ row_op, size_t n, int gadget_col,
                                                             u64* __restrict__ trace, int* __restrict__ err) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (tid >= 3 * E * n) return;
  const size_t row = tid % n;
  const int g = (int)(tid / n), blk = g / E, q = g % E;
  const int op = row_op[row];
  // sv (affine_lambda_kernel): value w in {ax ay bx by lam nx ny}, component c, 64-bit word i at ((w * E + c) * 4 + i) * n + row
  auto ld = [&](int w, int c, u64* out4) { for (int i = 0; i < 4; i++) out4[i] = sv[(size_t)((w * E + c) * 4 + i) * n + row]; };
  auto ldl = [&](int w, int c, int64_t* out16) { u64 t4[4]; ld(w, c, t4); limbs16(t4, out16); };
  u64* gc = trace + (size_t)gadget_col * n + row;
  auto putc = [&](int c, u64 v) { gc[(size_t)c * n] = v; };
  if (blk >= 1)   // limb columns of a (blk 1: ax, ay) / b (blk 2: bx, by), component q: trace columns 16 (w E + q) ..
    for (int c = 0; c < 2; c++) {
      const int w = 2 * (blk - 1) + c;
      u64 s4[4]; ld(w, q, s4);
      for (int i = 0; i < 16; i++) trace[(size_t)(16 * (w * E + q) + i) * n + row] = (s4[i >> 2] >> (16 * (i & 3))) & 0xffff;
    }
  // G1Output / G2Output columns (muladd.rs:79-94, g2/muladd.rs:56-80): limbs of lambda, new_x, new_y (48 E), the zero gadgets
  // (79 each), the x and y gadgets (95 each), 3 E quotient signs
  const int base = blk == 0 ? 48 * E + 79 * q : (blk == 1 ? 127 * E + 95 * q : 222 * E + 95 * q), sgn = 317 * E + blk * E + q;
  if (!op) {   // G1Output / G2Output::default: zeros, quotient signs = 1
    if (blk == 0) for (int i = 0; i < 16; i++) { putc(16 * q + i, 0); putc(16 * E + 16 * q + i, 0); putc(32 * E + 16 * q + i, 0); }
    for (int c = 0; c < (blk ? 95 : 79); c++) putc(base + c, 0);
    putc(sgn, 1);
    return;
  }
  const bool dbl = op == 2;
  int64_t l[E][16], X[E][16], Y[E][16], pol[31], v16[16];
  u64 outv[4] = {0, 0, 0, 0};
  for (int c = 0; c < E; c++) ldl(4, c, l[c]);
  if (blk == 0) {
    for (int i = 0; i < 16; i++) putc(16 * q + i, (u64)l[q][i]);
    ldl(5, q, v16); for (int i = 0; i < 16; i++) putc(16 * E + 16 * q + i, (u64)v16[i]);
    ldl(6, q, v16); for (int i = 0; i < 16; i++) putc(32 * E + 16 * q + i, (u64)v16[i]);
    if (dbl) {   // 2 lambda y - 3 x^2
      int64_t t2[31];
      for (int c = 0; c < E; c++) { ldl(1, c, Y[c]); ldl(0, c, X[c]); }
      prod_comp<E>(l, Y, q, pol); prod_comp<E>(X, X, q, t2);
      for (int k = 0; k < 31; k++) pol[k] = 2 * pol[k] - 3 * t2[k];
    } else {     // lambda (x2 - x1) - (y2 - y1)
      for (int c = 0; c < E; c++) { ldl(2, c, X[c]); ldl(0, c, Y[c]); for (int i = 0; i < 16; i++) X[c][i] -= Y[c][i]; }
      prod_comp<E>(l, X, q, pol);
      ldl(3, q, v16); for (int i = 0; i < 16; i++) pol[i] -= v16[i];
      ldl(1, q, v16); for (int i = 0; i < 16; i++) pol[i] += v16[i];
    }
  } else if (blk == 1) {   // lambda^2 - x1 - x2
    prod_comp<E>(l, l, q, pol);
    ldl(0, q, v16); for (int i = 0; i < 16; i++) pol[i] -= v16[i];
    ldl(dbl ? 0 : 2, q, v16); for (int i = 0; i < 16; i++) pol[i] -= v16[i];
    ld(5, q, outv);
  } else {                 // lambda (x1 - new_x) - y1
    for (int c = 0; c < E; c++) { ldl(0, c, X[c]); ldl(5, c, Y[c]); for (int i = 0; i < 16; i++) X[c][i] -= Y[c][i]; }
    prod_comp<E>(l, X, q, pol);
    ldl(1, q, v16); for (int i = 0; i < 16; i++) pol[i] -= v16[i];
    ld(6, q, outv);
  }
  ModW w;
  if (!mod_witness(pol, outv, blk > 0, w)) { atomicOr(err, TG_ERR_WITNESS); return; }
  int cur = base;
  if (blk) for (int i = 0; i < 16; i++) putc(cur++, (u64)w.out_aux_red[i]);
  for (int i = 0; i < 17; i++) putc(cur++, (u64)w.quot_abs[i]);
  for (int i = 0; i < 31; i++) putc(cur++, (u64)w.aux_lo[i]);
  for (int i = 0; i < 31; i++) putc(cur++, (u64)w.aux_hi[i]);
  putc(sgn, w.sign > 0 ? (u64)1 : GLP - 1);
}

// flags columns (flags.rs:46-134) in closed form: within the 64-row block q of an instance the u32 limb e[q] is
// consumed one bit per two rows; the limb window rotates after row 62 of the block.  iow = u32 words per instance,
// the scalar sits in its last 8.
__global__ void flags_kernel(const uint32_t* __restrict__ ios, size_t iow, size_t n, int sf, u64* __restrict__ trace) {
  const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (row >= n) return;
  const size_t k = row >> 9; const int r = (int)(row & 511), q = r >> 6, s = r & 63, t = s >> 1;
  const uint32_t* e = ios + iow * k + (iow - 8);
  auto limb = [&](int i) -> u64 { return i < 8 ? (u64)e[i] : 0; };
  const u64 a = r & 1, b = 1 - a, bit = (limb(q) >> t) & 1;
  auto col = [&](int c) -> u64& { return trace[(size_t)(sf + c) * n + row]; };
  col(0) = r == 511; col(1) = s == 62; col(2) = a; col(3) = b; col(4) = bit * b; col(5) = bit;
  const int sh = s == 63 ? 1 : 0;
  col(6) = s == 63 ? limb(q + 1) : (limb(q) >> (t + 1));
  for (int i = 1; i < 8; i++) col(6 + i) = limb(q + i + sh);
}

// flags of the u64 table (fq12_u64/flags_u64.rs:34-94) in closed form: bit t of e is consumed on rows 2t, 2t+1 of the
// 128-row instance.  Columns is_final, a, b, filtered_bit, bit, val.
__global__ void flags_u64_kernel(const uint32_t* __restrict__ ios, size_t iow, size_t n, int sf, u64* __restrict__ trace) {
  const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (row >= n) return;
  const size_t k = row >> 7; const unsigned r = (unsigned)(row & 127), t = r >> 1;
  const u64 e = (u64)ios[iow * k + iow - 2] | ((u64)ios[iow * k + iow - 1] << 32);
  const u64 a = r & 1, b = 1 - a, bit = (e >> t) & 1, val = t == 63 ? 0 : e >> (t + 1);
  auto col = [&](int c) -> u64& { return trace[(size_t)(sf + c) * n + row]; };
  col(0) = r == 127; col(1) = a; col(2) = b; col(3) = bit * b; col(4) = bit; col(5) = val;
}

// inv[i] = 1/i in Goldilocks for i = 1..n-1 (inv[0] = 0).
__global__ void small_inverse_kernel(u64* __restrict__ inv, size_t n) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  inv[i] = i == 0 ? 0 : f_inv(F(i)).v;
}

// periodic pulse (pulse.rs:100-144) + io-pulse counter, and the lookup table column (range_check.rs:20-47).
__global__ void periodic_kernel(const u64* __restrict__ inv, size_t n, int start_periodic, int start_io_pulses, int start_lookups, u64 table_max, u64* __restrict__ trace) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 c = (i + 1) & 63;
  if (start_periodic >= 0) {  // the u64 table has no rotation pulse
    trace[(size_t)start_periodic * n + i] = c;
    trace[(size_t)(start_periodic + 1) * n + i] = c == 63 ? 0 : (-F(inv[63 - c])).v;
  }
  trace[(size_t)start_io_pulses * n + i] = i;
  trace[(size_t)start_lookups * n + i] = i < table_max ? i : table_max;  // 0..table_max then table_max repeated
}
// io pulses (pulse.rs:20-43): pulse q sits at row pos(q) = rpb*(q/2) + (q odd ? rpb-1 : 0); witness = 1/(i - pos).
__global__ void io_pulse_kernel(const u64* __restrict__ inv, size_t n, size_t rpb, int first_col, u64* __restrict__ trace) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int q = blockIdx.y;
  const size_t pos = (size_t)(q >> 1) * rpb + ((q & 1) ? rpb - 1 : 0);
  u64* w = trace + (size_t)(first_col + 2 * q) * n;  // witness_col(q); pulse_col(q) = witness_col(q) + 1
  w[i] = i > pos ? inv[i - pos] : (i < pos ? (-F(inv[pos - i])).v : 0);
  w[n + i] = i == pos;
}

// ---- Fq12ExpStark rows (src/fields/fq12/exp.rs:229-319) -----------------------------------------------------------------
// One lane per row.  ca / cb: the square-and-multiply chains of every instance in standard form, [K][257][12][4]
// (A[t] = x^(2^t), B[t]); row r of instance k: a = A[r>>1]; even rows: b = B[r>>1], multiply when the bit is set
// (product B[(r>>1)+1]); odd rows: b = B[(r>>1)+1], square (product A[(r>>1)+1]).  Writes the 384 limb columns of
// a and b and the 1344 Fq12Output columns (mul.rs:217-231) straight into the column-major trace.
// Fq12ExpU64Stark (fq12_u64/exp_u64.rs) is the same with 128 rows per instance: log_rpb = 7, iow = 194, 65 chain entries.
__global__ void __launch_bounds__(64) fq12_row_kernel(const uint32_t* __restrict__ ios, size_t iow, int log_rpb, const u64* __restrict__ ca, const u64* __restrict__ cb,
                                                      size_t n, u64* __restrict__ trace, int* __restrict__ err) {
  const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (row >= n) return;
  const size_t k = row >> log_rpb, cl = ((size_t)1 << (log_rpb - 1)) + 1;  // chain entries per instance
  const int r = (int)(row & (((size_t)1 << log_rpb) - 1)), t = r >> 1; const bool sq = r & 1;
  const bool bit = (ios[iow * k + 192 + (t >> 5)] >> (t & 31)) & 1;
  const int op = sq ? 1 : (bit ? 2 : 0);  // 1 square, 2 multiply
  const u64 (*a)[4] = reinterpret_cast<const u64 (*)[4]>(ca + ((k * cl + t) * 12) * 4);
  const u64 (*b)[4] = reinterpret_cast<const u64 (*)[4]>(cb + ((k * cl + t + (sq ? 1 : 0)) * 12) * 4);
  for (int c = 0; c < 12; c++)
    for (int i = 0; i < 16; i++) {
      trace[(size_t)(16 * c + i) * n + row] = (a[c][i >> 2] >> (16 * (i & 3))) & 0xffff;
      trace[(size_t)(192 + 16 * c + i) * n + row] = (b[c][i >> 2] >> (16 * (i & 3))) & 0xffff;
    }
  u64* g = trace + (size_t)384 * n + row;
  if (op) {
    const u64 (*prod)[4] = reinterpret_cast<const u64 (*)[4]>((op == 1 ? ca : cb) + ((k * cl + t + 1) * 12) * 4);
    if (!fq12_output_row(a, op == 1 ? a : b, prod, [&](int i, u64 v) { g[(size_t)i * n] = v; })) atomicOr(err, TG_ERR_WITNESS);
  } else {  // Fq12Output::default (mul.rs:179-187)
    for (int i = 0; i < 1332; i++) g[(size_t)i * n] = 0;
    for (int i = 1332; i < 1344; i++) g[(size_t)i * n] = 1;
  }
}

// ---- the square-and-multiply chains of Fq12ExpStark on the device (round 3) ------------------------------------------------------
// A[t] = x^(2^t), B[0] = offset, B[t+1] = bit_t ? A[t] B[t] : B[t] (fq12/exp.rs:165-221): 2 x 256 DEPENDENT Fq12 products per
// instance.  A lane would take ~0.2 s for them and the host pool 1 ms per instance and core (32 ms for 512 instances on a 16-CPU
// share); but one Fq12 product is 144 independent Fq products, so ONE WORKGROUP walks the chain of one instance: lanes 0..143 form
// a_i a_j, lanes 144..287 a_i b_j (Montgomery, eight 32-bit limbs), the products meet in LDS, and 24 lanes fold them into the 12 + 12
// new coefficients (flat basis, w^6 = 9 + i, exactly fq12_mul_m's sums: field arithmetic is exact, so the standard forms written
// to ca / cb are the host's words).  512 instances = 512 workgroups side by side.
__device__ __forceinline__ Fq fq12_fold_coeff(const Fq* P, int c) {   // coefficient c of the product from P[i * 12 + j] = x_i y_j
  const int m = c < 6 ? c : c - 6;
  const bool re = c < 6;
  const Fq z = {{0, 0, 0, 0}};
  Fq lo = z, hd = z, hs = z;     // d[m] (or s[m]),  d[m+6],  s[m+6]
  for (int i = 0; i < 6; i++) {
    const int jl = m - i, jh = m + 6 - i;
    if (jl >= 0) {
      if (re) lo = fadd(lo, fsub(P[i * 12 + jl], P[(i + 6) * 12 + jl + 6]));
      else lo = fadd(lo, fadd(P[i * 12 + jl + 6], P[(i + 6) * 12 + jl]));
    }
    if (jh >= 0 && jh <= 5) {
      hd = fadd(hd, fsub(P[i * 12 + jh], P[(i + 6) * 12 + jh + 6]));
      hs = fadd(hs, fadd(P[i * 12 + jh + 6], P[(i + 6) * 12 + jh]));
    }
  }
  if (m == 5) return lo;
  const Fq& nine = re ? hd : hs;
  Fq t = fadd(nine, nine); t = fadd(t, t); t = fadd(t, t); t = fadd(t, nine);     // 9 x
  return re ? fsub(fadd(lo, t), hs) : fadd(fadd(lo, hd), t);
}
__global__ void __launch_bounds__(320) fq12_chain_kernel(const uint32_t* __restrict__ ios, size_t iow, int steps, u64* __restrict__ ca, u64* __restrict__ cb,
                                                         u64* __restrict__ outs) {   // outs: [K][12][4], B[steps] again, compact (the public inputs' outputs)
  __shared__ Fq A[12], B[12], PA[144], PB[144];
  const size_t k = blockIdx.x;
  const uint32_t* io = ios + iow * k;
  const int tid = threadIdx.x, c = tid % 12;
  const bool is_a = tid < 12, is_b = tid >= 12 && tid < 24;
  if (is_a || is_b) {
    u64 t4[4]; u32x8_to_u64x4(io + (is_a ? 0 : 96) + 8 * c, t4);
    (is_a ? A : B)[c] = to_m(t4);
  }
  __syncthreads();
  for (int t = 0;; t++) {
    if (is_a || is_b) from_m((is_a ? A : B)[c], (is_a ? ca : cb) + ((k * (size_t)(steps + 1) + t) * 12 + c) * 4);
    if (t == steps) { if (is_b) from_m(B[c], outs + (k * 12 + c) * 4); break; }
    const bool bit = (io[192 + (t >> 5)] >> (t & 31)) & 1;          // uniform over the workgroup
    if (tid < 144) PA[tid] = mmul(A[tid / 12], A[tid % 12]);
    else if (tid < 288 && bit) PB[tid - 144] = mmul(A[(tid - 144) / 12], B[(tid - 144) % 12]);
    __syncthreads();
    Fq nv;
    if (is_a) nv = fq12_fold_coeff(PA, c);
    else if (is_b && bit) nv = fq12_fold_coeff(PB, c);
    __syncthreads();                                                   // every product has been read, A / B may change
    if (is_a) A[c] = nv; else if (is_b && bit) B[c] = nv;
    __syncthreads();
  }
}

// The same rows with one lane per (row, output coefficient) -- round 3.  fq12_row_kernel keeps 22 limb polynomials of 31 signed
// 64-bit coefficients per lane (d[11], s[11] of fq12_output_row: 5.4 KB) next to the operand limbs: 9.8 KB of scratch per lane and
// 420 VGPRs, every limb product a round trip through scratch memory (44 ms for the 2^18 rows of Fq12ExpStark(512)).  Coefficient c
// of x*y needs only  d[m] + 9 d[m+6] - s[m+6]  (c < 6, m = c)  or  s[m] + d[m+6] + 9 s[m+6]  (c >= 6, m = c - 6)  with
// d[q] = sum_{i+j=q} x_i y_j - x_{i+6} y_{j+6},  s[q] = sum_{i+j=q} x_i y_{j+6} + x_{i+6} y_j  (mul.rs:24-87, w^6 = 9 + i): a lane
// accumulates those (at most 22) weighted limb products straight into ONE polynomial of 31 coefficients held in registers, runs
// mod_witness on it and writes the 16 + 95 + 1 columns of its coefficient plus the limb columns of a[c] and b[c].  1.4 x the limb
// products of the shared form (d[m+6], s[m+6] are formed by two lanes), 12 x the lanes, no scratch arrays for the products.  Lanes
// of a wave share c (lane index = c * n + row): no divergence, coalesced stores.
__device__ __forceinline__ void conv16_acc(const u64* __restrict__ x4, const u64* __restrict__ y4, int w, int64_t* t) {
  int xl[16], yl[16];
#pragma unroll
  for (int i = 0; i < 16; i++) { xl[i] = (int)((x4[i >> 2] >> (16 * (i & 3))) & 0xffff); yl[i] = w * (int)((y4[i >> 2] >> (16 * (i & 3))) & 0xffff); }
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int j = 0; j < 16; j++) t[i + j] += (int64_t)xl[i] * (int64_t)yl[j];
}
__global__ void __launch_bounds__(256) fq12_gadget_kernel(const uint32_t* __restrict__ ios, size_t iow, int log_rpb, const u64* __restrict__ ca,
                                                          const u64* __restrict__ cb, size_t n, u64* __restrict__ trace, int* __restrict__ err) {
  const size_t tid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (tid >= 12 * n) return;
  const size_t row = tid % n;
  const int c = (int)(tid / n);
  const size_t k = row >> log_rpb, cl = ((size_t)1 << (log_rpb - 1)) + 1;  // chain entries per instance
  const int r = (int)(row & (((size_t)1 << log_rpb) - 1)), t = r >> 1; const bool sq = r & 1;
  const bool bit = (ios[iow * k + 192 + (t >> 5)] >> (t & 31)) & 1;
  const int op = sq ? 1 : (bit ? 2 : 0);  // 1 square, 2 multiply
  const u64* a = ca + ((k * cl + t) * 12) * 4;                       // [12][4]
  const u64* b = cb + ((k * cl + t + (sq ? 1 : 0)) * 12) * 4;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    trace[(size_t)(16 * c + i) * n + row] = (a[4 * c + (i >> 2)] >> (16 * (i & 3))) & 0xffff;
    trace[(size_t)(192 + 16 * c + i) * n + row] = (b[4 * c + (i >> 2)] >> (16 * (i & 3))) & 0xffff;
  }
  u64* g = trace + (size_t)384 * n + row;
  auto put = [&](int i, u64 v) { g[(size_t)i * n] = v; };
  if (!op) {  // Fq12Output::default (mul.rs:179-187): zeros, quotient signs 1
    for (int i = 0; i < 16; i++) put(16 * c + i, 0);
    for (int i = 0; i < 95; i++) put(192 + 95 * c + i, 0);
    put(1332 + c, 1);
    return;
  }
  const u64* x = a;
  const u64* y = op == 1 ? a : b;
  const u64* out = (op == 1 ? ca : cb) + ((k * cl + t + 1) * 12 + c) * 4;
  const int m = c < 6 ? c : c - 6;
  const bool re = c < 6;
  int64_t pol[31];
#pragma unroll
  for (int q = 0; q < 31; q++) pol[q] = 0;
#pragma unroll 1
  for (int i = 0; i < 6; i++) {
    const int jl = m - i, jh = m + 6 - i;          // i + j = m (weights of d[m] / s[m]) and i + j = m + 6 (of d[m+6], s[m+6])
    if (jl >= 0) {
      if (re) { conv16_acc(x + 4 * i, y + 4 * jl, 1, pol); conv16_acc(x + 4 * (i + 6), y + 4 * (jl + 6), -1, pol); }
      else { conv16_acc(x + 4 * i, y + 4 * (jl + 6), 1, pol); conv16_acc(x + 4 * (i + 6), y + 4 * jl, 1, pol); }
    }
    if (jh >= 0 && jh <= 5) {
      const int wd = re ? 9 : 1, ws = re ? -1 : 9;
      conv16_acc(x + 4 * i, y + 4 * jh, wd, pol); conv16_acc(x + 4 * (i + 6), y + 4 * (jh + 6), -wd, pol);
      conv16_acc(x + 4 * i, y + 4 * (jh + 6), ws, pol); conv16_acc(x + 4 * (i + 6), y + 4 * jh, ws, pol);
    }
  }
  ModW w;
  if (!mod_witness(pol, out, true, w)) { atomicOr(err, TG_ERR_WITNESS); return; }
#pragma unroll
  for (int i = 0; i < 16; i++) put(16 * c + i, (out[i >> 2] >> (16 * (i & 3))) & 0xffff);
  int cur = 192 + 95 * c;
  for (int i = 0; i < 16; i++) put(cur++, (u64)w.out_aux_red[i]);
  for (int i = 0; i < 17; i++) put(cur++, (u64)w.quot_abs[i]);
  for (int i = 0; i < 31; i++) put(cur++, (u64)w.aux_lo[i]);
  for (int i = 0; i < 31; i++) put(cur++, (u64)w.aux_hi[i]);
  put(1332 + c, w.sign > 0 ? (u64)1 : GLP - 1);
}

// ---- FqExpStark rows (src/fields/fq/exp.rs:128-176): as fq12_row_kernel with one coefficient; ca / cb: [K][257][4] ---------
__global__ void __launch_bounds__(128) fq_row_kernel(const uint32_t* __restrict__ ios, const u64* __restrict__ ca, const u64* __restrict__ cb, size_t n,
                                                     u64* __restrict__ trace, int* __restrict__ err) {
  const size_t row = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (row >= n) return;
  const size_t k = row >> 9; const int r = (int)(row & 511), t = r >> 1; const bool sq = r & 1;
  const bool bit = (ios[24 * k + 16 + (t >> 5)] >> (t & 31)) & 1;
  const int op = sq ? 1 : (bit ? 2 : 0);  // 1 square, 2 multiply
  u64 a[4], b[4], prod[4];
  for (int i = 0; i < 4; i++) { a[i] = ca[(k * 257 + t) * 4 + i]; b[i] = cb[(k * 257 + t + (sq ? 1 : 0)) * 4 + i]; }
  for (int i = 0; i < 16; i++) {
    trace[(size_t)i * n + row] = (a[i >> 2] >> (16 * (i & 3))) & 0xffff;
    trace[(size_t)(16 + i) * n + row] = (b[i >> 2] >> (16 * (i & 3))) & 0xffff;
  }
  u64* g = trace + (size_t)32 * n + row;
  if (op) {
    for (int i = 0; i < 4; i++) prod[i] = (op == 1 ? ca : cb)[(k * 257 + t + 1) * 4 + i];
    u64 lv[112];
    if (!fq_output_row(a, op == 1 ? a : b, prod, lv)) { atomicOr(err, TG_ERR_WITNESS); return; }
    for (int c = 0; c < 112; c++) g[(size_t)c * n] = lv[c];
  } else {  // FqOutput::default (mul.rs:24-32)
    for (int c = 0; c < 111; c++) g[(size_t)c * n] = 0;
    g[(size_t)111 * n] = 1;
  }
}

// ---- split range check (range_check.rs:116-160): table 0..255 then 255 repeated; per target column the low and the high
// byte, each with its sorted column and permuted table (lookup.rs:60-111).  One workgroup of 256 lanes per target.
// With 256 distinct values the reference's merge is simulated value by value by one lane (push a missing value; the
// first occurrence of a present value takes its table entry, repeats pop the pool or are deferred; repeats of 255 keep
// matching the table's tail of 255s), then every sorted slot is classified independently from the per-value records.
__global__ void __launch_bounds__(256) split_range_check_kernel(u64* __restrict__ trace, size_t n, int first_target, int table_col, int* __restrict__ err) {
  __shared__ unsigned hist[2][256], tincl[256], npop[256], pbase[256], defbase[256];
  __shared__ unsigned char pool[256], poplist[256];
  __shared__ unsigned misc[8];
  const int tid = threadIdx.x, kcol = blockIdx.x;
  const u64* col = trace + (size_t)(first_target + kcol) * n;
  u64* base = trace + (size_t)(table_col + 1 + 6 * kcol) * n;  // lo, sorted lo, perm lo, hi, sorted hi, perm hi
  hist[0][tid] = 0; hist[1][tid] = 0;
  __syncthreads();
  bool bad = false;
  for (size_t i = tid; i < n; i += 256) {
    const u64 v = col[i];
    if (v >= 65536) { bad = true; continue; }
    const unsigned lo = (unsigned)v & 255, hi = (unsigned)v >> 8;
    base[i] = lo; base[3 * n + i] = hi;
    atomicAdd(&hist[0][lo], 1u); atomicAdd(&hist[1][hi], 1u);
  }
  if (bad) atomicOr(err, TG_ERR_RANGE);
  __syncthreads();
  for (int h = 0; h < 2; h++) {
    u64* sorted_out = base + (size_t)(3 * h + 1) * n;
    u64* perm_out = sorted_out + n;
    // inclusive prefix counts
    unsigned incl = hist[h][tid];
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) misc[tid >> 6] = incl;
    __syncthreads();
    for (int w = 0; w < (tid >> 6); w++) incl += misc[w];
    tincl[tid] = incl;
    __syncthreads();
    if (tid == 0) {
      unsigned sp = 0, npl = 0, ndef = 0;
      for (int v = 0; v < 255; v++) {
        const unsigned c = hist[h][v];
        npop[v] = 0; pbase[v] = npl; defbase[v] = ndef;
        if (c == 0) { pool[sp++] = (unsigned char)v; continue; }
        const unsigned k = c - 1 < sp ? c - 1 : sp;
        for (unsigned j = 0; j < k; j++) poplist[npl++] = pool[--sp];
        npop[v] = k; ndef += c - 1 - k;
      }
      const unsigned c255 = hist[h][255];
      const size_t tail = n - 255;             // table entries equal to 255
      npop[255] = 0; pbase[255] = npl; defbase[255] = ndef;
      if (c255 > tail) ndef += (unsigned)(c255 - tail);
      misc[4] = sp; misc[5] = ndef;
    }
    __syncthreads();
    const unsigned sp = misc[4];
    const size_t tail = n - 255;
    for (size_t i = tid; i < n; i += 256) {
      // value of sorted slot i: first v with tincl[v] > i
      int lo = 0, hi = 255;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (tincl[mid] > i) hi = mid; else lo = mid + 1; }
      const int v = lo;
      const size_t d = i - (tincl[v] - hist[h][v]);
      unsigned out;
      if (v < 255) {
        if (d == 0) out = v;
        else if (d <= npop[v]) out = poplist[pbase[v] + d - 1];
        else { const unsigned rank = defbase[v] + (unsigned)(d - 1 - npop[v]); out = rank < sp ? pool[rank] : 255u; }
      } else {
        if (d < tail) out = 255u;
        else { const unsigned rank = defbase[255] + (unsigned)(d - tail); out = rank < sp ? pool[rank] : 255u; }
      }
      sorted_out[i] = (u64)v; perm_out[i] = (u64)out;
    }
    __syncthreads();
  }
}

// ---- u16 range check (range_check.rs:20-47, lookup.rs:60-111), n == 65536 ------------------------------------------
// One workgroup per range-checked column; everything after the histogram stays in LDS.
//
// The reference merges the sorted column against the table 0..65535 (lookup.rs:60-111): a table value missing from
// the column goes on a LIFO pool; the first occurrence of a present value takes its own table entry; every further
// occurrence pops the pool or, when the pool is empty, is deferred; deferred slots finally take the remaining pool
// bottom-up (the repeats of 65535 are always deferred: the loop has run off the table by then).  Replaying that
// serially costs a lane ~400 cycles per value, so it is restated as independent searches:
//   c[v] = multiplicity, T[v] = #entries <= v (LDS, u16; 65536 is implied for v >= max value),
//   S[v] = v + 1 - T[v]  = (#pushes - #pops) after value v, unclamped, for v <= 65534,
//   a pushed value u is popped by the first x > u with S[x] < S[u], into sorted slot x - S[u] + 1;
//   if there is no such x it is still pooled at the end, at height h = S[u] - min(0, min S), and fills the deferred
//   slot of rank h-1, which is slot v + h where v is the first value with S[v] < -(h-1) (or a repeat of 65535).
// Every lane owns 64 consecutive values; "first x >= from with S[x] < target" walks a two-level min tree (32 / 1024).
// tools/microbench/range_check_phases.hip defines SBN_RC_PROFILE: workgroup 0 then stamps the constant 100 MHz clock at the phase
// boundaries of range_check_kernel (a barrier in front of each stamp); the library never defines it.
#ifdef SBN_RC_PROFILE
__device__ unsigned long long g_rc_prof[16];
#define RC_MARK(k) do { __syncthreads(); if (blockIdx.x == 0 && threadIdx.x == 0) g_rc_prof[k] = wall_clock64(); } while (0)
#else
#define RC_MARK(k) do { } while (0)
#endif
static constexpr int RC_THREADS = 1024;
static constexpr int RC_HEAVY_MAX = 1024;  // values with more than 64 (BIG: 256) occurrences are expanded by the whole block
static constexpr size_t RC_LDS_BYTES = 65536 * 2 + 2 * 2048 * 4 + 64 * 4 + 64 * 4 + RC_HEAVY_MAX * 8;

// Tables with MORE than 2^16 rows (G1ExpStark(256) ... : the reference pads to any power of two >= 128 instances,
// src/curves/g1/circuit.rs:273-277): multiplicities no longer fit the u16 LDS counters, so range_count_kernel builds
// the histogram of every target column in HBM (u32, one atomic per row) and the BIG variant of range_check_kernel reads
// it.  The prefix counts T[v] <= n stay in LDS as T mod 2^16 (u16) plus the <= 3 values at which T crosses a multiple
// of 2^16 (T is monotone, so the high part is the number of those thresholds <= v).
__global__ void range_count_kernel(const u64* __restrict__ trace, size_t n, int first_col, unsigned int* __restrict__ cnt, int* __restrict__ err) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u64 v = trace[(size_t)(first_col + blockIdx.y) * n + i];
  if (v >= 65536) { atomicOr(err, TG_ERR_RANGE); return; }
  atomicAdd(&cnt[(size_t)blockIdx.y * 65536 + v], 1u);
}

// The table column is 0..65534 once each, then 65535 K = n - 65535 times (range_check.rs:20-47: min(i, 65535)); the merge
// gives the first min(c, K) occurrences of 65535 their own table entry (Ordering::Equal), further occurrences run off
// the table and are deferred, and table copies left over when the inputs end are appended to the pool (lookup.rs:100-101).
template <bool BIG>
__global__ void __launch_bounds__(RC_THREADS) __attribute__((amdgpu_waves_per_eu(5, 5))) range_check_kernel(u64* __restrict__ trace, size_t n, int first_col, int start_lookups, int* __restrict__ err,
                                                                 const unsigned int* __restrict__ cnt, int old_form) {
  extern __shared__ unsigned int lds[];
  unsigned int* t32 = lds;                                                  // 32768 words = 65536 u16 counters, then T (mod 2^16)
  const unsigned short* t16 = reinterpret_cast<const unsigned short*>(lds);
  int* L1 = reinterpret_cast<int*>(lds + 32768);                            // min S over 32 values
  int* SM1 = L1 + 2048;                                                     // suffix minima of L1
  int* L2 = SM1 + 2048;                                                     // min over 32 L1 entries
  unsigned int* misc = reinterpret_cast<unsigned int*>(L2 + 64);            // [0..15] wave sums, [32] differs, [33] vmax, [34] heavy count, [40..42] thresholds
  uint2* heavy = reinterpret_cast<uint2*>(misc + 64);                       // (value, start) of values with many occurrences
  const int tid = threadIdx.x, kcol = blockIdx.x;
  const u64* col = trace + (size_t)(first_col + kcol) * n;
  u64* sorted_out = trace + (size_t)(start_lookups + 1 + 2 * kcol) * n;
  u64* perm_out = sorted_out + n;
  constexpr int INF = 0x3fffffff;
  const int N = (int)n, KT = N - 65535;             // rows; copies of 65535 in the table
  constexpr int HEAVY = BIG ? 256 : 64;

  RC_MARK(0);
  for (int i = tid; i < 32768; i += RC_THREADS) t32[i] = 0;
  if (tid < 64) misc[tid] = tid >= 40 && tid < 44 ? 65536u : 0u;
  __syncthreads();
  RC_MARK(1);
  if constexpr (!BIG) {
    const u64 v0 = col[0];
    bool differs = false, bad = false; unsigned vm = 0;
    for (size_t i0 = tid; i0 < n; i0 += 8 * RC_THREADS) {   // eight loads in flight, then their eight LDS atomics
      u64 vv[8];
#pragma unroll
      for (int k = 0; k < 8; k++) { const size_t i = i0 + (size_t)k * RC_THREADS; vv[k] = i < n ? col[i] : v0; }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const u64 v = vv[k];
        if (i0 + (size_t)k * RC_THREADS >= n) continue;
        if (v >= 65536) { bad = true; continue; }
        differs |= v != v0; vm = vm > (unsigned)v ? vm : (unsigned)v;
        atomicAdd(&t32[v >> 1], 1u << (16 * (v & 1)));
      }
    }
    if (bad) atomicOr(err, TG_ERR_RANGE);
    if (differs) misc[32] = 1;
    atomicMax(&misc[33], vm);
    __syncthreads();
    // a constant column wrapped its single u16 counter (65536 occurrences): clear it, T is implied by vmax alone
    if (misc[32] == 0 && tid == 0) t32[v0 >> 1] = 0;
    __syncthreads();
  }

  RC_MARK(2);
  // counts -> inclusive prefix counts T (mod 2^16), in place
  if constexpr (!BIG) {
    uint4* p = reinterpret_cast<uint4*>(t32 + 32 * tid);
    unsigned cw[32];
    for (int j = 0; j < 8; j++) { uint4 q = p[j]; cw[4 * j] = q.x; cw[4 * j + 1] = q.y; cw[4 * j + 2] = q.z; cw[4 * j + 3] = q.w; }
    unsigned tot = 0;
    for (int j = 0; j < 32; j++) tot += (cw[j] & 0xffff) + (cw[j] >> 16);
    unsigned incl = tot;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) misc[tid >> 6] = incl;
    __syncthreads();
    unsigned run = incl - tot;
    for (int w = 0; w < (tid >> 6); w++) run += misc[w];
    for (int j = 0; j < 32; j++) { unsigned lo = run + (cw[j] & 0xffff), hi = lo + (cw[j] >> 16); run = hi; cw[j] = (lo & 0xffff) | (hi << 16); }
    for (int j = 0; j < 8; j++) p[j] = make_uint4(cw[4 * j], cw[4 * j + 1], cw[4 * j + 2], cw[4 * j + 3]);
  } else {
    // this lane's 64 values: counts from the HBM histogram (twice: total first, then the running prefix)
    const unsigned int* cg = cnt + (size_t)kcol * 65536 + 64 * tid;
    unsigned tot = 0, vm = 0;
    for (int j = 0; j < 64; j++) { const unsigned c = cg[j]; tot += c; if (c) vm = (unsigned)(64 * tid + j); }
    atomicMax(&misc[33], vm);
    unsigned incl = tot;
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(incl, o, 64); if ((tid & 63) >= o) incl += t; }
    if ((tid & 63) == 63) misc[tid >> 6] = incl;
    __syncthreads();
    unsigned run = incl - tot;
    for (int w = 0; w < (tid >> 6); w++) run += misc[w];
    for (int j = 0; j < 32; j++) {
      const unsigned before = run;
      const unsigned lo = run + cg[2 * j], hi = lo + cg[2 * j + 1];
      run = hi;
      t32[32 * tid + j] = (lo & 0xffff) | (hi << 16);
      // thresholds: the first value at which T reaches k * 2^16 (k = 1..3)
      for (unsigned k = (before >> 16) + 1; k <= (lo >> 16) && k < 4; k++) atomicMin(&misc[39 + k], (unsigned)(64 * tid + 2 * j));
      for (unsigned k = (lo >> 16) + 1; k <= (hi >> 16) && k < 4; k++) atomicMin(&misc[39 + k], (unsigned)(64 * tid + 2 * j + 1));
    }
  }
  __syncthreads();
  RC_MARK(3);
  const int vmax = (int)misc[33];
  const int thr1 = (int)misc[40], thr2 = (int)misc[41], thr3 = (int)misc[42];
  auto Thi = [&](int v) __attribute__((always_inline)) -> int { return BIG ? ((v >= thr1) + (v >= thr2) + (v >= thr3)) << 16 : 0; };
  auto T = [&](int v) __attribute__((always_inline)) -> int { return v < 0 ? 0 : (v >= vmax ? N : (int)t16[v] + Thi(v)); };
  auto S = [&](int v) __attribute__((always_inline)) -> int { return v >= 65535 ? INF : v + 1 - T(v); };
  unsigned long long pool = 0;   // two-pass form: the unused table values of this lane's segment that nothing inside the segment took
  int s_end = 0;                 // pool height after the segment's last value below 65535
  if (old_form) {
    for (int b = 0; b < 2; b++) {
      int m = INF;
      for (int j = 0; j < 32; j++) { int s = S(tid * 64 + b * 32 + j); m = s < m ? s : m; }
      L1[2 * tid + b] = m;
    }
  } else {
    // Lane l owns the values 64 l .. 64 l + 63: their 64 prefix counts come out of LDS once (eight 128-bit reads) and serve both the
    // two minima of the search tree and the LIFO pool INSIDE the segment, which is a bit mask: an absent value sets its bit (push), the
    // k-th repeat of a present value takes the highest set bit (pop: slot tprev + k, the same slot the search x - S(u) + 1 of the
    // one-pass form finds, since S(u) = S(x - 1) - k + 1 for the k-th pop at x).  What is still set at the end goes through the
    // search tree below; the one-pass form searched for EVERY absent value, a wave waiting for the longest of its 64 searches in
    // every iteration (2.2e8 wave-instructions per launch, as many as a sponge launch: profiles/r3_v10_pmc_summary.json).
    const uint4* p = reinterpret_cast<const uint4*>(t32 + 32 * tid);
    unsigned cw[32];
#pragma unroll
    for (int j = 0; j < 8; j++) { const uint4 q = p[j]; cw[4 * j] = q.x; cw[4 * j + 1] = q.y; cw[4 * j + 2] = q.z; cw[4 * j + 3] = q.w; }
    int tprev = T(tid * 64 - 1), m0 = INF, m1 = INF;
#pragma unroll
    for (int j = 0; j < 64; j++) {
      const int v = tid * 64 + j;
      const int tv = v >= vmax ? N : (int)((cw[j >> 1] >> (16 * (j & 1))) & 0xffff) + Thi(v);
      if (v < 65535) {
        const int sv = v + 1 - tv;
        if (j < 32) m0 = sv < m0 ? sv : m0; else m1 = sv < m1 ? sv : m1;
        s_end = sv;
        const int c = tv - tprev;
        if (c == 0) pool |= 1ull << j;
        else
          for (int k = 1; k < c && pool; k++) {
            const int u = 63 - __builtin_clzll(pool);
            pool &= ~(1ull << u);
            perm_out[tprev + k] = (u64)(tid * 64 + u);
          }
      }
      tprev = tv;
    }
    L1[2 * tid] = m0; L1[2 * tid + 1] = m1;
  }
  __syncthreads();
  if (tid < 64) { int m = INF; for (int j = 0; j < 32; j++) { int s = L1[tid * 32 + j]; m = s < m ? s : m; } L2[tid] = m; }
  __syncthreads();
  int mfin = 0;
  for (int j = 0; j < 64; j++) { int s = L2[j]; mfin = s < mfin ? s : mfin; }

  // suffix minima of L1: SM1[b] = min L1[b..2047] (lets a push that is never popped stop after its own 32-block)
  if (tid < 64) {
    int m = INF;
    for (int g = tid + 1; g < 64; g++) { int s = L2[g]; m = s < m ? s : m; }
    for (int j = 31; j >= 0; j--) { int s = L1[tid * 32 + j]; m = s < m ? s : m; SM1[tid * 32 + j] = m; }
  }
  __syncthreads();

  RC_MARK(4);
  // first x in [from, from | 31] with S(x) < target, or -1 (8 prefix counts per LDS read)
  auto scan32 = [&](int from, int target) __attribute__((always_inline)) -> int {
    const int hi = from | 31;
    for (int a = from & ~7; a <= hi; a += 8) {
      const uint4 q = *reinterpret_cast<const uint4*>(t16 + a);
      const unsigned w[4] = {q.x, q.y, q.z, q.w};
      for (int e = 0; e < 8; e++) {
        const int x = a + e;
        const int tv = x >= vmax ? N : (int)((w[e >> 1] >> (16 * (e & 1))) & 0xffff) + Thi(x);
        const int sx = x >= 65535 ? INF : x + 1 - tv;
        if (x >= from && sx < target) return x;
      }
    }
    return -1;
  };
  // first index in [from, hi] with arr[index] < target, or -1 (arr 16-byte aligned, hi + 1 a multiple of 4)
  auto scan_min = [&](const int* arr, int from, int hi, int target) __attribute__((always_inline)) -> int {
    for (int a = from & ~3; a <= hi; a += 4) {
      const int4 q = *reinterpret_cast<const int4*>(arr + a);
      if (a >= from && q.x < target) return a;
      if (a + 1 >= from && q.y < target) return a + 1;
      if (a + 2 >= from && q.z < target) return a + 2;
      if (a + 3 >= from && q.w < target) return a + 3;
    }
    return -1;
  };
  // first x in [from, 65534] with S(x) < target, or -1.  (Measured and dropped, round 4: issuing every level's LDS reads together
  // and scanning registers -- 0.48 -> 0.65 ms per launch in tools/microbench/range_check_phases.hip: with 16 waves per CU the walk
  // is bound by the instructions of its longest searches, not by LDS latency.)
  auto firstbelow = [&](int from, int target) __attribute__((always_inline)) -> int {
    if (from > 65534) return -1;
    const int x = scan32(from, target);
    if (x >= 0) return x;
    const int b = (from >> 5) + 1;
    if (b >= 2048 || SM1[b] >= target) return -1;
    int fb = scan_min(L1, b, b | 31, target);
    if (fb < 0) { const int fg = scan_min(L2, (b >> 5) + 1, 63, target); fb = scan_min(L1, fg * 32, fg * 32 + 31, target); }
    return scan32(fb * 32, target);
  };
  const int t_last = T(65534);  // = start of the 65535 run
  // deferred slot of rank `rank`: the first -mfin of them are repeats met with an empty pool below 65535, the rest are the
  // occurrences of 65535 beyond the table's KT copies (slots t_last + KT ...)
  auto assign_deferred = [&](int rank, int val) __attribute__((always_inline)) {
    int slot;
    if (rank < -mfin) { const int v = firstbelow(0, -rank); slot = v + rank + 1; }
    else slot = t_last + KT + (rank + mfin);
    perm_out[slot] = (u64)val;
  };

  if (old_form) {   // round 3 (SBN_RANGE_CHECK=1, A/B): one pass, lane l owns the values 64 l .. 64 l + 63, every unused table value searches its slot
    int tprev = T(tid * 64 - 1);
    for (int j = 0; j < 64; j++) {
      const int v = tid * 64 + j;
      if (v == 65535) break;                           // handled by the whole block below
      const int tv = T(v), c = tv - tprev;
      if (c > 0) {
        perm_out[tprev] = (u64)v;  // first occurrence takes its own table entry
        if (c <= HEAVY) { for (int d = 0; d < c; d++) sorted_out[tprev + d] = (u64)v; }
        else { const unsigned h = atomicAdd(&misc[34], 1u); heavy[h] = make_uint2((unsigned)v, (unsigned)tprev); }
      } else {
        const int s = v + 1 - tv;
        const int x = firstbelow(v + 1, s);
        if (x >= 0) perm_out[x - s + 1] = (u64)v;
        else assign_deferred(s - mfin - 1, v);
      }
      tprev = tv;
    }
  } else {
    // The pushes left in the segment's pool, top first: their levels are s_end, s_end - 1, ... (nothing after a left-over push
    // drops below it inside the segment), so ONE forward walk serves them all -- the search for level s - 1 continues at the
    // value that popped level s -- and it starts at the next segment.  Once a level is never undercut, neither are the lower ones:
    // those pushes stay in the pool to the end and fill the deferred slots.
    {
      int from = tid * 64 + 64, s = s_end;
      while (pool) {
        const int u = 63 - __builtin_clzll(pool);
        const int x = firstbelow(from, s);
        if (x < 0) break;
        pool &= ~(1ull << u);
        perm_out[x - s + 1] = (u64)(tid * 64 + u);
        from = x; s--;
      }
      // never popped: BOTTOM first -- ranks ascend, so the value at which the pool first runs `rank + 1` short (the deferred slot of
      // that rank sits right behind it) moves forward too, and this walk also continues where the last one ended
      if (pool) {
        int lvl = s - __builtin_popcountll(pool) + 1, dfrom = 0;
        while (pool) {
          const int u = __builtin_ctzll(pool);
          pool &= pool - 1;
          const int rank = lvl - mfin - 1;
          int slot;
          if (rank < -mfin) { const int vv = firstbelow(dfrom, -rank); slot = vv + rank + 1; dfrom = vv; }
          else slot = t_last + KT + (rank + mfin);
          perm_out[slot] = (u64)(tid * 64 + u);
          lvl++;
        }
      }
    }
    RC_MARK(5);
    // Lane l <-> values l, l + 1024, ...: neighbouring lanes write neighbouring slots (the one-pass form stored with a stride of
    // ~512 bytes between lanes: 3.2 x the algorithmic write traffic).  The sorted copy and the table entry of every first occurrence.
    for (int j = 0; j < 64; j++) {
      const int v = j * RC_THREADS + tid;
      if (v == 65535) break;                           // handled by the whole block below
      const int tp = T(v - 1), c = T(v) - tp;
      if (c > 0) {
        perm_out[tp] = (u64)v;
        if (c <= HEAVY) { for (int d = 0; d < c; d++) sorted_out[tp + d] = (u64)v; }
        else { const unsigned h = atomicAdd(&misc[34], 1u); heavy[h] = make_uint2((unsigned)v, (unsigned)tp); }
      }
    }
    RC_MARK(6);
  }
  __syncthreads();
  RC_MARK(7);
  const unsigned nheavy = misc[34];
  for (unsigned h = 0; h < nheavy; h++) {
    const int v = (int)heavy[h].x, st = (int)heavy[h].y, c = T(v) - st;
    for (int d = tid; d < c; d += RC_THREADS) sorted_out[st + d] = (u64)v;
  }
  // the run of 65535: c65 occurrences against KT table copies
  {
    const int c65 = N - t_last, eq = c65 < KT ? c65 : KT;
    for (int d = tid; d < c65; d += RC_THREADS) sorted_out[t_last + d] = 65535;
    for (int d = tid; d < eq; d += RC_THREADS) perm_out[t_last + d] = 65535;
    const int pooled = S(65534) - mfin;              // pool height after the values below 65535 = rank of the first left-over 65535
    for (int r = tid; r < KT - c65; r += RC_THREADS) assign_deferred(pooled + r, 65535);
  }
  RC_MARK(8);
}

}  // namespace tg
