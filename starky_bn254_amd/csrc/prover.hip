// Host orchestration of prove() on one MI355X: replaces starky prover.rs `prove` for the reference's
// G1 tables (call sites src/curves/g1/exp.rs:818-825, src/curves/g1/muladd.rs:669-676).  The host
// owns only the Fiat-Shamir transcript and proof assembly; every polynomial / hashing / constraint
// stage runs in the kernels of kernels.cuh on one HIP stream.
#include "host_common.hpp"
#include "settings.hpp"
#include "kernels.cuh"
#include "kernels_tracegen.cuh"
#include <algorithm>
#include <cstring>
#include <cstdlib>
#include <thread>
#include <atomic>
#include <chrono>

using namespace sbn;

namespace sbn { thread_local std::string g_last_error; }

#define HIPC(expr)                                                                                   \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) return fail(SBN_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// sbn_set_device: the device of the provers created afterwards by the calling thread, and the default of threads that never
// chose one; sbn_set_thread_device: the calling thread only (the ranks of a local split group are threads with one device each:
// they must not move the process default under the main thread).  Both also make it HIP's current device of the calling thread,
// so that a plain-C caller's next HIP-level step (sbn_rccl_comm_create allocates its staging buffers) lands on the same GPU.
static std::atomic<int> g_default_device{0};
static thread_local int t_device = -1;
static inline int cur_device() { return t_device >= 0 ? t_device : g_default_device.load(); }
#define g_device (cur_device())
namespace sbn { int current_device() { return cur_device(); } }   // transport.hip

// Stage k spans [ev[k], ev[k+1]) on the prover's main stream.  The commit stages overlap NTT (main
// stream) with sponge absorption (hash stream); the absorption kernels are additionally timed one by
// one with events on the hash stream (EXTRA_* entries = sum over the chunk launches of one proof).
enum Stage {
  ST_TRACE_COMMIT, ST_PERM_Z, ST_Z_COMMIT, ST_QUOTIENT_EVAL, ST_QUOTIENT_COMMIT,
  ST_OPENINGS, ST_FRI_COMBINE, ST_FRI_LAYERS, ST_POW, ST_QUERIES, ST_COUNT
};
enum Extra { EX_TRACE_ABSORB_MS, EX_TRACE_ABSORB_LAUNCHES, EX_Z_ABSORB_MS, EX_Z_ABSORB_LAUNCHES, EX_TRACEGEN_MS, EX_COMM_MS, EX_COUNT };
static const char* STAGE_NAMES[ST_COUNT + EX_COUNT] = {
  "trace_commit", "perm_z", "z_commit", "quotient_eval", "quotient_commit",
  "openings", "fri_combine", "fri_layers", "pow", "queries",
  "trace_absorb_kernels_ms", "trace_absorb_launches", "z_absorb_kernels_ms", "z_absorb_launches", "device_tracegen_ms",
  "split_exchange_ms"};
static constexpr int MAX_CHUNKS = 512;

struct DevTree {  // Merkle digests, levels concatenated (leaf level first)
  u64* d = nullptr; size_t nleaf = 0; u32 nlevels = 0;  // nlevels = number of levels BELOW the cap
  u64* level(u32 l) const { return d + (2 * nleaf - ((2 * nleaf) >> l)) * 4; }
};

// Oversized-trace split (include/sbn.h, sbn_split_prover_*): this rank's share of one proof.
// The columns of a matrix are dealt to the ranks in blocks of `ob` columns, round-robin: rank r owns the blocks
// b = r, r + R, r + 2R, ... and keeps them compactly (own block k = global block k * R + r at local columns k * ob ...);
// only the globally last block can be short, and it is the last own block of its owner.  In step k of a commitment every
// rank transforms its own block k, the all-to-all of that step moves the blocks k * R .. k * R + R - 1 to their row
// owners, and the leaf sponge -- sequential over the columns of a row -- absorbs exactly those blocks next.
struct ColShare {
  size_t total = 0, ob = 64; u32 R = 1, rank = 0;
  size_t nblocks() const { return (total + ob - 1) / ob; }
  size_t steps() const { return (nblocks() + R - 1) / R; }
  size_t block_cols(size_t b) const { return b < nblocks() ? std::min(ob, total - b * ob) : 0; }
  size_t own_cols(u32 r) const { size_t s = 0; for (size_t b = r; b < nblocks(); b += R) s += block_cols(b); return s; }
  size_t own() const { return own_cols(rank); }
  size_t max_own() const { size_t s = 0; for (u32 r = 0; r < R; r++) s = std::max(s, own_cols(r)); return s; }
  size_t global_col(u32 r, size_t local) const { return ((local / ob) * R + r) * ob + local % ob; }   // of rank r's local column
};
struct SplitCtx {
  sbn_comm comm;
  u32 log_r = 0, rho = 0;                 // world = 2^log_r; this rank owns the LDE rows i = j * world + rho
  size_t ml = 0;                          // local LDE rows = m >> log_r
  ColShare cs, zs;                        // trace / Z columns of this rank
  size_t cr = 0, zr = 0;                  // = cs.own(), zs.own()
  u32 planes = 1;                         // 2 from four ranks up: the rows i + 2 of the local rows arrive as a second plane
  u64 *lde_l = nullptr, *lde_n = nullptr, *zlde_l = nullptr, *zlde_n = nullptr, *scratch = nullptr;   // views of comm.recv_buf
  size_t scratch_words = 0;
  size_t slot_words = 0;                  // one send slot = [plane][dest][ob][ml]; two slots, used alternately by the steps
  u64* d_ldechunk = nullptr;              // [ntt_chunk][m]: one column block of this rank's LDE before it is packed
  u32* d_idx_local = nullptr;             // query leaf indices inside this rank's subtrees
  PairCols* d_pairs_own = nullptr;        // permutation pairs of the own Z columns, local order
  hipStream_t cstream = nullptr;          // the exchanges of the commit pipeline
  hipEvent_t xchg_done[MAX_CHUNKS];       // comm stream: the blocks of step k have arrived (and send slot k & 1 is free again)
  std::vector<hipEvent_t> tev;            // timing events around the exchanges (pairs), consumed in order
  size_t tev_used = 0;
};

struct sbn_prover {
  AirShape air; sbn_config cfg; FriShape fri;
  SplitCtx* sp = nullptr;                 // null: the whole proof on this GPU
  size_t lde_scratch_words = 0;           // capacity of d_lde as witness-generation scratch
  u32 degree_bits, lde_log; size_t n, m;
  int device; hipStream_t stream;
  // matrices
  u64 *d_trace = nullptr, *d_coef = nullptr, *d_lde = nullptr, *d_tmp = nullptr;
  u64 *d_zval = nullptr, *d_zcoef = nullptr, *d_zlde = nullptr;
  u64 *d_q = nullptr, *d_qlde = nullptr;
  DevTree tree_t, tree_z, tree_q;
  std::vector<DevTree> fri_trees;
  // tables
  u64 *d_tw_f = nullptr, *d_tw_i = nullptr, *d_shift = nullptr, *d_shift_inv = nullptr;
  u64 *d_shift_odd = nullptr;   // 2^19-point LDE (1,024 x 512): 7^i w_1024^(i >> 9), the input scale of the odd half of its split first pass
  u64 *d_xs = nullptr, *d_lag_first = nullptr, *d_lag_last = nullptr;
  u64 *d_apow = nullptr;  // [2][apow_n]
  size_t apow_n = 0;
  void* d_pic = nullptr;  // ExpPiConsts<F>
  PairCols* d_pairs = nullptr;
  // openings / FRI
  u64 *d_zpow = nullptr;        // 4 planes [n]: z^i (a,b), (g z)^i (a,b)
  u64 *d_open = nullptr;        // [(ncols + nzs + 4)][4]
  u64 *d_part = nullptr;        // 2 planes [groups][n]
  u64 *d_w = nullptr;           // group weights
  u64 *d_fa = nullptr, *d_fb = nullptr;    // F0 / F1 scratch planes [n] each (a,b) x2
  u64 *d_fcoef = nullptr;       // final poly coefficient planes [2][m]
  u64 *d_fcoef2 = nullptr;      // ping-pong for folding [2][m/2^arity]
  std::vector<u64*> fri_vals;   // per layer value planes [2][size]
  u64 *d_pow = nullptr;
  u32 *d_idx = nullptr;
  u64 *d_qbuf = nullptr; size_t qstride = 0;
  std::vector<u64> pi;
  bool loaded = false;
  hipEvent_t ev[ST_COUNT + 1];
  float stage_ms[ST_COUNT + EX_COUNT];
  size_t ntt_chunk;
  size_t ntt_sub = 0;                        // SBN_NTT_SUB: transform the columns of a chunk in sub-chunks of this many (0: whole chunk)
  bool ntt_xcd = true;                       // (columns, tiles) grid order in the fast passes; SBN_NTT_XCD=0: (tiles, columns)
  bool fast_ntt = true;                      // SBN_FAST_NTT=0 selects the generic radix-2 pass everywhere
  bool ntt_fused = false;                    // the inverse transform's pass B and the LDE's pass A as ONE kernel (2^16 / 2^17 rows)
  u64* d_tmp2 = nullptr;                     // its output: the fused kernel cannot work in place
  // EXPERIMENT (SBN_RANGE_ASYNC=1; measured, no gain: profiles/r4_range_async_ab.txt)
  // u16 range check of the curve witness BEHIND the call that generated it (n <= 2^16 rows): it writes the last columns of the trace
  // (start_lookups ..), which the trace commitment reaches after ~14 of its 27 chunks; it runs on its own stream, the commitment
  // joins it before the first chunk that holds such a column, every other reader of the trace joins it first (rc_finish), and its
  // error word (own allocation: the witness scratch lives in the LDE buffer the commitment overwrites) is checked by prove()
  hipStream_t rstream = nullptr;
  hipEvent_t rows_done = nullptr, rc_done = nullptr;
  int* d_rc_err = nullptr;
  bool rc_pending = false;
  size_t rc_first_col = 0;
  u64* d_tmp3 = nullptr;                     // 2^18 rows, two transform streams: the fused kernel's second output buffer (chunks alternate)
  bool ntt_fused512 = false;                 // 2^18-row tables: kernels_ntt.cuh ntt_fused512_inv_b_lde_a_kernel (SBN_NTT_FUSED=0: separate passes)
  hipStream_t hstream = nullptr;             // sponge absorption / Merkle stream
  hipStream_t nstream = nullptr;             // second transform stream (2^19 LDE rows and up): the LDE of chunk k beside the inverse transform of chunk k+1
  hipEvent_t intt_done[MAX_CHUNKS];          // main -> second transform stream: the coefficients of chunk k are complete
  bool ntt_two_streams = false;
  Settings set;                              // the SBN_* switches this prover was created under (settings.hpp)
  unsigned gen_calls = 0;                    // generate_trace calls so far (SBN_TRACEGEN_SKIP spares the first)
  int chain_mode = 0;                        // curve witness: 0 host pool, 1 one lane per instance, 2 one wave per instance
  hipEvent_t chunk_ready[MAX_CHUNKS];        // main -> hash: LDE chunk k is complete
  hipEvent_t abs_ev[2 * MAX_CHUNKS];         // hash stream: before/after each absorb launch
  hipEvent_t hash_done;                      // hash -> main
  u64* d_sponge = nullptr;                   // [12][m] sponge state carried between column chunks
  u64* h_chain = nullptr;                    // pinned staging for the host-computed curve chains (device tracegen)
  size_t h_chain_words = 0;
  u64* h_io = nullptr;                       // pinned staging of the device witness: the instance list in, the outputs + error word back
  size_t h_io_words = 0;
  u64* h_open = nullptr;                     // pinned landing buffer of the opened values [(ncols + nzs + 4)][4]
  u64* h_open2 = nullptr;                    // second landing buffer: the values at g*zeta of the trace and Z columns (the host is still reading the first)
};

static int dmalloc(u64** p, size_t words) {
  HIPC(hipMalloc((void**)p, words * sizeof(u64)));
  return 0;
}

// ---- NTT driver -----------------------------------------------------------------------------------
static u32 pick_log_t(u32 log_r, u32 log_s) {
  u32 lt = 4;
  while (lt > 0 && ((size_t)(1u << log_r) * ((1u << lt) + 1) * 8) > 48 * 1024) lt--;
  return std::min(lt, log_s);
}
// One full n-point transform per column (two passes through `tmp`).
static int ntt_columns(sbn_prover* P, const u64* in, size_t in_cs, u64* out, size_t out_cs, u64* tmp, size_t tmp_cs, size_t ncols,
                       u32 log_n, bool inverse, size_t n_in, const u64* pre, const u64* post, u64 scale, hipStream_t st = nullptr, int which = 3) {
  if (ncols == 0) return 0;                 // which: bit 0 = pass A (in -> tmp), bit 1 = pass B (tmp -> out)
  if (!st) st = P->stream;
  u32 log_n1 = (log_n + 1) / 2, log_n2 = log_n - log_n1;
  size_t n1 = (size_t)1 << log_n1, n2 = (size_t)1 << log_n2;
  NttPassParams a{};
  a.in = in; a.out = tmp; a.in_col_stride = in_cs; a.out_col_stride = tmp_cs;
  a.log_r = log_n1; a.log_t = pick_log_t(log_n1, log_n2); a.log_n = log_n;
  a.in_sr = n2; a.in_st = 1; a.out_sr = n2; a.out_st = 1; a.n_in = n_in;
  a.r_fast_load = 0; a.twiddle = 1; a.tw = inverse ? P->d_tw_i : P->d_tw_f; a.tw_log = P->lde_log;
  a.pre = pre; a.post = nullptr; a.scale = 1;
  NttPassParams b{};
  b.in = tmp; b.out = out; b.in_col_stride = tmp_cs; b.out_col_stride = out_cs;
  b.log_r = log_n2; b.log_t = pick_log_t(log_n2, log_n1); b.log_n = log_n;
  b.in_sr = 1; b.in_st = n2; b.out_sr = n1; b.out_st = 1; b.n_in = (size_t)1 << log_n;
  b.r_fast_load = 1; b.twiddle = 0; b.tw = a.tw; b.tw_log = P->lde_log;
  b.pre = nullptr; b.post = post; b.scale = scale;
  for (size_t c0 = 0; c0 < ncols; c0 += 32768) {
    size_t nc = std::min<size_t>(32768, ncols - c0);
    NttPassParams pa = a, pb = b;
    pa.in = a.in + c0 * in_cs; pa.out = a.out + c0 * tmp_cs;
    pb.in = b.in + c0 * tmp_cs; pb.out = b.out + c0 * out_cs;
    dim3 ga((unsigned)(n2 >> pa.log_t), (unsigned)nc), gb((unsigned)(n1 >> pb.log_t), (unsigned)nc);
    size_t la = ((size_t)1 << pa.log_r) * ((1u << pa.log_t) + 1) * 8, lb = ((size_t)1 << pb.log_r) * ((1u << pb.log_t) + 1) * 8;
    const u32 kperm = inverse ? 11u : 5u;  // omega_16 = (2^12)^13, omega_16^-1 = (2^12)^3: 13^-1 = 5, 3^-1 = 11 (mod 16)
    auto launch = [&](NttPassParams q, dim3 grid, size_t lds_bytes) {
      const bool fast = P->fast_ntt && q.log_t == 4 && (q.log_r == 8 || q.log_r == 9);
      if (fast && P->ntt_xcd) { q.xcd_order = 1; std::swap(grid.x, grid.y); }
      if (fast && q.log_r == 8) hipLaunchKernelGGL(ntt_fast_pass_kernel<0>, grid, dim3(256), 16 * 272 * 8, st, q, kperm);
      else if (fast) hipLaunchKernelGGL(ntt_fast_pass_kernel<1>, grid, dim3(256), 32 * 272 * 8, st, q, kperm);
      else hipLaunchKernelGGL(ntt_pass_kernel, grid, dim3(NTT_THREADS), lds_bytes, st, q);
    };
    if (P->fast_ntt && pa.log_r == 9 && log_n2 >= 4) pa.log_t = 4;  // the fast kernel always uses 16-wide tiles
    if (P->fast_ntt && pb.log_r == 9 && log_n1 >= 4) pb.log_t = 4;
    // 1,024-point first pass over an input whose rows 512.. are zero (the 2^19-point coset LDE): two 512-point fast passes
    // on grid.z (kernels_ntt.cuh NttPassParams::split) instead of the generic radix-2 pass
    const bool split_a = P->fast_ntt && !inverse && pa.log_r == 10 && log_n2 >= 4 && pa.in_st == 1 && n_in <= (size_t)512 * pa.in_sr && pre == P->d_shift && P->d_shift_odd;
    if (split_a) { pa.log_r = 9; pa.log_t = 4; pa.split = 1; pa.pre2 = P->d_shift_odd; }
    ga = dim3((unsigned)(n2 >> pa.log_t), (unsigned)nc, split_a ? 2u : 1u); gb = dim3((unsigned)(n1 >> pb.log_t), (unsigned)nc);
    la = ((size_t)1 << pa.log_r) * ((1u << pa.log_t) + 1) * 8; lb = ((size_t)1 << pb.log_r) * ((1u << pb.log_t) + 1) * 8;
    if (which & 1) launch(pa, ga, la);
    if (which & 2) launch(pb, gb, lb);
  }
  HIPC(hipGetLastError());
  return 0;
}

static u64 host_inv_pow2(u32 k) { return f_inv(F((u64)1 << k)).v; }
// the R = 512 fast pass needs 69,632 bytes of dynamic LDS (> the 64 KiB default)
// Function attributes are per DEVICE: one flag per device of the process (sbn_set_device may select another GPU later).
static constexpr int SBN_MAX_DEVICES = 64;
static int ntt_fast_setup() {  // idempotent, so a race between prover threads is harmless; the flag only saves the call
  static std::atomic<bool> done[SBN_MAX_DEVICES];
  const int d = g_device >= 0 && g_device < SBN_MAX_DEVICES ? g_device : 0;
  if (!done[d].load()) {
    HIPC(hipFuncSetAttribute((const void*)ntt_fast_pass_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8));
    HIPC(hipFuncSetAttribute((const void*)ntt_fused_inv_b_lde_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8));
    HIPC(hipFuncSetAttribute((const void*)ntt_fused512_inv_b_lde_a_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 272 * 8));
    done[d].store(true);
  }
  return 0;
}

// values [ncols][n] -> coefficients [ncols][n]
static int intt_values(sbn_prover* P, const u64* vals, u64* coef, size_t ncols) {
  for (size_t c0 = 0; c0 < ncols; c0 += P->ntt_chunk) {
    size_t nc = std::min(P->ntt_chunk, ncols - c0);
    int rc = ntt_columns(P, vals + c0 * P->n, P->n, coef + c0 * P->n, P->n, P->d_tmp, P->m, nc, P->degree_bits, true, P->n, nullptr, nullptr,
                         host_inv_pow2(P->degree_bits));
    if (rc) return rc;
  }
  return 0;
}
// coefficients [ncols][n] -> coset LDE [ncols][m]   (lde(rate_bits).coset_fft(7))
static int lde_coeffs(sbn_prover* P, const u64* coef, u64* lde, size_t ncols) {
  for (size_t c0 = 0; c0 < ncols; c0 += P->ntt_chunk) {
    size_t nc = std::min(P->ntt_chunk, ncols - c0);
    int rc = ntt_columns(P, coef + c0 * P->n, P->n, lde + c0 * P->m, P->m, P->d_tmp, P->m, nc, P->lde_log, false, P->n, P->d_shift, nullptr, 1);
    if (rc) return rc;
  }
  return 0;
}
// fused variant used for the big matrices: per chunk iNTT then LDE so the chunk stays cache resident
static int intt_then_lde(sbn_prover* P, const u64* vals, u64* coef, u64* lde, size_t ncols) {
  for (size_t c0 = 0; c0 < ncols; c0 += P->ntt_chunk) {
    size_t nc = std::min(P->ntt_chunk, ncols - c0);
    int rc = ntt_columns(P, vals + c0 * P->n, P->n, coef + c0 * P->n, P->n, P->d_tmp, P->m, nc, P->degree_bits, true, P->n, nullptr, nullptr,
                         host_inv_pow2(P->degree_bits));
    if (rc) return rc;
    rc = ntt_columns(P, coef + c0 * P->n, P->n, lde + c0 * P->m, P->m, P->d_tmp, P->m, nc, P->lde_log, false, P->n, P->d_shift, nullptr, 1);
    if (rc) return rc;
  }
  return 0;
}

// The fused middle pass: inverse pass A -> d_tmp, [inverse pass B + LDE pass A] -> coefficients and `tmp2` on the main stream
// (kernels_ntt.cuh ntt_fused_inv_b_lde_a_kernel at 2^16 / 2^17 rows; ntt_fused512_inv_b_lde_a_kernel at 2^18 rows, where the LDE's
// pass A is the 1,024-point one in two halves); the LDE's pass B follows on `lde_stream` (the main stream, or the second transform
// stream behind `handoff`).  Three launches per chunk instead of four / five.
static int intt_lde_cols_fused(sbn_prover* P, const u64* v, u64* cf, u64* lde_out, size_t nc, u64* tmp2, hipStream_t lde_stream, hipEvent_t handoff) {
  int rc = ntt_columns(P, v, P->n, cf, P->n, P->d_tmp, P->m, nc, P->degree_bits, true, P->n, nullptr, nullptr, host_inv_pow2(P->degree_bits), nullptr, 1);
  if (rc) return rc;
  NttFusedParams f{};
  const u32 log_n1 = (P->degree_bits + 1) / 2;
  f.in = P->d_tmp; f.in_col_stride = P->m; f.coef = cf; f.coef_col_stride = P->n; f.out = tmp2; f.out_col_stride = P->m;
  f.n1 = (size_t)1 << log_n1; f.log_n = P->degree_bits; f.tw_i = P->d_tw_i; f.tw_f = P->d_tw_f; f.tw_log = P->lde_log; f.pre = P->d_shift; f.pre2 = P->d_shift_odd;
  f.scale = host_inv_pow2(P->degree_bits);
  if (P->ntt_fused512) hipLaunchKernelGGL(ntt_fused512_inv_b_lde_a_kernel, dim3((unsigned)nc, 32u), dim3(256), 32 * 272 * 8, P->stream, f);
  else hipLaunchKernelGGL(ntt_fused_inv_b_lde_a_kernel, dim3((unsigned)nc, (unsigned)(f.n1 >> 4)), dim3(256), 32 * 272 * 8, P->stream, f);
  if (lde_stream != P->stream) {
    HIPC(hipEventRecord(handoff, P->stream));
    HIPC(hipStreamWaitEvent(lde_stream, handoff, 0));
  }
  return ntt_columns(P, cf, P->n, lde_out, P->m, tmp2, P->m, nc, P->lde_log, false, P->n, P->d_shift, nullptr, 1, lde_stream, 2);
}
// values -> coefficients -> coset LDE of nc columns (column strides n, n, m): the unit of the commit pipelines
static int intt_lde_cols(sbn_prover* P, const u64* v, u64* cf, u64* lde_out, size_t nc) {
  if ((P->ntt_fused || P->ntt_fused512) && P->d_tmp2) return intt_lde_cols_fused(P, v, cf, lde_out, nc, P->d_tmp2, P->stream, nullptr);
  int rc = ntt_columns(P, v, P->n, cf, P->n, P->d_tmp, P->m, nc, P->degree_bits, true, P->n, nullptr, nullptr, host_inv_pow2(P->degree_bits));
  if (rc) return rc;
  return ntt_columns(P, cf, P->n, lde_out, P->m, P->d_tmp, P->m, nc, P->lde_log, false, P->n, P->d_shift, nullptr, 1);
}
static int intt_then_lde_chunk(sbn_prover* P, const u64* vals, u64* coef, u64* lde, size_t c0, size_t nc) {
  if (!P->ntt_sub) return intt_lde_cols(P, vals + c0 * P->n, coef + c0 * P->n, lde + c0 * P->m, nc);
  const size_t sub = P->ntt_sub ? P->ntt_sub : nc;
  for (size_t s0 = 0; s0 < nc; s0 += sub) {   // (sub-chunks reuse the front of d_tmp: an experiment in L2 residency, SBN_NTT_SUB)
    const size_t a = c0 + s0, k = std::min(sub, nc - s0);
    int rc = ntt_columns(P, vals + a * P->n, P->n, coef + a * P->n, P->n, P->d_tmp, P->m, k, P->degree_bits, true, P->n, nullptr, nullptr,
                         host_inv_pow2(P->degree_bits));
    if (rc) return rc;
    rc = ntt_columns(P, coef + a * P->n, P->n, lde + a * P->m, P->m, P->d_tmp, P->m, k, P->lde_log, false, P->n, P->d_shift, nullptr, 1);
    if (rc) return rc;
  }
  return 0;
}

static int tree_alloc(DevTree& t, size_t nleaf, u32 cap_height) {
  t.nleaf = nleaf;
  u32 lg = 0; while (((size_t)1 << lg) < nleaf) lg++;
  t.nlevels = lg - cap_height;
  return dmalloc(&t.d, 2 * nleaf * 4);
}
static int tree_build_inner(sbn_prover* P, DevTree& t, hipStream_t st) {
  // One permutation costs 39 us on a lane and 16 us on 16 lanes, and a single wave saturates its SIMD.  So: levels with
  // >= 2^14 parents run one lane per parent over the whole GPU; narrower levels run 16 lanes per parent, 16 parents per
  // workgroup (one wave per SIMD), one launch per level; the last <= 16 parents finish inside one workgroup.
  // Round 3: the narrow levels (<= 8192 parents) run FIVE levels per launch -- a workgroup owns 32 consecutive nodes and hashes its
  // own 16 -> 8 -> 4 -> 2 -> 1 parents through LDS (merkle_subtree_kernel, 16 lanes per permutation), so a 2^17-leaf tree takes
  // 3 + 2 launches instead of 3 + 10 and the dependent chain loses eight launch gaps (SBN_MERKLE_FUSE=0: one launch per level, A/B).
  const bool fuse = P->set.merkle_fuse;
  u32 l0 = 0;
  while (l0 < t.nlevels) {
    const size_t parents = t.nleaf >> (l0 + 1);
    if (fuse && parents <= 8192 && parents >= 16 && ((t.nleaf >> l0) % 32) == 0) {
      const u32 nlev = std::min<u32>(t.nlevels - l0, 5);
      hipLaunchKernelGGL(merkle_subtree_kernel, dim3((unsigned)((t.nleaf >> l0) / 32)), dim3(MERKLE_SUBTREE_THREADS), 0, st, t.d, t.nleaf, l0, nlev, 32u);
      l0 += nlev;
      continue;
    }
    if (parents >= 16384) {
      hipLaunchKernelGGL(merkle_level_thread_kernel, dim3((unsigned)((parents + 255) / 256)), dim3(256), 0, st, t.d, t.nleaf, l0);
      l0++;
    } else if (parents > 16) {
      hipLaunchKernelGGL(merkle_level_coop_kernel, dim3((unsigned)((parents * 16 + 255) / 256)), dim3(256), 0, st, t.d, t.nleaf, l0);
      l0++;
    } else {
      const size_t nodes = t.nleaf >> l0;
      u32 nchild = (u32)std::min<size_t>(512, nodes);   // children per workgroup
      u32 lg = 0; while ((2u << lg) < nchild) lg++;       // nchild = 2^(lg+1)
      u32 nlev = std::min(t.nlevels - l0, lg + 1);
      hipLaunchKernelGGL(merkle_subtree_kernel, dim3((unsigned)(nodes / nchild)), dim3(MERKLE_SUBTREE_THREADS), 0, st, t.d, t.nleaf, l0, nlev, nchild);
      l0 += nlev;
    }
  }
  HIPC(hipGetLastError());
  return 0;
}
// narrow matrices (quotient chunks): single-launch leaf hash + tree on the main stream
static int tree_from_matrix(sbn_prover* P, DevTree& t, const u64* lde, size_t ncols) {
  hipLaunchKernelGGL(leaf_hash_kernel, dim3((unsigned)((P->m + 255) / 256)), dim3(256), 0, P->stream, lde, P->m, P->lde_log, (u32)ncols, t.d);
  return tree_build_inner(P, t, P->stream);
}
// PolynomialBatch::from_values for a wide matrix: per column chunk iNTT + coset LDE on the main stream,
// sponge absorption of that chunk on the hash stream, then the Merkle levels; main waits at the end.
static int intt_then_lde_chunk(sbn_prover* P, const u64* vals, u64* coef, u64* lde, size_t c0, size_t nc);
static int commit_pipeline(sbn_prover* P, const u64* vals, u64* coef, u64* lde, size_t ncols, DevTree& t, int ex_ms, int ex_launches) {
  size_t ch = P->ntt_chunk;
  size_t nchunks = (ncols + ch - 1) / ch;
  if (nchunks > (size_t)MAX_CHUNKS) return fail(SBN_ERR_UNSUPPORTED, "too many column chunks");
  if (ncols <= 4) {   // hash_or_noop: a leaf of at most 4 elements is its own digest (MyStark's 4 columns and its 2 Z columns)
    if (P->rc_pending && vals == P->d_trace) HIPC(hipStreamWaitEvent(P->stream, P->rc_done, 0));
    int rc = intt_then_lde(P, vals, coef, lde, ncols);
    if (rc) return rc;
    P->stage_ms[ST_COUNT + ex_launches] = 0;
    return tree_from_matrix(P, t, lde, ncols);
  }
  // Two streams: iNTT + coset LDE of chunk k on the main stream, the sponge over the chunk's LDE on the hash stream.
  // MEASURED and dropped (profiles/r2_bench_3stream.json): a third stream for the LDE, so that the iNTT of chunk k+1 hides
  // behind the LDE of chunk k -- 28.3 -> 29.6 ms per proof: the stage is bound by the VALU work of sponge + transforms
  // together, not by the latency of the transform stream, and more transform waves in flight only slow the sponge launches
  // (12.3 -> 14.8 ms of sponge kernel time).
  bool rc_joined = !(P->rc_pending && vals == P->d_trace);
  for (size_t k = 0; k < nchunks; k++) {
    size_t c0 = k * ch, nc = std::min(ch, ncols - c0);
    int rc;
    if (!rc_joined && c0 + nc > P->rc_first_col) {   // the first chunk with a column the asynchronous range check writes
      HIPC(hipStreamWaitEvent(P->stream, P->rc_done, 0));
      rc_joined = true;
    }
    if (P->ntt_two_streams && !P->ntt_sub && (P->ntt_fused || P->ntt_fused512) && P->d_tmp3) {
      // the fused kernel of chunk k writes buffer k & 1, which the LDE pass B of chunk k - 2 (second stream) must have left
      if (k >= 2) HIPC(hipStreamWaitEvent(P->stream, P->chunk_ready[k - 2], 0));
      rc = intt_lde_cols_fused(P, vals + c0 * P->n, coef + c0 * P->n, lde + c0 * P->m, nc, (k & 1) ? P->d_tmp3 : P->d_tmp2, P->nstream, P->intt_done[k]);
      if (rc) return rc;
      HIPC(hipEventRecord(P->chunk_ready[k], P->nstream));
    } else if (P->ntt_two_streams && !P->ntt_sub && !P->ntt_fused && !P->ntt_fused512) {
      rc = ntt_columns(P, vals + c0 * P->n, P->n, coef + c0 * P->n, P->n, P->d_tmp, P->m, nc, P->degree_bits, true, P->n, nullptr, nullptr, host_inv_pow2(P->degree_bits));
      if (rc) return rc;
      HIPC(hipEventRecord(P->intt_done[k], P->stream));
      HIPC(hipStreamWaitEvent(P->nstream, P->intt_done[k], 0));
      rc = ntt_columns(P, coef + c0 * P->n, P->n, lde + c0 * P->m, P->m, P->d_tmp2, P->m, nc, P->lde_log, false, P->n, P->d_shift, nullptr, 1, P->nstream);
      if (rc) return rc;
      HIPC(hipEventRecord(P->chunk_ready[k], P->nstream));
    } else {
      rc = intt_then_lde_chunk(P, vals, coef, lde, c0, nc);
      if (rc) return rc;
      HIPC(hipEventRecord(P->chunk_ready[k], P->stream));
    }
    HIPC(hipStreamWaitEvent(P->hstream, P->chunk_ready[k], 0));
    HIPC(hipEventRecord(P->abs_ev[2 * k], P->hstream));
    hipLaunchKernelGGL(leaf_absorb_kernel, dim3((unsigned)((P->m + 255) / 256)), dim3(256), 0, P->hstream, lde + c0 * P->m, P->m, P->lde_log, (u32)nc,
                       P->d_sponge, k == 0 ? 1 : 0, k + 1 == nchunks ? 1 : 0, t.d,
                       (k + 1 < nchunks && std::min(ch, ncols - (c0 + nc)) >= 8) ? 1 : 0);
    HIPC(hipEventRecord(P->abs_ev[2 * k + 1], P->hstream));
  }
  int rc = tree_build_inner(P, t, P->hstream);
  if (rc) return rc;
  HIPC(hipEventRecord(P->hash_done, P->hstream));
  HIPC(hipStreamWaitEvent(P->stream, P->hash_done, 0));
  P->stage_ms[ST_COUNT + ex_launches] = (float)nchunks;
  (void)ex_ms;
  return 0;
}
// Host waits inside prove().  Polling hipStreamQuery / hipEventQuery instead was measured: no gain over the runtime's waits.
// A rank of a split proof (world > 1) waits with a DEADLINE instead: a peer that died leaves this rank's stream behind a
// receive that never completes, and hipStreamSynchronize would hang the whole job; polling returns hipErrorNotReady after
// SBN_COMM_TIMEOUT_S seconds (default 600), prove() fails with SBN_ERR_HIP and the caller can tear the job down.
static thread_local double t_wait_deadline_s = 0;   // 0: plain runtime waits
// (the ranks of a local group are threads that may outnumber the CPUs of the box: spin briefly, then sleep between polls)
static inline void poll_backoff(unsigned spins) {
  if (spins < 2000) std::this_thread::yield();
  else std::this_thread::sleep_for(std::chrono::microseconds(spins < 20000 ? 20 : 200));
}
static inline hipError_t stream_wait(hipStream_t st) {
  if (t_wait_deadline_s <= 0) return hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t e = hipStreamQuery(st);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > t_wait_deadline_s) return hipErrorNotReady;
    poll_backoff(++spins);
  }
}
static inline hipError_t event_wait(hipEvent_t ev) {
  if (t_wait_deadline_s <= 0) return hipEventSynchronize(ev);
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t e = hipEventQuery(ev);
    if (e != hipErrorNotReady) return e;
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > t_wait_deadline_s) return hipErrorNotReady;
    poll_backoff(++spins);
  }
}
// ---- oversized-trace split: exchange helpers ------------------------------------------------------------------------
// Every exchange is STREAM-ORDERED (include/sbn.h sbn_comm.all_to_all): it is enqueued on `st` behind the kernels that
// packed its blocks, and the kernels that consume the received blocks are enqueued on `st` (or behind an event of `st`)
// after it.  Timing: a pair of events around each call, summed after the proof (split_exchange_ms).
static int split_exchange(sbn_prover* P, hipStream_t st, const std::vector<uint64_t>& so, const std::vector<uint64_t>& sl, const std::vector<uint64_t>& ro,
                          const std::vector<uint64_t>& rl) {
  SplitCtx* S = P->sp;
  bool any = false;
  for (u32 r = 0; r < S->comm.world; r++) any = any || sl[r] || rl[r];
  if (!any) return 0;   // (every rank sees the same emptiness: the block counts are functions of the shape alone)
  const bool timed = S->tev_used + 2 <= S->tev.size();
  if (timed) HIPC(hipEventRecord(S->tev[S->tev_used], st));
  const int rc = S->comm.all_to_all(S->comm.ctx, (void*)st, so.data(), sl.data(), ro.data(), rl.data());
  if (rc) return fail(SBN_ERR_HIP, "sbn_comm.all_to_all failed (%d)", rc);
  if (timed) { HIPC(hipEventRecord(S->tev[S->tev_used + 1], st)); S->tev_used += 2; }
  return 0;
}
static int split_all_gather_host(sbn_prover* P, const void* send, void* recv, size_t bytes) {
  SplitCtx* S = P->sp;
  const int rc = S->comm.all_gather_host(S->comm.ctx, send, recv, bytes);
  if (rc) return fail(SBN_ERR_HIP, "sbn_comm.all_gather_host failed (%d)", rc);
  return 0;
}
// every rank contributes `words` u64 at the start of the send buffer; the result [world][words] lands in the receive
// scratch; ordered on the main stream
static int split_all_gather_device(sbn_prover* P, size_t words) {
  SplitCtx* S = P->sp;
  const u32 R = S->comm.world;
  if (words * R > S->scratch_words || words * sizeof(u64) > S->comm.send_bytes) return fail(SBN_ERR_HIP, "internal: split scratch too small");
  const uint64_t base = (uint64_t)((const char*)S->scratch - (const char*)S->comm.recv_buf);
  std::vector<uint64_t> so(R, 0), sl(R, words * sizeof(u64)), ro(R), rl(R, words * sizeof(u64));
  for (u32 r = 0; r < R; r++) ro[r] = base + (uint64_t)r * words * sizeof(u64);
  return split_exchange(P, P->stream, so, sl, ro, rl);
}
// PolynomialBatch::from_values for this rank's share, as a three-stage pipeline over the column blocks (ColShare):
//   main stream   step k: iNTT + coset LDE of the own block k (coefficients stay here), rows packed per destination
//   comm stream   step k: one all-to-all per plane, columns -> rows (the own rows never travel: the pack kernel writes them
//                 straight into the receive matrix)
//   hash stream   step k: the leaf sponge absorbs the blocks k*R .. k*R+R-1 of its LDE rows; then the Merkle subtrees.
// With one rank the LDE lands in the row matrix directly (local row = natural row) and nothing is exchanged.
// `vals`: [..][n] values; block b of the matrix starts at vals + (vals_global ? b : own index k) * ob * n.
static int commit_split(sbn_prover* P, const ColShare& sh, const u64* vals, bool vals_global, u64* coef_own, u64* plane_l, u64* plane_n, DevTree& t) {
  SplitCtx* S = P->sp;
  const u32 R = S->comm.world, me = S->comm.rank;
  const size_t n = P->n, m = P->m, ml = S->ml, ob = sh.ob;
  const size_t steps = sh.steps();
  if (steps > (size_t)MAX_CHUNKS) return fail(SBN_ERR_UNSUPPORTED, "too many column blocks");
  const char* rbase = (const char*)S->comm.recv_buf;
  const char* sbase = (const char*)S->comm.send_buf;
  const u32 row_log = P->lde_log - S->log_r;
  for (size_t k = 0; k < steps; k++) {
    const size_t b = k * R + me, nc = sh.block_cols(b);
    hipStream_t tstream = P->stream;                                        // the stream that completes block k's LDE (and pack)
    u64* const slot = (u64*)S->comm.send_buf + (k & 1) * S->slot_words;     // [plane][dest][ob][ml]
    u64* const slot_n = S->planes == 2 ? slot + (size_t)R * ob * ml : nullptr;
    if (nc) {
      const u64* v = vals + (vals_global ? b : k) * ob * n;
      u64* cf = coef_own + k * ob * n;
      u64* lde_out = R == 1 ? plane_l + b * ob * m : S->d_ldechunk;
      // two transform streams from 2^19 LDE rows up (as in commit_pipeline): the LDE (and the pack) of block k beside the inverse
      // transform of block k + 1
      const bool fused = P->ntt_fused || P->ntt_fused512;
      const bool two = P->ntt_two_streams && P->d_tmp2 && (!fused || P->d_tmp3);
      tstream = two ? P->nstream : P->stream;
      int rc;
      if (two && fused) {   // (as in commit_pipeline: the fused kernel's output buffers alternate)
        if (k >= 2) HIPC(hipStreamWaitEvent(P->stream, P->chunk_ready[k - 2], 0));
        rc = intt_lde_cols_fused(P, v, cf, lde_out, nc, (k & 1) ? P->d_tmp3 : P->d_tmp2, P->nstream, P->intt_done[k]);
      } else if (two) {
        rc = ntt_columns(P, v, n, cf, n, P->d_tmp, m, nc, P->degree_bits, true, n, nullptr, nullptr, host_inv_pow2(P->degree_bits));
        if (rc) return rc;
        HIPC(hipEventRecord(P->intt_done[k], P->stream));
        HIPC(hipStreamWaitEvent(P->nstream, P->intt_done[k], 0));
        rc = ntt_columns(P, cf, n, lde_out, m, P->d_tmp2, m, nc, P->lde_log, false, n, P->d_shift, nullptr, 1, P->nstream);
      } else rc = intt_lde_cols(P, v, cf, lde_out, nc);
      if (rc) return rc;
      if (R > 1 && k >= 2) HIPC(hipStreamWaitEvent(tstream, S->xchg_done[k - 2], 0));   // send slot k & 1 has left
      if (R > 1)
        hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)nc), dim3(256), 0, tstream, S->d_ldechunk, m, (u32)nc, (u32)ob, S->log_r, me,
                           slot, slot_n, plane_l + b * ob * ml, plane_n ? plane_n + b * ob * ml : nullptr);
      HIPC(hipGetLastError());
    }
    HIPC(hipEventRecord(P->chunk_ready[k], tstream));
    hipEvent_t arrived = P->chunk_ready[k];
    if (R > 1) {
      HIPC(hipStreamWaitEvent(S->cstream, P->chunk_ready[k], 0));
      for (u32 pl = 0; pl < S->planes; pl++) {
        const u64* dst = pl ? plane_n : plane_l;
        std::vector<uint64_t> so(R, 0), sl(R, 0), ro(R, 0), rl(R, 0);
        for (u32 r = 0; r < R; r++) {
          if (r == me) continue;
          so[r] = (uint64_t)((const char*)(slot + ((size_t)pl * R + r) * ob * ml) - sbase);
          sl[r] = (uint64_t)(nc * ml * sizeof(u64));
          const size_t br = k * R + r;
          ro[r] = (uint64_t)((const char*)(dst + br * ob * ml) - rbase);
          rl[r] = (uint64_t)(sh.block_cols(br) * ml * sizeof(u64));
        }
        int rc = split_exchange(P, S->cstream, so, sl, ro, rl);
        if (rc) return rc;
      }
      HIPC(hipEventRecord(S->xchg_done[k], S->cstream));
      arrived = S->xchg_done[k];
    }
    HIPC(hipStreamWaitEvent(P->hstream, arrived, 0));
    // the sponge over the blocks of this step, in launches of one block (64 columns: the carried state stays small in L2)
    const size_t c_begin = k * R * ob, c_end = std::min(sh.total, (k + 1) * R * ob);
    for (size_t c0 = c_begin; c0 < c_end; c0 += ob) {
      const size_t ncs = std::min(ob, c_end - c0);
      hipLaunchKernelGGL(leaf_absorb_kernel, dim3((unsigned)((ml + 255) / 256)), dim3(256), 0, P->hstream, plane_l + c0 * ml, ml, row_log, (u32)ncs,
                         P->d_sponge, c0 == 0 ? 1 : 0, c0 + ncs == sh.total ? 1 : 0, t.d,
                         (c0 + ncs < sh.total && sh.total - (c0 + ncs) >= 8) ? 1 : 0);
    }
    HIPC(hipGetLastError());
  }
  int rc = tree_build_inner(P, t, P->hstream);
  if (rc) return rc;
  HIPC(hipEventRecord(P->hash_done, P->hstream));
  HIPC(hipStreamWaitEvent(P->stream, P->hash_done, 0));
  return 0;
}
// the Merkle cap: this rank's 2^(cap_height - log R) subtree roots, all-gathered in rank order = cap order
static int split_cap_to_host(sbn_prover* P, const DevTree& t, std::vector<u64>& cap) {
  SplitCtx* S = P->sp;
  const size_t capn = (size_t)1 << P->cfg.cap_height, own = capn >> S->log_r;
  std::vector<u64> mine(own * 4);
  HIPC(hipMemcpyAsync(mine.data(), t.level(t.nlevels), own * 4 * sizeof(u64), hipMemcpyDeviceToHost, P->stream));
  HIPC(stream_wait(P->stream));
  cap.resize(capn * 4);
  return split_all_gather_host(P, mine.data(), cap.data(), own * 4 * sizeof(u64));
}
static int absorb_times(sbn_prover* P, size_t ncols, int ex_ms) {
  size_t nchunks = (ncols + P->ntt_chunk - 1) / P->ntt_chunk;
  float tot = 0;
  for (size_t k = 0; k < nchunks; k++) { float ms = 0; HIPC(hipEventElapsedTime(&ms, P->abs_ev[2 * k], P->abs_ev[2 * k + 1])); tot += ms; }
  P->stage_ms[ST_COUNT + ex_ms] = tot;
  return 0;
}
static int tree_cap_to_host(sbn_prover* P, const DevTree& t, std::vector<u64>& cap) {
  size_t capn = (size_t)1 << P->cfg.cap_height;
  cap.resize(capn * 4);
  HIPC(hipMemcpyAsync(cap.data(), t.level(t.nlevels), capn * 4 * sizeof(u64), hipMemcpyDeviceToHost, P->stream));
  HIPC(stream_wait(P->stream));
  return 0;
}

// ---- create / destroy -----------------------------------------------------------------------------
// Send side: two slots of one column block per plane, [plane][dest][ob][m/R] = planes * ob * m words each (one rank: no
// slot, the LDE lands in the row matrix), and room for the gathered vectors (2 m/R quotient values, 2 n partial sums).
// Receive side: the row-sharded trace and Z matrices per plane plus the gather scratch.
static constexpr size_t SPLIT_BLOCK = 64;   // columns per ownership block = the prover's NTT chunk (create_ctx checks)
static int split_sizes(const AirShape& as, u32 degree_bits, u32 rate_bits, u32 world, size_t ob, uint64_t* send_bytes, uint64_t* recv_bytes, size_t* scratch_words,
                       size_t* slot_words) {
  const size_t n = (size_t)1 << degree_bits, m = n << rate_bits;
  const u32 planes = world >= 4 ? 2 : 1;
  const size_t sw = std::max<size_t>(2 * m, 2 * n * world);   // gathered quotient values / FRI partial sums
  const size_t slot = world > 1 ? (size_t)planes * ob * m : 0;
  *send_bytes = (uint64_t)(std::max(std::max(2 * slot, 2 * (m / world)), 2 * n) * sizeof(u64));
  *recv_bytes = (uint64_t)(((as.ncols + as.nzs) * (m / world) * planes + sw) * sizeof(u64));
  *scratch_words = sw;
  if (slot_words) *slot_words = slot;
  return 0;
}
static int create_ctx(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, const sbn_comm* comm, sbn_prover** out);
extern "C" int sbn_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, sbn_prover** out) {
  return create_ctx(air, cfg, degree_bits, nullptr, out);
}
static int create_ctx(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, const sbn_comm* comm, sbn_prover** out) {
  if (!air || !cfg || !out) return fail(SBN_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (!config_supported(cfg)) return fail(SBN_ERR_UNSUPPORTED, "unsupported StarkConfig (need num_challenges=2, rate_bits=1)");
  AirShape as;
  if (!air_shape(air, cfg, as)) return fail(SBN_ERR_BAD_ARG, "unknown air kind / num_io");
  if (degree_bits < 9 || degree_bits > 22) return fail(SBN_ERR_UNSUPPORTED, "degree_bits out of range");
  if (as.kind == SBN_AIR_FLAGS && (512 * (size_t)as.num_io) != ((size_t)1 << degree_bits)) return fail(SBN_ERR_BAD_ARG, "FlagStark needs 512*num_io rows");
  if (as.kind == SBN_AIR_FLAGS_U64 && (128 * (size_t)as.num_io) != ((size_t)1 << degree_bits)) return fail(SBN_ERR_BAD_ARG, "the u64 FlagStark needs 128*num_io rows");
  if (is_exp_air(as.kind)) {
    if ((exp_rows_per_instance(as.kind) * as.num_io) != ((size_t)1 << degree_bits)) return fail(SBN_ERR_BAD_ARG, "the Exp tables need 512*num_io rows (FQ12_EXP_U64: 128*num_io)");
    if (as.kind != SBN_AIR_FQ12_EXP && as.kind != SBN_AIR_FQ12_EXP_U64 && degree_bits < 16)
      return fail(SBN_ERR_UNSUPPORTED, "G1_EXP / G2_EXP / FQ_EXP need >= 2^16 rows (u16 range check, range_check.rs:26)");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SBN_ERR_NO_DEVICE, "no HIP device available: the prover path has no CPU fallback");
  HIPC(hipSetDevice(g_device));
  sbn_prover* P = new sbn_prover();
  P->air = as; P->cfg = *cfg; P->degree_bits = degree_bits; P->lde_log = degree_bits + cfg->rate_bits;
  P->n = (size_t)1 << degree_bits; P->m = (size_t)1 << P->lde_log;
  P->fri = fri_shape(*cfg, degree_bits);
  P->device = g_device;
  // every failure below releases what was created so far (sbn_prover_destroy accepts a partly built context) and
  // returns the FIRST error code
  int rc = 0;
  auto acc = [&](int r) { if (!rc) rc = r; };
  auto hipc = [&](hipError_t e, const char* what) { if (e != hipSuccess && !rc) rc = fail(SBN_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e)); };
  if (P->fri.total_arity() > degree_bits + cfg->rate_bits - cfg->cap_height) { sbn_prover_destroy(P); return fail(SBN_ERR_UNSUPPORTED, "FRI total reduction arity is too large"); }
  // Column chunk of the commit pipeline: a multiple of 8 (the sponge permutes after every 8 absorbed columns, so any other
  // chunk boundary would change the leaf digests), between 8 and 256.
  // Default 64; 48 from 2^19 LDE rows up: a chunk's LDE (chunk * M * 8 bytes) is written by the transform stream and read by the
  // sponge out of the 256 MiB Infinity Cache, and 64 columns of 2^19 rows are 268 MB (Fq12ExpStark(512): 0.708 -> 0.675 s,
  // profiles/r3_v2_fq12_chunk.txt).
  // 2^18 LDE rows: 64 columns are 134 MB and the sponge launches themselves run 17 % slower (35.7 against 30.4 ms per trace commitment of
  // G1ExpStark(256)); 48 columns: prove() 61.1 -> 56.1 ms (profiles/r4_chunk_2pow17.txt).
  P->ntt_chunk = P->lde_log >= 18 ? 48 : 64;
  {
    std::string serr;
    if (!P->set.load(serr)) { sbn_prover_destroy(P); return fail(SBN_ERR_BAD_ARG, "%s", serr.c_str()); }
  }
  const Settings& set = P->set;
  if (set.ntt_chunk) P->ntt_chunk = (size_t)set.ntt_chunk;
  P->fast_ntt = set.fast_ntt; P->ntt_xcd = set.ntt_xcd;
  P->ntt_fused = P->fast_ntt && (degree_bits == 16 || degree_bits == 17) && set.ntt_fused;   // SBN_NTT_FUSED=0: four separate passes (A/B)
  P->ntt_fused512 = P->fast_ntt && degree_bits == 18 && P->lde_log == 19 && set.ntt_split1024 && set.ntt_fused;
  // with the fused middle pass the chunk also crosses the transform's second buffer: 40 columns (0.605 -> 0.589 s, profiles/r4_fused512.txt)
  if (P->ntt_fused512 && !set.ntt_chunk) P->ntt_chunk = 40;
  P->ntt_sub = (size_t)set.ntt_sub;
  // curve chains of the device witness: the host pool when the CPU has AVX-512 IFMA (eight instances per register: 128 instances
  // are 16 tasks of ~0.15 ms, chains_ifma.hpp) or the pool has the threads for the scalar form (2.2 ms on 16), else one wave per
  // instance on the device (5.2 ms whatever the host share: an 8-rank node may leave a rank two CPUs, where the scalar host chains
  // take ~18 ms)
  P->chain_mode = set.device_chain >= 0 ? set.device_chain : ((tracegen_host_chains_vectorized() || tracegen_host_threads() >= 8) ? 0 : 2);
  acc(ntt_fast_setup());
  hipc(hipStreamCreate(&P->stream), "hipStreamCreate");
  hipc(hipStreamCreate(&P->hstream), "hipStreamCreate");
  if (is_exp_air(as.kind)) {   // the asynchronous range check of the curve witness (sbn_prover::rstream)
    hipc(hipStreamCreate(&P->rstream), "hipStreamCreate");
    hipc(hipEventCreateWithFlags(&P->rows_done, hipEventDisableTiming), "hipEventCreate");
    hipc(hipEventCreateWithFlags(&P->rc_done, hipEventDisableTiming), "hipEventCreate");
    hipc(hipMalloc((void**)&P->d_rc_err, 256), "hipMalloc");
  }
  // From 2^19 LDE rows up the transform stream, not the sum of the instruction streams, bounds the commitments (profiles/
  // r3_v8_fq12_512_kernel_stats.csv: five passes of 410 us per chunk beside a 1.19 ms sponge launch, 1.63 ms of VALU work in a 2.05 ms
  // period): the LDE passes of chunk k then run on a second stream beside the inverse passes of chunk k+1 (SBN_NTT_STREAMS=1: one
  // stream, =2: two streams at any size -- at 2^16 rows, where the stage IS at its VALU bound, this was measured slower in round 2).
  P->ntt_two_streams = set.ntt_streams ? set.ntt_streams == 2 : P->lde_log >= 19;
  if (P->ntt_two_streams) {
    hipc(hipStreamCreate(&P->nstream), "hipStreamCreate");
    for (auto& e : P->intt_done) hipc(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
  }
  for (auto& e : P->ev) hipc(hipEventCreate(&e), "hipEventCreate");
  for (auto& e : P->abs_ev) hipc(hipEventCreate(&e), "hipEventCreate");
  for (auto& e : P->chunk_ready) hipc(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
  hipc(hipEventCreateWithFlags(&P->hash_done, hipEventDisableTiming), "hipEventCreate");
  for (auto& v : P->stage_ms) v = 0;
  if (rc) { sbn_prover_destroy(P); return rc; }
  const size_t n = P->n, m = P->m, C = as.ncols, Z = as.nzs;
  if (comm && (C <= 4 || Z == 0)) { sbn_prover_destroy(P); return fail(SBN_ERR_UNSUPPORTED, "the split prover covers the wide tables (this one has %zu columns and %zu permutation Zs)", C, Z); }
  if (comm) {
    // one trace over comm->world GPUs: this rank keeps the coefficients of its own columns and the LDE ROWS of its Merkle
    // subtrees (views of the caller's receive buffer); the trace values stay whole
    const u32 R = comm->world;
    u32 lr = 0; while ((1u << lr) < R) lr++;
    uint64_t sb = 0, rb = 0; size_t sw = 0, slotw = 0;
    if (comm->struct_size != sizeof(sbn_comm)) { sbn_prover_destroy(P); return fail(SBN_ERR_BAD_ARG, "sbn_comm.struct_size does not match this library (ABI %d)", SBN_ABI_VERSION); }
    if (!set.ntt_chunk) P->ntt_chunk = SPLIT_BLOCK;   // (the size-dependent default above is the single-GPU pipeline's: a rank's rows are M / world)
    if (P->ntt_chunk != SPLIT_BLOCK) { sbn_prover_destroy(P); return fail(SBN_ERR_UNSUPPORTED, "the split prover deals columns in blocks of %zu: unset SBN_NTT_CHUNK", SPLIT_BLOCK); }
    split_sizes(as, degree_bits, cfg->rate_bits, R, SPLIT_BLOCK, &sb, &rb, &sw, &slotw);
    if ((1u << lr) != R || lr > cfg->cap_height || comm->rank >= R || !comm->all_to_all || !comm->all_gather_host || !comm->send_buf || !comm->recv_buf ||
        comm->send_bytes < sb || comm->recv_bytes < rb) {
      sbn_prover_destroy(P);
      return fail(SBN_ERR_BAD_ARG, "bad sbn_comm (world must be a power of two <= 2^cap_height; staging buffers of sbn_split_exchange_bytes)");
    }
    for (const void* buf : {(const void*)comm->send_buf, (const void*)comm->recv_buf}) {   // (a transport created on another GPU: ADVICE r3)
      hipPointerAttribute_t pa{};
      if (hipPointerGetAttributes(&pa, buf) != hipSuccess || pa.device != P->device) {
        (void)hipGetLastError();
        sbn_prover_destroy(P);
        return fail(SBN_ERR_BAD_ARG, "sbn_comm staging buffers do not live on this prover's device %d (call sbn_set_device before creating the transport)", g_device);
      }
    }
    SplitCtx* S = P->sp = new SplitCtx();
    for (auto& e : S->xchg_done) e = nullptr;
    S->comm = *comm; S->log_r = lr; S->rho = bitrev32(comm->rank, lr); S->ml = m >> lr; S->planes = R >= 4 ? 2 : 1;
    S->slot_words = slotw;
    S->cs.total = C; S->cs.ob = SPLIT_BLOCK; S->cs.R = R; S->cs.rank = comm->rank;
    S->zs = S->cs; S->zs.total = Z;
    S->cr = S->cs.own(); S->zr = S->zs.own();
    hipc(hipStreamCreate(&S->cstream), "hipStreamCreate");
    for (auto& e : S->xchg_done) hipc(hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    S->tev.assign(2 * (2 * (S->cs.steps() + S->zs.steps()) + 8), nullptr);
    for (auto& e : S->tev) hipc(hipEventCreate(&e), "hipEventCreate");
    u64* rbuf = (u64*)comm->recv_buf;
    S->lde_l = rbuf; rbuf += C * S->ml;
    if (S->planes == 2) { S->lde_n = rbuf; rbuf += C * S->ml; }
    S->zlde_l = rbuf; rbuf += Z * S->ml;
    if (S->planes == 2) { S->zlde_n = rbuf; rbuf += Z * S->ml; }
    S->scratch = rbuf; S->scratch_words = sw;
    P->lde_scratch_words = (size_t)(rb / sizeof(u64));   // witness generation runs before the first exchange
    acc(dmalloc(&P->d_trace, C * n)); acc(dmalloc(&P->d_coef, std::max<size_t>(S->cr, 1) * n));
    acc(dmalloc(&P->d_zval, std::max<size_t>(S->zr, 1) * n)); acc(dmalloc(&P->d_zcoef, std::max<size_t>(S->zr, 1) * n));
    acc(dmalloc(&S->d_ldechunk, P->ntt_chunk * m));
    hipc(hipMalloc((void**)&S->d_idx_local, cfg->num_query_rounds * sizeof(u32)), "hipMalloc");
    acc(tree_alloc(P->tree_t, S->ml, cfg->cap_height - lr)); acc(tree_alloc(P->tree_z, S->ml, cfg->cap_height - lr));
  } else {
    P->lde_scratch_words = C * m;
    acc(dmalloc(&P->d_trace, C * n)); acc(dmalloc(&P->d_coef, C * n)); acc(dmalloc(&P->d_lde, C * m));
    acc(dmalloc(&P->d_zval, std::max<size_t>(Z, 1) * n)); acc(dmalloc(&P->d_zcoef, std::max<size_t>(Z, 1) * n)); acc(dmalloc(&P->d_zlde, std::max<size_t>(Z, 1) * m));
    acc(tree_alloc(P->tree_t, m, cfg->cap_height)); acc(tree_alloc(P->tree_z, m, cfg->cap_height));
  }
  acc(dmalloc(&P->d_tmp, std::max(P->ntt_chunk, (size_t)4) * m));
  if (P->ntt_fused || P->ntt_fused512 || P->ntt_two_streams) acc(dmalloc(&P->d_tmp2, std::max(P->ntt_chunk, (size_t)4) * m));
  if ((P->ntt_fused || P->ntt_fused512) && P->ntt_two_streams) acc(dmalloc(&P->d_tmp3, std::max(P->ntt_chunk, (size_t)4) * m));
  acc(dmalloc(&P->d_q, 2 * m)); acc(dmalloc(&P->d_qlde, 4 * m));
  acc(tree_alloc(P->tree_q, m, cfg->cap_height));
  acc(dmalloc(&P->d_tw_f, m)); acc(dmalloc(&P->d_tw_i, m)); acc(dmalloc(&P->d_shift, m)); acc(dmalloc(&P->d_shift_inv, m));
  if (P->lde_log == 19 && set.ntt_split1024) acc(dmalloc(&P->d_shift_odd, n));   // SBN_NTT_SPLIT1024=0: generic first pass (A/B)
  acc(dmalloc(&P->d_xs, m)); acc(dmalloc(&P->d_lag_first, m)); acc(dmalloc(&P->d_lag_last, m));
  P->apow_n = apow_len(as.nconstraints, as.nzs);
  acc(dmalloc(&P->d_apow, (size_t)SBN_NCH * P->apow_n));
  acc(dmalloc(&P->d_zpow, 4 * n)); acc(dmalloc(&P->d_open, (C + Z + 4) * 4));
  hipc(hipHostMalloc((void**)&P->h_open, (C + Z + 4) * 4 * sizeof(u64), hipHostMallocDefault), "hipHostMalloc");
  hipc(hipHostMalloc((void**)&P->h_open2, (C + Z) * 4 * sizeof(u64), hipHostMallocDefault), "hipHostMalloc");
  acc(dmalloc(&P->d_part, 2 * 32 * n)); acc(dmalloc(&P->d_w, 4096)); acc(dmalloc(&P->d_sponge, 12 * m));
  acc(dmalloc(&P->d_fa, 4 * n)); acc(dmalloc(&P->d_fcoef, 2 * m)); acc(dmalloc(&P->d_fcoef2, 2 * m));
  acc(dmalloc(&P->d_pow, 1));
  if (rc) { sbn_prover_destroy(P); return rc; }
  P->d_fb = P->d_fa + 2 * n;
  hipc(hipMalloc((void**)&P->d_pic, sizeof(ExpPiConsts<F>)), "hipMalloc");
  hipc(hipMalloc((void**)&P->d_idx, cfg->num_query_rounds * sizeof(u32)), "hipMalloc");
  // FRI layer buffers
  {
    u32 bits = P->lde_log;
    for (u32 ab : P->fri.arity_bits) {
      u64* v = nullptr; acc(dmalloc(&v, 2 * ((size_t)1 << bits))); P->fri_vals.push_back(v);
      DevTree t; acc(tree_alloc(t, (size_t)1 << (bits - ab), cfg->cap_height)); P->fri_trees.push_back(t);
      bits -= ab;
    }
    if (rc) { sbn_prover_destroy(P); return rc; }
  }
  // query section stride (words per query round)
  {
    size_t sib = (size_t)(P->lde_log - cfg->cap_height) * 4;
    size_t s = C + sib + (Z ? Z + sib : 0) + 4 + sib;
    u32 bits = P->lde_log;
    for (u32 ab : P->fri.arity_bits) { bits -= ab; s += 2 * ((size_t)1 << ab) + (size_t)(bits - cfg->cap_height) * 4; }
    P->qstride = s;
    acc(dmalloc(&P->d_qbuf, s * cfg->num_query_rounds));
    if (rc) { sbn_prover_destroy(P); return rc; }
  }
  // tables
  auto blocks = [](size_t k) { return dim3((unsigned)((k + 255) / 256)); };
  F w = f_root_of_unity(P->lde_log);
  // FULL root tables (w^i for every i < m: the second half is the negated first): the register passes index them without the
  // compare / negate of a half table (tw_full, kernels_ntt.cuh); everything else reads the first half only
  hipLaunchKernelGGL(pow_table_kernel, blocks(m), dim3(256), 0, P->stream, P->d_tw_f, m, w.v);
  hipLaunchKernelGGL(pow_table_kernel, blocks(m), dim3(256), 0, P->stream, P->d_tw_i, m, f_inv(w).v);
  hipLaunchKernelGGL(pow_table_kernel, blocks(m), dim3(256), 0, P->stream, P->d_shift, m, (u64)GL_GEN);
  hipLaunchKernelGGL(pow_table_kernel, blocks(m), dim3(256), 0, P->stream, P->d_shift_inv, m, f_inv(F(GL_GEN)).v);
  hipLaunchKernelGGL(domain_tables_kernel, blocks(m), dim3(256), 0, P->stream, P->d_xs, P->d_lag_first, P->d_lag_last, m, P->lde_log, degree_bits);
  if (P->d_shift_odd) hipLaunchKernelGGL(shift_odd_table_kernel, blocks(n), dim3(256), 0, P->stream, P->d_shift_odd, n, P->d_shift, P->d_tw_f, 9u);
  // permutation pairs
  {
    std::vector<PairCols> pairs(Z);
    for (size_t z = 0; z < Z; z++) {
      int l, r;
      if (as.kind == SBN_AIR_G1_OP) G1OpShape::pair((int)z, l, r);
      else if (as.kind == SBN_AIR_LOOKUP) LookupShape().pair((int)z, l, r);
      else if (is_op_air(as.kind)) OpShape(as.kind).pair((int)z, l, r);
      else exp_shape(as).pair((int)z, l, r);
      pairs[z].lhs = l; pairs[z].rhs = r;
    }
    hipc(hipMalloc((void**)&P->d_pairs, std::max<size_t>(Z, 1) * sizeof(PairCols)), "hipMalloc");
    if (!rc && Z) hipc(hipMemcpy(P->d_pairs, pairs.data(), Z * sizeof(PairCols), hipMemcpyHostToDevice), "hipMemcpy");
    if (P->sp && P->sp->zr) {   // the split: the pairs of this rank's Z columns in local order
      SplitCtx* S = P->sp;
      std::vector<PairCols> own(S->zr);
      for (size_t l = 0; l < S->zr; l++) own[l] = pairs[S->zs.global_col(S->zs.rank, l)];
      hipc(hipMalloc((void**)&S->d_pairs_own, S->zr * sizeof(PairCols)), "hipMalloc");
      if (!rc) hipc(hipMemcpy(S->d_pairs_own, own.data(), S->zr * sizeof(PairCols), hipMemcpyHostToDevice), "hipMemcpy");
    }
  }
  hipc(hipStreamSynchronize(P->stream), "hipStreamSynchronize");
  hipc(hipGetLastError(), "table kernels");
  if (rc) { sbn_prover_destroy(P); return rc; }
  *out = P;
  return SBN_OK;
}

extern "C" void sbn_prover_destroy(sbn_prover* P) {
  if (!P) return;
  (void)hipSetDevice(P->device);
  if (P->rc_pending) (void)hipEventSynchronize(P->rc_done);
  u64* bufs[] = {P->d_trace, P->d_coef, P->d_lde, P->d_tmp, P->d_tmp2, P->d_tmp3, P->d_zval, P->d_zcoef, P->d_zlde, P->d_q, P->d_qlde, P->tree_t.d, P->tree_z.d,
                 P->tree_q.d, P->d_tw_f, P->d_tw_i, P->d_shift, P->d_shift_inv, P->d_xs, P->d_lag_first, P->d_lag_last, P->d_apow, P->d_zpow,
                 P->d_open, P->d_part, P->d_w, P->d_fa, P->d_fcoef, P->d_fcoef2, P->d_pow, P->d_qbuf, P->d_shift_odd};
  for (u64* b : bufs) if (b) (void)hipFree(b);
  for (u64* b : P->fri_vals) if (b) (void)hipFree(b);
  for (auto& t : P->fri_trees) if (t.d) (void)hipFree(t.d);
  if (P->sp) {
    if (P->sp->d_ldechunk) (void)hipFree(P->sp->d_ldechunk);
    if (P->sp->d_idx_local) (void)hipFree(P->sp->d_idx_local);
    if (P->sp->d_pairs_own) (void)hipFree(P->sp->d_pairs_own);
    for (auto& e : P->sp->xchg_done) if (e) (void)hipEventDestroy(e);
    for (auto& e : P->sp->tev) if (e) (void)hipEventDestroy(e);
    if (P->sp->cstream) (void)hipStreamDestroy(P->sp->cstream);
    delete P->sp;
  }
  if (P->d_pic) (void)hipFree(P->d_pic);
  if (P->d_idx) (void)hipFree(P->d_idx);
  if (P->d_pairs) (void)hipFree(P->d_pairs);
  for (auto& e : P->ev) if (e) (void)hipEventDestroy(e);   // a partly built context (failed create) holds null handles
  for (auto& e : P->abs_ev) if (e) (void)hipEventDestroy(e);
  for (auto& e : P->chunk_ready) if (e) (void)hipEventDestroy(e);
  if (P->hash_done) (void)hipEventDestroy(P->hash_done);
  if (P->d_sponge) (void)hipFree(P->d_sponge);
  if (P->h_chain) (void)hipHostFree(P->h_chain);
  if (P->h_io) (void)hipHostFree(P->h_io);
  if (P->h_open) (void)hipHostFree(P->h_open);
  if (P->h_open2) (void)hipHostFree(P->h_open2);
  if (P->hstream) (void)hipStreamDestroy(P->hstream);
  if (P->rstream) (void)hipStreamDestroy(P->rstream);
  if (P->rows_done) (void)hipEventDestroy(P->rows_done);
  if (P->rc_done) (void)hipEventDestroy(P->rc_done);
  if (P->d_rc_err) (void)hipFree(P->d_rc_err);
  if (P->nstream) { (void)hipStreamDestroy(P->nstream); for (auto& e : P->intt_done) if (e) (void)hipEventDestroy(e); }
  if (P->stream) (void)hipStreamDestroy(P->stream);
  delete P;
}

static int check_pi(sbn_prover* P, const uint64_t* pi, size_t n_pi) {
  if (n_pi != P->air.npi) return fail(SBN_ERR_BAD_ARG, "expected %zu public inputs, got %zu", P->air.npi, n_pi);
  if (n_pi && !pi) return fail(SBN_ERR_BAD_ARG, "null public inputs");
  for (size_t i = 0; i < n_pi; i++) if (pi[i] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "public input %zu is not canonical", i);
  P->pi.assign(pi, pi + n_pi);
  return 0;
}
// The asynchronous range check (sbn_prover::rstream): every reader or writer of the trace other than the trace commitment waits
// for it here and learns its verdict; prove() calls this after the commitment, which joined it on the stream.
static int rc_finish(sbn_prover* P) {
  if (!P->rc_pending) return SBN_OK;
  P->rc_pending = false;
  int err = 0;
  HIPC(hipEventSynchronize(P->rc_done));
  HIPC(hipMemcpy(&err, P->d_rc_err, sizeof(int), hipMemcpyDeviceToHost));
  if (err) { P->loaded = false; return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16"); }
  return SBN_OK;
}
extern "C" int sbn_prover_load_trace(sbn_prover* P, const uint64_t* trace, const uint64_t* pi, size_t n_pi) {
  if (!P || !trace) return fail(SBN_ERR_BAD_ARG, "null argument");
  int rc = check_pi(P, pi, n_pi); if (rc) return rc;
  (void)rc_finish(P);   // (a range check still writing the columns this call overwrites)
  size_t words = P->air.ncols * P->n;
  {  // canonical-form check on several host threads (the copy below is the PCIe-bound part)
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 16) nt = 16;
    std::atomic<size_t> bad(words);
    std::vector<std::thread> th;
    size_t per = (words + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++)
      th.emplace_back([&, t] {
        size_t a = t * per, b = std::min(words, a + per);
        for (size_t i = a; i < b; i++) if (trace[i] >= GLP) { size_t cur = bad.load(); while (i < cur && !bad.compare_exchange_weak(cur, i)) {} break; }
      });
    for (auto& x : th) x.join();
    if (bad.load() < words) return fail(SBN_ERR_NON_CANONICAL, "trace word %zu is not canonical", bad.load());
  }
  HIPC(hipSetDevice(P->device));
  HIPC(hipMemcpy(P->d_trace, trace, words * sizeof(u64), hipMemcpyHostToDevice));
  P->loaded = true;
  return SBN_OK;
}
extern "C" int sbn_prover_load_trace_device(sbn_prover* P, const uint64_t* d_trace, const uint64_t* pi, size_t n_pi) {
  if (!P || !d_trace) return fail(SBN_ERR_BAD_ARG, "null argument");
  int rc = check_pi(P, pi, n_pi); if (rc) return rc;
  HIPC(hipSetDevice(P->device));
  (void)rc_finish(P);
  if (d_trace != P->d_trace) HIPC(hipMemcpy(P->d_trace, d_trace, P->air.ncols * P->n * sizeof(u64), hipMemcpyDeviceToDevice));
  P->loaded = true;
  return SBN_OK;
}
// ---- on-device witness generation (kernels_tracegen.cuh) ----------------------------------------------------------
// Scratch lives in the (not yet used) LDE buffer; the only host traffic is the instance list in (20 KB) and the
// instance outputs + error word back (8 KB).
// the u16 range-check kernel keeps 156 KB in LDS (> the 64 KiB default); idempotent, see ntt_fast_setup
static int range_check_setup(int device) {
  static std::atomic<bool> done[SBN_MAX_DEVICES];
  const int d = device >= 0 && device < SBN_MAX_DEVICES ? device : 0;
  if (!done[d].load()) {
    HIPC(hipFuncSetAttribute((const void*)tg::range_check_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tg::RC_LDS_BYTES));
    HIPC(hipFuncSetAttribute((const void*)tg::range_check_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tg::RC_LDS_BYTES));
    done[d].store(true);
  }
  return 0;
}
// One step of the curve chains as levels of independent micro-operations for tg::chain_coop_kernel: the formulas of
// bnw::jac_double / bnw::jac_add (bn254w.cuh) written once over builder values, Fq2 products expanded into four Fq products,
// every value in a fresh slot (no hazards), level = 1 + the deepest operand.  Program 0: a <- 2a (exponent bit clear);
// program 1: b <- b + a, a <- 2a.  Persistent slots: a.X a.Y a.Z b.X b.Y b.Z, E each, in that order from slot 0.
struct ChainProgram {
  std::vector<uint32_t> ops[2];
  int levels[2] = {0, 0};
  int slots = 0;
};
static ChainProgram build_chain_program(int E) {
  struct Op { int kind, d, a, b, level; };
  struct Val { int s[2]; };
  ChainProgram out;
  for (int bit = 0; bit < 2; bit++) {
    std::vector<int> lvl;
    std::vector<Op> ops;
    auto fresh = [&](int level) { lvl.push_back(level); return (int)lvl.size() - 1; };
    auto emit = [&](int kind, int a, int b) { const int L = 1 + std::max(lvl[a], lvl[b]); const int d = fresh(L); ops.push_back({kind, d, a, b, L}); return d; };
    auto add = [&](Val x, Val y) { Val r{}; for (int q = 0; q < E; q++) r.s[q] = emit(tg::CP_ADD, x.s[q], y.s[q]); return r; };
    auto sub = [&](Val x, Val y) { Val r{}; for (int q = 0; q < E; q++) r.s[q] = emit(tg::CP_SUB, x.s[q], y.s[q]); return r; };
    auto mul = [&](Val x, Val y) {
      Val r{};
      if (E == 1) { r.s[0] = emit(tg::CP_MUL, x.s[0], y.s[0]); return r; }
      const int t0 = emit(tg::CP_MUL, x.s[0], y.s[0]), t1 = emit(tg::CP_MUL, x.s[1], y.s[1]);
      const int t2 = emit(tg::CP_MUL, x.s[0], y.s[1]), t3 = emit(tg::CP_MUL, x.s[1], y.s[0]);
      r.s[0] = emit(tg::CP_SUB, t0, t1); r.s[1] = emit(tg::CP_ADD, t2, t3);   // Fq2 = Fq[i] / (i^2 + 1), cmul of bn254w.cuh
      return r;
    };
    auto chk_zero = [&](Val x) {   // czero: every component zero -> TG_ERR_DEGENERATE
      const int a = x.s[0], b = E == 2 ? x.s[1] : x.s[0];
      ops.push_back({E == 2 ? tg::CP_CHK2 : tg::CP_CHK1, 0, a, b, 1 + std::max(lvl[a], lvl[b])});
    };
    Val pa[3], pb[3];                                           // persistent a, b: slots 0 .. 6E-1, level 0
    for (int c = 0; c < 3; c++) for (int q = 0; q < E; q++) pa[c].s[q] = fresh(0);
    for (int c = 0; c < 3; c++) for (int q = 0; q < E; q++) pb[c].s[q] = fresh(0);
    Val nb[3] = {pb[0], pb[1], pb[2]};
    if (bit) {   // b + a: add-2007-bl, p = b, q = a (exp_chains: b = jac_add(b, a))
      const Val Z1Z1 = mul(pb[2], pb[2]), Z2Z2 = mul(pa[2], pa[2]);
      const Val U1 = mul(pb[0], Z2Z2), U2 = mul(pa[0], Z1Z1);
      const Val S1 = mul(mul(pb[1], pa[2]), Z2Z2), S2 = mul(mul(pa[1], pb[2]), Z1Z1);
      const Val H = sub(U2, U1);
      chk_zero(H);
      const Val H2 = add(H, H), I = mul(H2, H2), J = mul(H, I);
      const Val rr0 = sub(S2, S1), r = add(rr0, rr0);
      const Val V = mul(U1, I);
      nb[0] = sub(sub(mul(r, r), J), add(V, V));
      const Val sj = mul(S1, J);
      nb[1] = sub(mul(r, sub(V, nb[0])), add(sj, sj));
      const Val zs = add(pb[2], pa[2]);
      nb[2] = mul(sub(sub(mul(zs, zs), Z1Z1), Z2Z2), H);
    }
    chk_zero(pa[1]);   // exp_chains: czero(a.Y) before every doubling
    Val na[3];
    {   // 2a: dbl-2009-l
      const Val A = mul(pa[0], pa[0]), B = mul(pa[1], pa[1]), C = mul(B, B);
      const Val t0 = add(pa[0], B), t1 = mul(t0, t0), t2 = sub(sub(t1, A), C);
      const Val D = add(t2, t2), Ee = add(add(A, A), A), F = mul(Ee, Ee);
      na[0] = sub(F, add(D, D));
      const Val C2 = add(C, C), C4 = add(C2, C2), C8 = add(C4, C4);
      na[1] = sub(mul(Ee, sub(D, na[0])), C8);
      const Val yz = mul(pa[1], pa[2]);
      na[2] = add(yz, yz);
    }
    int top = 0;
    for (const Op& o : ops) top = std::max(top, o.level);
    for (int c = 0; c < 3; c++) for (int q = 0; q < E; q++) {   // the new points replace the old ones after every read
      ops.push_back({tg::CP_COPY, pa[c].s[q], na[c].s[q], na[c].s[q], top + 1});
      if (bit) ops.push_back({tg::CP_COPY, pb[c].s[q], nb[c].s[q], nb[c].s[q], top + 1});
    }
    const int nl = top + 1;
    out.levels[bit] = nl;
    out.slots = std::max(out.slots, (int)lvl.size());
    out.ops[bit].assign((size_t)nl * tg::CP_LANES, 0u);
    std::vector<int> fill(nl, 0);
    for (const Op& o : ops) {
      const int L = o.level - 1;
      if (fill[L] >= tg::CP_LANES || (int)lvl.size() > tg::CP_MAX_SLOTS) { out.levels[0] = out.levels[1] = -1; return out; }   // (cannot happen for E <= 2: checked by the caller)
      out.ops[bit][(size_t)L * tg::CP_LANES + fill[L]++] = (uint32_t)o.kind | ((uint32_t)o.d << 8) | ((uint32_t)o.a << 16) | ((uint32_t)o.b << 24);
    }
  }
  return out;
}

template <int E>
static int generate_trace_device(sbn_prover* P, const uint32_t* ios, size_t K, uint64_t* pi_out) {
  const size_t n = P->n;
  const size_t IOW = 8 * (4 * E + 1);  // u32 words per instance: x and offset (2E Fq each) + exp_val
  for (size_t k = 0; k < K; k++)
    for (int v = 0; v < 4 * E; v++) {
      u64 t[4]; for (int i = 0; i < 4; i++) t[i] = (u64)ios[IOW * k + 8 * v + 2 * i] | ((u64)ios[IOW * k + 8 * v + 2 * i + 1] << 32);
      if (bnw::geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coordinate >= p (instance %zu)", k);
    }
  HIPC(hipSetDevice(P->device));
  hipStream_t st = P->stream;
  const ExpShape sh = exp_shape(P->air);
  P->loaded = false;
  // carve the scratch
  u64* const wbase = P->sp ? (u64*)P->sp->comm.recv_buf : P->d_lde;   // scratch: the LDE buffer, not yet in use
  u64* w = wbase;
  auto take = [&](size_t words) { u64* r = w; w += (words + 7) & ~(size_t)7; return r; };
  const size_t cw = 257 * 12 * E * K;  // one Jacobian chain of every instance
  u64* ja = take(cw); u64* jb = take(cw);
  u64* sv = take(28 * E * n);       u64* inv = take(n);
  u64* d_out = take(16 * E * K);
  unsigned char* row_op = (unsigned char*)take(n / 8 + 1);
  uint32_t* d_ios = (uint32_t*)take(IOW * K / 2 + 1);
  uint32_t* d_prog[2] = {(uint32_t*)take(64 * tg::CP_LANES / 2), (uint32_t*)take(64 * tg::CP_LANES / 2)};   // chain programs: <= 64 levels of 64 micro-operations
  int* d_err = (int*)take(1);
  unsigned int* d_cnt = n > 65536 ? (unsigned int*)take((size_t)sh.num_rc * 32768) : nullptr;   // u32 histograms of the range-checked columns
  if ((size_t)(w - wbase) > P->lde_scratch_words) return fail(SBN_ERR_UNSUPPORTED, "scratch does not fit");
  if (int rc = range_check_setup(P->device)) return rc;

  const bool timing = P->set.trace_timing;
  hipEvent_t e0 = P->abs_ev[0], e1 = P->abs_ev[1];
  std::vector<hipEvent_t> kev;
  auto mark = [&]() { if (timing) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, st) == hipSuccess) kev.push_back(e); } };
  // pinned staging (a copy from / to pageable memory blocks the calling thread inside the runtime, once per copy)
  const size_t io_words = (IOW * K + 1) / 2, out_words = 16 * E * K + 1;
  if (P->h_io_words < io_words + out_words) {
    if (P->h_io) (void)hipHostFree(P->h_io);
    P->h_io = nullptr; P->h_io_words = 0;
    HIPC(hipHostMalloc((void**)&P->h_io, (io_words + out_words) * sizeof(u64), hipHostMallocDefault));
    P->h_io_words = io_words + out_words;
  }
  memcpy(P->h_io, ios, IOW * K * sizeof(uint32_t));
  u64* const h_out = P->h_io + io_words;
  HIPC(hipEventRecord(e0, st));
  HIPC(hipMemcpyAsync(d_ios, P->h_io, IOW * K * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIPC(hipMemsetAsync(d_err, 0, sizeof(int), st));
  auto blocks = [](size_t k, unsigned b) { return dim3((unsigned)((k + b - 1) / b)); };
  mark();
  const int skip = P->gen_calls++ ? P->set.tracegen_skip : 0;   // measurement only (settings.hpp): the first call of a prover always writes the whole trace
  if (!(skip & 1)) {
  hipLaunchKernelGGL(tg::flags_kernel, blocks(n, 256), dim3(256), 0, st, d_ios, IOW, n, sh.start_flags, P->d_trace);
  hipLaunchKernelGGL(tg::small_inverse_kernel, blocks(n, 256), dim3(256), 0, st, inv, n);
  hipLaunchKernelGGL(tg::periodic_kernel, blocks(n, 256), dim3(256), 0, st, inv, n, sh.start_periodic, sh.start_io_pulses, sh.start_lookups, (u64)65535, P->d_trace);
  hipLaunchKernelGGL(tg::io_pulse_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(2 * K)), dim3(256), 0, st, inv, n, (size_t)sh.rpb, sh.witness_col(0), P->d_trace);
  }
  mark();
  // the two 256-step curve chains per instance: host threads while the device writes the input-independent columns
  // chain_mode (SBN_TRACEGEN_DEVICE_CHAIN; create_ctx picks by the host pool's size): 2 = one wave per instance walking levels of
  // independent Fq operations (tg::chain_coop_kernel), 1 = one lane per instance (tg::chain_kernel, 13 ms), 0 = host threads +
  // pinned upload
  if (skip & 2) {
  } else if (P->chain_mode == 2) {
    static const ChainProgram prog = build_chain_program(E);
    if (prog.levels[0] <= 0 || prog.levels[0] > 24 || prog.levels[1] > 24) return fail(SBN_ERR_UNSUPPORTED, "internal: chain program does not fit");
    tg::ChainProgDev cp{};
    for (int b = 0; b < 2; b++) {
      HIPC(hipMemcpyAsync(d_prog[b], prog.ops[b].data(), prog.ops[b].size() * sizeof(uint32_t), hipMemcpyHostToDevice, st));
      cp.ops[b] = d_prog[b]; cp.levels[b] = prog.levels[b];
    }
    for (int v = 0; v < 4; v++) for (int q = 0; q < E; q++) cp.in_slot[v * E + q] = (unsigned char)((v < 2 ? v : v + 1) * E + q);   // a.X a.Y | b.X b.Y
    cp.one_slot[0] = (unsigned char)(2 * E); cp.one_slot[1] = (unsigned char)(5 * E);
    cp.zero_slot[0] = (unsigned char)(2 * E + 1); cp.zero_slot[1] = (unsigned char)(5 * E + 1);
    for (int i = 0; i < 6 * E; i++) cp.coord[i] = (unsigned char)i;
    hipLaunchKernelGGL(tg::chain_coop_kernel<E>, dim3((unsigned)K), dim3(tg::CP_LANES), 0, st, d_ios, K, ja, jb, d_err, cp);
  } else if (P->chain_mode == 1) hipLaunchKernelGGL(tg::chain_kernel<E>, blocks(K, 64), dim3(64), 0, st, d_ios, K, ja, jb, d_err);
  else {
    if (P->h_chain_words < 2 * cw) {
      if (P->h_chain) (void)hipHostFree(P->h_chain);
      P->h_chain = nullptr; P->h_chain_words = 0;
      HIPC(hipHostMalloc((void**)&P->h_chain, 2 * cw * sizeof(u64), hipHostMallocDefault));
      P->h_chain_words = 2 * cw;
    }
    if (tracegen_host_chains(E, ios, K, P->h_chain, P->h_chain + cw)) return fail(SBN_ERR_WITNESS, "degenerate affine operation (x1 == x2 or y == 0)");
    HIPC(hipMemcpyAsync(ja, P->h_chain, cw * sizeof(u64), hipMemcpyHostToDevice, st));
    HIPC(hipMemcpyAsync(jb, P->h_chain + cw, cw * sizeof(u64), hipMemcpyHostToDevice, st));
  }
  mark();
  if (!(skip & 2))
  hipLaunchKernelGGL(tg::affine_lambda_kernel<E>, blocks((n + tg::TG_ROWS - 1) / tg::TG_ROWS, 64), dim3(64), 0, st, d_ios, K, ja, jb, n, sv, row_op, d_out, d_err);
  mark();
  if (!(skip & 4))
  hipLaunchKernelGGL(tg::gadget_witness_kernel<E>, blocks(3 * E * n, 256), dim3(256), 0, st, sv, row_op, n, sh.gadget_col, P->d_trace, d_err);
  mark();
  if (skip & 8) {
  } else if (n > 65536) {   // multiplicities beyond u16: histogram of every target column in HBM first (kernels_tracegen.cuh)
    HIPC(hipMemsetAsync(d_cnt, 0, (size_t)sh.num_rc * 65536 * sizeof(unsigned int), st));
    hipLaunchKernelGGL(tg::range_count_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)sh.num_rc), dim3(256), 0, st, P->d_trace, n, sh.rc_start, d_cnt, d_err);
    hipLaunchKernelGGL(tg::range_check_kernel<true>, dim3((unsigned)sh.num_rc), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, st, P->d_trace, n, sh.rc_start, sh.start_lookups, d_err, d_cnt, P->set.range_check);
  } else if (P->set.range_async && !timing && P->rstream) {
    // behind this call: on its own stream after the rows are written; the trace commitment joins it (commit_pipeline), prove() reads its verdict
    HIPC(hipEventRecord(P->rows_done, st));
    HIPC(hipStreamWaitEvent(P->rstream, P->rows_done, 0));
    HIPC(hipMemsetAsync(P->d_rc_err, 0, sizeof(int), P->rstream));
    hipLaunchKernelGGL(tg::range_check_kernel<false>, dim3((unsigned)sh.num_rc), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, P->rstream, P->d_trace, n, sh.rc_start, sh.start_lookups, P->d_rc_err, (const unsigned int*)nullptr, P->set.range_check);
    HIPC(hipEventRecord(P->rc_done, P->rstream));
    P->rc_pending = true; P->rc_first_col = sh.start_lookups;
  } else {
    hipLaunchKernelGGL(tg::range_check_kernel<false>, dim3((unsigned)sh.num_rc), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, st, P->d_trace, n, sh.rc_start, sh.start_lookups, d_err, (const unsigned int*)nullptr, P->set.range_check);
  }
  mark();
  HIPC(hipGetLastError());
  HIPC(hipMemcpyAsync(h_out, d_out, 16 * E * K * sizeof(u64), hipMemcpyDeviceToHost, st));
  HIPC(hipMemcpyAsync(h_out + 16 * E * K, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipEventRecord(e1, st));
  HIPC(hipStreamSynchronize(st));
  const u64* out = h_out;
  const int err = (int)(h_out[16 * E * K] & 0xffffffffu);
  float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
  P->stage_ms[ST_COUNT + EX_TRACEGEN_MS] = ms;
  if (timing) {
    static const char* names[] = {"flags+pulses", "chains", "affine+lambda", "row_witness", "range_check"};
    for (size_t i = 0; i + 1 < kev.size(); i++) { float t = 0; (void)hipEventElapsedTime(&t, kev[i], kev[i + 1]); fprintf(stderr, "[device tracegen] %-14s %8.3f ms\n", names[i], t); }
    for (auto e : kev) (void)hipEventDestroy(e);
    fprintf(stderr, "[device tracegen] %-14s %8.3f ms\n", "total", ms);
  }
  if (err & tg::TG_ERR_DEGENERATE) return fail(SBN_ERR_WITNESS, "degenerate affine operation (x1 == x2 or y == 0)");
  if (err & tg::TG_ERR_WITNESS) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (err & tg::TG_ERR_RANGE) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  // public inputs: x, offset, exp_val, output as u32 limbs (g1/exp.rs:124-135, g2/exp.rs:139-156)
  P->pi.resize(P->air.npi);
  for (size_t k = 0; k < K; k++) {
    u64* p = P->pi.data() + (size_t)sh.pi_per_io * k;
    for (size_t i = 0; i < IOW; i++) p[i] = ios[IOW * k + i];
    for (int i = 0; i < 16 * E; i++) p[IOW + i] = out[16 * E * k + i];
  }
  if (pi_out) memcpy(pi_out, P->pi.data(), P->pi.size() * sizeof(u64));
  P->loaded = true;
  return SBN_OK;
}

// Fq12ExpStark: the square-and-multiply chains (no inversion anywhere) on host threads in standard form, then one lane
// per row for the limb columns and the twelve modular-gadget witnesses, and the split range check per target column.
static int generate_trace_device_fq12(sbn_prover* P, const uint32_t* ios, size_t K, uint64_t* pi_out) {
  const bool u64e = P->air.kind == SBN_AIR_FQ12_EXP_U64;        // 128-row instances, one-element exponent
  const size_t n = P->n, IOW = u64e ? 194 : 200;
  const int steps = u64e ? 64 : 256, log_rpb = u64e ? 7 : 9;
  for (size_t k = 0; k < K; k++)
    for (int v = 0; v < 24; v++) {
      u64 t[4]; for (int i = 0; i < 4; i++) t[i] = (u64)ios[IOW * k + 8 * v + 2 * i] | ((u64)ios[IOW * k + 8 * v + 2 * i + 1] << 32);
      if (bnw::geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coefficient >= p (instance %zu)", k);
    }
  HIPC(hipSetDevice(P->device));
  hipStream_t st = P->stream;
  const ExpShape sh = exp_shape(P->air);
  P->loaded = false;
  u64* const wbase = P->sp ? (u64*)P->sp->comm.recv_buf : P->d_lde;   // scratch: the LDE buffer, not yet in use
  u64* w = wbase;
  auto take = [&](size_t words) { u64* r = w; w += (words + 7) & ~(size_t)7; return r; };
  if (u64e)
    for (size_t k = 0; k < K; k++)
      if (((u64)ios[IOW * k + 192] | ((u64)ios[IOW * k + 193] << 32)) >= GLP) return fail(SBN_ERR_NON_CANONICAL, "exponent of instance %zu is not a canonical field element", k);
  const size_t cw = (size_t)(steps + 1) * 48 * K;  // one chain of every instance, standard form
  u64* ca = take(cw); u64* cb = take(cw);
  u64* inv = take(n);
  u64* d_outs = take(K * 48);
  uint32_t* d_ios = (uint32_t*)take(IOW * K / 2 + 1);
  int* d_err = (int*)take(1);
  if ((size_t)(w - wbase) > P->lde_scratch_words) return fail(SBN_ERR_UNSUPPORTED, "scratch does not fit");

  const bool timing = P->set.trace_timing;
  hipEvent_t e0 = P->abs_ev[0], e1 = P->abs_ev[1];
  std::vector<hipEvent_t> kev;
  auto mark = [&]() { if (timing) { hipEvent_t e; if (hipEventCreate(&e) == hipSuccess && hipEventRecord(e, st) == hipSuccess) kev.push_back(e); } };
  HIPC(hipEventRecord(e0, st));
  HIPC(hipMemcpyAsync(d_ios, ios, IOW * K * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIPC(hipMemsetAsync(d_err, 0, sizeof(int), st));
  auto blocks = [](size_t k, unsigned b) { return dim3((unsigned)((k + b - 1) / b)); };
  mark();
  if (u64e) hipLaunchKernelGGL(tg::flags_u64_kernel, blocks(n, 256), dim3(256), 0, st, d_ios, IOW, n, sh.start_flags, P->d_trace);
  else hipLaunchKernelGGL(tg::flags_kernel, blocks(n, 256), dim3(256), 0, st, d_ios, IOW, n, sh.start_flags, P->d_trace);
  hipLaunchKernelGGL(tg::small_inverse_kernel, blocks(n, 256), dim3(256), 0, st, inv, n);
  hipLaunchKernelGGL(tg::periodic_kernel, blocks(n, 256), dim3(256), 0, st, inv, n, sh.start_periodic, sh.start_io_pulses, sh.start_lookups, (u64)255, P->d_trace);
  hipLaunchKernelGGL(tg::io_pulse_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(2 * K)), dim3(256), 0, st, inv, n, (size_t)sh.rpb, sh.witness_col(0), P->d_trace);
  mark();
  // the square-and-multiply chains: one workgroup per instance on the device (kernels_tracegen.cuh fq12_chain_kernel);
  // SBN_FQ12_HOST_CHAIN=1: the library's host threads + a pinned upload, as in round 2 (A/B)
  const bool host_chain = P->set.fq12_host_chain;
  if (host_chain) {
    if (P->h_chain_words < 2 * cw) {
      if (P->h_chain) (void)hipHostFree(P->h_chain);
      P->h_chain = nullptr; P->h_chain_words = 0;
      HIPC(hipHostMalloc((void**)&P->h_chain, 2 * cw * sizeof(u64), hipHostMallocDefault));
      P->h_chain_words = 2 * cw;
    }
    tracegen_host_chains_fq12(ios, IOW, steps, K, P->h_chain, P->h_chain + cw);
    HIPC(hipMemcpyAsync(ca, P->h_chain, 2 * cw * sizeof(u64), hipMemcpyHostToDevice, st));  // ca and cb are adjacent
  } else hipLaunchKernelGGL(tg::fq12_chain_kernel, dim3((unsigned)K), dim3(320), 0, st, d_ios, IOW, steps, ca, cb, d_outs);
  mark();
  // one lane per (row, output coefficient) by default; SBN_FQ12_ROW_KERNEL=1: round 2's one lane per row (A/B)
  const bool row_kernel = P->set.fq12_row_kernel;
  if (row_kernel) hipLaunchKernelGGL(tg::fq12_row_kernel, blocks(n, 64), dim3(64), 0, st, d_ios, IOW, log_rpb, ca, cb, n, P->d_trace, d_err);
  else hipLaunchKernelGGL(tg::fq12_gadget_kernel, blocks(12 * n, 256), dim3(256), 0, st, d_ios, IOW, log_rpb, ca, cb, n, P->d_trace, d_err);
  mark();
  hipLaunchKernelGGL(tg::split_range_check_kernel, dim3((unsigned)sh.num_rc), dim3(256), 0, st, P->d_trace, n, sh.rc_start, sh.start_lookups, d_err);
  mark();
  HIPC(hipGetLastError());
  int err = 0;
  HIPC(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  std::vector<u64> chain_out;                 // B[steps] of every instance (the outputs among the public inputs) when the chains ran on the device
  if (!host_chain) {
    chain_out.resize(K * 48);
    HIPC(hipMemcpyAsync(chain_out.data(), d_outs, K * 48 * sizeof(u64), hipMemcpyDeviceToHost, st));
  }
  HIPC(hipEventRecord(e1, st));
  HIPC(hipStreamSynchronize(st));
  float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
  P->stage_ms[ST_COUNT + EX_TRACEGEN_MS] = ms;
  if (timing) {
    static const char* names[] = {"flags+pulses", "chains", "row_witness", "range_check"};
    for (size_t i = 0; i + 1 < kev.size(); i++) { float t = 0; (void)hipEventElapsedTime(&t, kev[i], kev[i + 1]); fprintf(stderr, "[device tracegen] %-14s %8.3f ms\n", names[i], t); }
    for (auto e : kev) (void)hipEventDestroy(e);
    fprintf(stderr, "[device tracegen] %-14s %8.3f ms\n", "total", ms);
  }
  if (err & tg::TG_ERR_WITNESS) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (err & tg::TG_ERR_RANGE) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  // public inputs: x, offset as 16-bit limbs, exp_val, output = b at the last row (fq12/exp.rs:95-117)
  P->pi.resize(P->air.npi);
  for (size_t k = 0; k < K; k++) {
    u64* p = P->pi.data() + (size_t)sh.pi_per_io * k;
    for (int c = 0; c < 24; c++)
      for (int i = 0; i < 16; i++) p[16 * c + i] = (ios[IOW * k + 8 * c + (i >> 1)] >> (16 * (i & 1))) & 0xffff;
    if (u64e) p[384] = (u64)ios[IOW * k + 192] | ((u64)ios[IOW * k + 193] << 32);
    else for (int i = 0; i < 8; i++) p[384 + i] = ios[IOW * k + 192 + i];
    const u64* out = host_chain ? P->h_chain + cw + ((k * (steps + 1) + steps) * 12) * 4 : chain_out.data() + k * 48;  // B[steps]
    const int ob = 384 + sh.n_exp_slots;
    for (int c = 0; c < 12; c++) for (int i = 0; i < 16; i++) p[ob + 16 * c + i] = (out[4 * c + (i >> 2)] >> (16 * (i & 3))) & 0xffff;
  }
  if (pi_out) memcpy(pi_out, P->pi.data(), P->pi.size() * sizeof(u64));
  P->loaded = true;
  return SBN_OK;
}

// FqExpStark: chains on host threads (512 Montgomery products per instance), rows and the u16 range check on the device.
static int generate_trace_device_fq(sbn_prover* P, const uint32_t* ios, size_t K, uint64_t* pi_out) {
  const size_t n = P->n, IOW = 24;
  for (size_t k = 0; k < K; k++)
    for (int v = 0; v < 2; v++) {
      u64 t[4]; for (int i = 0; i < 4; i++) t[i] = (u64)ios[IOW * k + 8 * v + 2 * i] | ((u64)ios[IOW * k + 8 * v + 2 * i + 1] << 32);
      if (bnw::geq_p(t)) return fail(SBN_ERR_BAD_ARG, "value >= p (instance %zu)", k);
    }
  HIPC(hipSetDevice(P->device));
  hipStream_t st = P->stream;
  const ExpShape sh = exp_shape(P->air);
  P->loaded = false;
  u64* const wbase = P->sp ? (u64*)P->sp->comm.recv_buf : P->d_lde;   // scratch: the LDE buffer, not yet in use
  u64* w = wbase;
  auto take = [&](size_t words) { u64* r = w; w += (words + 7) & ~(size_t)7; return r; };
  const size_t cw = 257 * 4 * K;
  u64* ca = take(cw); u64* cb = take(cw);
  u64* inv = take(n);
  uint32_t* d_ios = (uint32_t*)take(IOW * K / 2 + 1);
  int* d_err = (int*)take(1);
  unsigned int* d_cnt = n > 65536 ? (unsigned int*)take((size_t)sh.num_rc * 32768) : nullptr;   // u32 histograms of the range-checked columns
  if ((size_t)(w - wbase) > P->lde_scratch_words) return fail(SBN_ERR_UNSUPPORTED, "scratch does not fit");
  if (int rc = range_check_setup(P->device)) return rc;
  hipEvent_t e0 = P->abs_ev[0], e1 = P->abs_ev[1];
  HIPC(hipEventRecord(e0, st));
  HIPC(hipMemcpyAsync(d_ios, ios, IOW * K * sizeof(uint32_t), hipMemcpyHostToDevice, st));
  HIPC(hipMemsetAsync(d_err, 0, sizeof(int), st));
  auto blocks = [](size_t k, unsigned b) { return dim3((unsigned)((k + b - 1) / b)); };
  hipLaunchKernelGGL(tg::flags_kernel, blocks(n, 256), dim3(256), 0, st, d_ios, IOW, n, sh.start_flags, P->d_trace);
  hipLaunchKernelGGL(tg::small_inverse_kernel, blocks(n, 256), dim3(256), 0, st, inv, n);
  hipLaunchKernelGGL(tg::periodic_kernel, blocks(n, 256), dim3(256), 0, st, inv, n, sh.start_periodic, sh.start_io_pulses, sh.start_lookups, (u64)65535, P->d_trace);
  hipLaunchKernelGGL(tg::io_pulse_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(2 * K)), dim3(256), 0, st, inv, n, (size_t)sh.rpb, sh.witness_col(0), P->d_trace);
  if (P->h_chain_words < 2 * cw) {
    if (P->h_chain) (void)hipHostFree(P->h_chain);
    P->h_chain = nullptr; P->h_chain_words = 0;
    HIPC(hipHostMalloc((void**)&P->h_chain, 2 * cw * sizeof(u64), hipHostMallocDefault));
    P->h_chain_words = 2 * cw;
  }
  tracegen_host_chains_fq(ios, K, P->h_chain, P->h_chain + cw);
  HIPC(hipMemcpyAsync(ca, P->h_chain, 2 * cw * sizeof(u64), hipMemcpyHostToDevice, st));  // ca and cb are adjacent
  hipLaunchKernelGGL(tg::fq_row_kernel, blocks(n, 128), dim3(128), 0, st, d_ios, ca, cb, n, P->d_trace, d_err);
  if (n > 65536) {   // multiplicities beyond u16: histogram of every target column in HBM first (kernels_tracegen.cuh)
    HIPC(hipMemsetAsync(d_cnt, 0, (size_t)sh.num_rc * 65536 * sizeof(unsigned int), st));
    hipLaunchKernelGGL(tg::range_count_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)sh.num_rc), dim3(256), 0, st, P->d_trace, n, sh.rc_start, d_cnt, d_err);
    hipLaunchKernelGGL(tg::range_check_kernel<true>, dim3((unsigned)sh.num_rc), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, st, P->d_trace, n, sh.rc_start, sh.start_lookups, d_err, d_cnt, P->set.range_check);
  } else {
    hipLaunchKernelGGL(tg::range_check_kernel<false>, dim3((unsigned)sh.num_rc), dim3(tg::RC_THREADS), tg::RC_LDS_BYTES, st, P->d_trace, n, sh.rc_start, sh.start_lookups, d_err, (const unsigned int*)nullptr, P->set.range_check);
  }
  HIPC(hipGetLastError());
  int err = 0;
  HIPC(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPC(hipEventRecord(e1, st));
  HIPC(hipStreamSynchronize(st));
  float ms = 0; HIPC(hipEventElapsedTime(&ms, e0, e1));
  P->stage_ms[ST_COUNT + EX_TRACEGEN_MS] = ms;
  if (err & tg::TG_ERR_WITNESS) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (err & tg::TG_ERR_RANGE) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  // public inputs: x, offset, exp_val, output = b at the last row, as u32 limbs (fq/exp.rs:98-108)
  P->pi.resize(P->air.npi);
  for (size_t k = 0; k < K; k++) {
    u64* p = P->pi.data() + (size_t)sh.pi_per_io * k;
    for (size_t i = 0; i < IOW; i++) p[i] = ios[IOW * k + i];
    const u64* out = P->h_chain + cw + (k * 257 + 256) * 4;  // B[256]
    for (int i = 0; i < 8; i++) p[24 + i] = (out[i >> 1] >> (32 * (i & 1))) & 0xffffffffULL;
  }
  if (pi_out) memcpy(pi_out, P->pi.data(), P->pi.size() * sizeof(u64));
  P->loaded = true;
  return SBN_OK;
}

extern "C" int sbn_prover_generate_trace(sbn_prover* P, const uint32_t* ios, size_t num_io, uint64_t* pi_out) {
  if (!P || !ios) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (!is_exp_air(P->air.kind)) return fail(SBN_ERR_UNSUPPORTED, "device witness generation covers the Exp tables (use sbn_generate_trace_g1_op + sbn_prover_load_trace)");
  if (num_io != P->air.num_io) return fail(SBN_ERR_BAD_ARG, "prover was created for %u instances, got %zu", P->air.num_io, num_io);
  if (P->n != exp_rows_per_instance(P->air.kind) * num_io) return fail(SBN_ERR_BAD_ARG, "degree_bits does not match the rows per instance");
  (void)rc_finish(P);   // (the range check of an earlier call that no proof consumed)
  if (P->air.kind == SBN_AIR_FQ12_EXP || P->air.kind == SBN_AIR_FQ12_EXP_U64) return generate_trace_device_fq12(P, ios, num_io, pi_out);
  if (P->n < 65536 || P->n > 262144) return fail(SBN_ERR_UNSUPPORTED, "device witness generation of the u16-range-check tables covers 2^16 .. 2^18 rows");
  if (P->air.kind == SBN_AIR_FQ_EXP) return generate_trace_device_fq(P, ios, num_io, pi_out);
  return P->air.kind == SBN_AIR_G1_EXP ? generate_trace_device<1>(P, ios, num_io, pi_out) : generate_trace_device<2>(P, ios, num_io, pi_out);
}

extern "C" int sbn_prover_read_trace(sbn_prover* P, uint64_t* out) {
  if (!P || !out) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (!P->loaded) return fail(SBN_ERR_BAD_ARG, "no trace loaded");
  HIPC(hipSetDevice(P->device));
  { const int rc = rc_finish(P); if (rc) return rc; }
  HIPC(hipMemcpy(out, P->d_trace, P->air.ncols * P->n * sizeof(u64), hipMemcpyDeviceToHost));
  return SBN_OK;
}
extern "C" uint64_t* sbn_prover_trace_device_ptr(sbn_prover* P) {
  if (!P) return nullptr;
  (void)hipSetDevice(P->device);
  (void)rc_finish(P);   // the caller reads or writes the buffer next
  return P->d_trace;
}

extern "C" int sbn_prover_stage_times(const sbn_prover* P, float* ms, int cap) {
  if (!P || !ms) return 0;
  int k = std::min(cap, (int)(ST_COUNT + EX_COUNT));
  for (int i = 0; i < k; i++) ms[i] = P->stage_ms[i];
  return k;
}
extern "C" const char* sbn_prover_stage_name(int i) { return (i >= 0 && i < ST_COUNT + EX_COUNT) ? STAGE_NAMES[i] : ""; }

// ---- prove ----------------------------------------------------------------------------------------
// quotient_kernel<KIND, PART>: PART 0 then 1 on the main stream, PART 2 (both permutation-check segments) on the second
template <int KIND>
static void launch_quotient_kind(sbn_prover* P, const QuotientParams& qp, size_t qblocks) {
  const dim3 g1((unsigned)qblocks, 1), g2((unsigned)qblocks, 2);
  if (qp.seg_mask & 12u) hipLaunchKernelGGL((quotient_kernel<KIND, 2>), g2, dim3(256), 0, P->hstream, qp, qp.apow[0], qp.apow[1], qp.pic);
  // SBN_QUOTIENT_TAIL (A/B): 1 = the tail segment behind the permutation checks on the second stream instead of behind the head;
  // 2 = the tail IN FRONT of the head on the main stream (it then walks the lookup columns at the same time as the permutation checks)
  const int tail_mode = P->set.quotient_tail;
  if (tail_mode == 2 && (qp.seg_mask & 2u)) hipLaunchKernelGGL((quotient_kernel<KIND, 1>), g1, dim3(256), 0, P->stream, qp, qp.apow[0], qp.apow[1], qp.pic);
  if (qp.seg_mask & 1u) hipLaunchKernelGGL((quotient_kernel<KIND, 0>), g1, dim3(256), 0, P->stream, qp, qp.apow[0], qp.apow[1], qp.pic);
  if (tail_mode != 2 && (qp.seg_mask & 2u)) hipLaunchKernelGGL((quotient_kernel<KIND, 1>), g1, dim3(256), 0, tail_mode == 1 ? P->hstream : P->stream, qp, qp.apow[0], qp.apow[1], qp.pic);
}
static int launch_quotient_parts(sbn_prover* P, const QuotientParams& qp, size_t qblocks) {
  switch (P->air.kind) {
    case SBN_AIR_G1_OP: launch_quotient_kind<1>(P, qp, qblocks); break;
    case SBN_AIR_G1_EXP: launch_quotient_kind<2>(P, qp, qblocks); break;
    case SBN_AIR_G2_EXP: launch_quotient_kind<3>(P, qp, qblocks); break;
    case SBN_AIR_FQ_EXP: launch_quotient_kind<5>(P, qp, qblocks); break;
    case SBN_AIR_FQ12_EXP_U64: launch_quotient_kind<6>(P, qp, qblocks); break;
    case SBN_AIR_MODULAR: launch_quotient_kind<7>(P, qp, qblocks); break;
    case SBN_AIR_FQ12_MUL: launch_quotient_kind<8>(P, qp, qblocks); break;
    case SBN_AIR_LOOKUP: launch_quotient_kind<9>(P, qp, qblocks); break;
    case SBN_AIR_FLAGS: launch_quotient_kind<10>(P, qp, qblocks); break;
    case SBN_AIR_FLAGS_U64: launch_quotient_kind<11>(P, qp, qblocks); break;
    default: launch_quotient_kind<4>(P, qp, qblocks); break;
  }
  HIPC(hipGetLastError());
  return 0;
}

extern "C" int sbn_prover_prove(sbn_prover* P, sbn_proof** out) {
  if (!P || !out) return fail(SBN_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  if (!P->loaded && !(P->sp && P->sp->comm.world > 1)) return fail(SBN_ERR_BAD_ARG, "no trace loaded");   // (split: agreed with the other ranks below)
  HIPC(hipSetDevice(P->device));
  hipStream_t st = P->stream;
  const size_t n = P->n, m = P->m, C = P->air.ncols, Z = P->air.nzs;
  const sbn_config& cfg = P->cfg;
  const size_t capw = ((size_t)1 << cfg.cap_height) * 4;
  auto blocks = [](size_t k) { return dim3((unsigned)((k + 255) / 256)); };
  int rc;
  Challenger ch;
  std::vector<u64> trace_cap, z_cap, q_cap;

  // P1 trace commitment ---------------------------------------------------------------------------
  HIPC(hipEventRecord(P->ev[ST_TRACE_COMMIT], st));
  SplitCtx* const S = P->sp;   // non-null: this rank's share of one trace split over S->comm.world GPUs
  struct DeadlineScope { ~DeadlineScope() { t_wait_deadline_s = 0; } } deadline_scope;
  if (S && S->comm.world > 1) {
    // every rank must have a witness before anyone walks into the first exchange (a rank whose generate_trace failed would
    // leave the others waiting for its blocks): agree on a status word first
    t_wait_deadline_s = P->set.comm_timeout_s;
    std::vector<uint32_t> st_all(S->comm.world, 0);
    const uint32_t mine = P->loaded ? 1u : 0u;
    if ((rc = split_all_gather_host(P, &mine, st_all.data(), sizeof(uint32_t)))) return rc;
    for (u32 r = 0; r < S->comm.world; r++) if (!st_all[r]) return fail(SBN_ERR_BAD_ARG, "split proof abandoned: rank %u has no trace loaded", r);
  }
  if (!P->loaded) return fail(SBN_ERR_BAD_ARG, "no trace loaded");
  if (S) {
    S->tev_used = 0;
    if (P->rc_pending) HIPC(hipStreamWaitEvent(st, P->rc_done, 0));
    if ((rc = commit_split(P, S->cs, P->d_trace, true, P->d_coef, S->lde_l, S->lde_n, P->tree_t))) return rc;
  } else if ((rc = commit_pipeline(P, P->d_trace, P->d_coef, P->d_lde, C, P->tree_t, EX_TRACE_ABSORB_MS, EX_TRACE_ABSORB_LAUNCHES))) return rc;
  HIPC(hipEventRecord(P->ev[ST_PERM_Z], st));
  if ((rc = S ? split_cap_to_host(P, P->tree_t, trace_cap) : tree_cap_to_host(P, P->tree_t, trace_cap))) return rc;
  if ((rc = rc_finish(P))) return rc;   // (the commitment joined the asynchronous range check; its verdict)
  ch.observe_words(trace_cap.data(), capw);

  // P2 permutation argument -------------------------------------------------------------------------
  // get_n_permutation_challenge_sets(num_challenges, batch_size=2): sets[s].challenges[c] = (beta, gamma)
  // (a table without permutation pairs -- FlagStark -- draws no permutation challenges and commits no Z: starky prover.rs
  //  `stark.uses_permutation_args()`)
  F gam[2][2] = {};
  if (Z) for (int s = 0; s < 2; s++) for (int c = 0; c < (int)cfg.num_challenges; c++) { (void)ch.challenge(); gam[s][c] = ch.challenge(); }
  const F gamma0 = gam[0][0], gamma1 = gam[1][1];  // instance i of a batch uses sets[i].challenges[chal]
  {
    const size_t zn = S ? S->zr : Z;                       // Z columns computed here (the split: this rank's range)
    const PairCols* pairs = S ? S->d_pairs_own : P->d_pairs;
    auto launch_z = [&](size_t z0, size_t cnt, hipStream_t s) {
      if (cnt == 0) return;
      const PairCols* pp = pairs + z0; u64* out = P->d_zval + z0 * n;
      // from 2^13 rows up: chunk products, one wave per column for their prefix / suffix, then Z written once -- 5 words per row
      // instead of 7 (kernels.cuh permz_chunk_*; the chunk products wait in the idle quotient scratch).  SBN_PERM_Z=1 (experiment
      // switch): the one-workgroup-per-column kernel of rounds 1-3.
      const size_t chunks = n / 2048;
      if (n >= 8192 && n % 2048 == 0 && P->set.perm_z != 1 && cnt <= 65535 && cnt * chunks * 4 <= 2 * 32 * n) {
        u64* tot = P->d_part; u64* pq = P->d_part + cnt * chunks * 2;
        hipLaunchKernelGGL(permz_chunk_products_kernel<8>, dim3((unsigned)chunks, (unsigned)cnt), dim3(256), 0, s, P->d_trace, n, pp, gamma0.v, gamma1.v, tot);
        hipLaunchKernelGGL(permz_chunk_scan_kernel, dim3((unsigned)cnt), dim3(64), 0, s, tot, (u32)chunks, pq);
        hipLaunchKernelGGL(permz_chunk_write_kernel<8>, dim3((unsigned)chunks, (unsigned)cnt), dim3(256), 0, s, P->d_trace, n, pp, gamma0.v, gamma1.v, pq, out);
        return;
      }
      if (n % 2048 == 0) hipLaunchKernelGGL(permutation_z_kernel<8>, dim3((unsigned)cnt), dim3(256), 0, s, P->d_trace, n, pp, gamma0.v, gamma1.v, out);
      else if (n % 1024 == 0) hipLaunchKernelGGL(permutation_z_kernel<4>, dim3((unsigned)cnt), dim3(256), 0, s, P->d_trace, n, pp, gamma0.v, gamma1.v, out);
      else hipLaunchKernelGGL(permutation_z_kernel<2>, dim3((unsigned)cnt), dim3(256), 0, s, P->d_trace, n, pp, gamma0.v, gamma1.v, out);   // n = 512
    };
    // MEASURED and dropped (profiles/r3_ab_z_overlap.txt): only the first column chunk in front of the commit pipeline and the
    // other Z columns on a third stream beside that chunk's transforms and sponge -- the stage in front shrinks by 0.2 ms (a
    // single workgroup's scan chain is 0.36 ms long), the Z commitment grows by 0.44 ms (26.25 -> 26.5 ms per proof, five pairs).
    launch_z(0, zn, st);
  }
  HIPC(hipGetLastError());
  HIPC(hipEventRecord(P->ev[ST_Z_COMMIT], st));
  if (S) {
    if ((rc = commit_split(P, S->zs, P->d_zval, false, P->d_zcoef, S->zlde_l, S->zlde_n, P->tree_z))) return rc;
  } else {
    if (C > 4) if ((rc = absorb_times(P, C, EX_TRACE_ABSORB_MS))) return rc;   // (27 event queries: behind the Z kernel, not in front of it)
    if (Z) if ((rc = commit_pipeline(P, P->d_zval, P->d_zcoef, P->d_zlde, Z, P->tree_z, EX_Z_ABSORB_MS, EX_Z_ABSORB_LAUNCHES))) return rc;
  }
  HIPC(hipEventRecord(P->ev[ST_QUOTIENT_EVAL], st));
  if (Z) {
    if ((rc = S ? split_cap_to_host(P, P->tree_z, z_cap) : tree_cap_to_host(P, P->tree_z, z_cap))) return rc;
    ch.observe_words(z_cap.data(), capw);
  }

  // P3 quotient -------------------------------------------------------------------------------------
  F alphas[SBN_NCH];
  for (int j = 0; j < SBN_NCH; j++) alphas[j] = ch.challenge();
  {
    const size_t APN = P->apow_n;
    std::vector<u64> apow((size_t)SBN_NCH * APN);
    std::vector<F> ap[SBN_NCH];
    for (int j = 0; j < SBN_NCH; j++) {
      ap[j].resize(APN);
      F a(1);
      for (size_t k = 0; k < APN; k++) { ap[j][k] = a; apow[(size_t)j * APN + k] = a.v; a = a * alphas[j]; }
    }
    HIPC(hipMemcpyAsync(P->d_apow, apow.data(), apow.size() * sizeof(u64), hipMemcpyHostToDevice, st));
    if (is_exp_air(P->air.kind)) {
      static thread_local ExpPiConsts<F> pic;  // 3 x 2 x 512 field elements
      const F* app[SBN_NCH] = {ap[0].data(), ap[1].data()};
      std::vector<F> pif(P->pi.size());
      for (size_t i = 0; i < pif.size(); i++) pif[i] = F(P->pi[i]);
      exp_pi_consts<F>(exp_shape(P->air), app, pif.data(), pic);
      HIPC(hipMemcpyAsync(P->d_pic, &pic, sizeof(pic), hipMemcpyHostToDevice, st));
    }
    HIPC(stream_wait(st));
  }
  {
    QuotientParams qp{};
    qp.lde = P->d_lde; qp.zlde = P->d_zlde; qp.m = m; qp.next_step = 2;  // 2^quotient_degree_bits
    qp.lde_next = qp.lde; qp.zlde_next = qp.zlde; qp.row_shift = 0; qp.row_rho = 0;
    if (S) {
      // this rank's LDE points j * R + rho; the row two LDE points on is local row j + 1 (two ranks) or sits at local row j
      // of the second plane (four ranks and more)
      qp.lde = S->lde_l; qp.zlde = S->zlde_l; qp.m = S->ml; qp.row_shift = S->log_r; qp.row_rho = S->rho;
      qp.lde_next = S->planes == 2 ? S->lde_n : S->lde_l; qp.zlde_next = S->planes == 2 ? S->zlde_n : S->zlde_l;
      qp.next_step = S->comm.world == 1 ? 2 : (S->comm.world == 2 ? 1 : 0);
    }
    qp.xs = P->d_xs; qp.lag_first = P->d_lag_first; qp.lag_last = P->d_lag_last;
    F gn = f_exp_pow2(F(GL_GEN), P->degree_bits);
    qp.zh_inv[0] = f_inv(gn - F(1)).v; qp.zh_inv[1] = f_inv(-gn - F(1)).v;  // Z_H(7 w^i) = 7^N (-1)^i - 1
    qp.last = f_inv(f_root_of_unity(P->degree_bits)).v;
    for (int j = 0; j < SBN_NCH; j++) { qp.alpha[j] = alphas[j].v; qp.apow[j] = P->d_apow + (size_t)j * P->apow_n; }
    qp.gamma0 = gamma0.v; qp.gamma1 = gamma1.v; qp.num_zs = (int)Z; qp.num_io = (int)P->air.num_io; qp.pic = P->d_pic; qp.qout = P->d_q;
    if (S) qp.qout = (u64*)S->comm.send_buf;   // [2][ml]: all-gathered below
    qp.part = P->d_part;   // QSEG x SBN_NCH planes of m words (the FRI combine's scratch, idle here)
    {
      // constraints that follow each segment: [AIR head][AIR tail], then the permutation block, whose two segments take the Z
      // columns below / from zsplit with exponents counted from the end of the stream (nothing follows either of them).
      // (Measured and dropped: the AIR tail on a third stream beside the other two -- 1.22 -> 1.32 ms for the stage, G2 2.0 -> 2.3.)
      qp.zsplit = (int)(Z / 2);
      qp.lookups_in_perm = P->set.quotient_lookups == 1;   // measured slower (quotient stage 1.18 -> 1.25 ms for G1): an experiment switch only
      const u64 n_tail = is_exp_air(P->air.kind) ? (u64)ExpShape(exp_e(P->air.kind), (int)P->air.num_io).num_tail_constraints() : 0;
      const u64 after[4] = {n_tail + 2 * (u64)Z, 2 * (u64)Z, 0, 0};
      qp.seg_count[0] = (int)(P->air.nconstraints - n_tail); qp.seg_count[1] = (int)n_tail;
      qp.seg_count[2] = qp.seg_count[3] = 2 * (int)Z;
      for (int sgm = 0; sgm < 4; sgm++) for (int j = 0; j < SBN_NCH; j++) qp.seg_shift[sgm][j] = f_pow(alphas[j], after[sgm]).v;
    }
    const size_t qblocks = (qp.m + 255) / 256;
    qp.seg_mask = 0xfu;
#ifdef SBN_DIAG   // diagnostic builds only (make CXXFLAGS+=-DSBN_DIAG): time single segments; the proof is invalid unless the mask is 15
    { const char* e = std::getenv("SBN_DIAG_QUOTIENT_SEGMASK"); if (e) qp.seg_mask = (u32)atoi(e) & 0xfu; }
    if (qp.seg_mask != 0xfu) HIPC(hipMemsetAsync(qp.part, 0, (size_t)QSEG * SBN_NCH * qp.m * sizeof(u64), st));
#endif
    // AIR head and tail on the main stream, the permutation checks beside them on the second stream (idle here)
    HIPC(hipEventRecord(P->chunk_ready[3], st));
    HIPC(hipStreamWaitEvent(P->hstream, P->chunk_ready[3], 0));
    if ((rc = launch_quotient_parts(P, qp, qblocks))) return rc;
    HIPC(hipEventRecord(P->hash_done, P->hstream));
    HIPC(hipStreamWaitEvent(st, P->hash_done, 0));
    hipLaunchKernelGGL(quotient_combine_kernel, blocks(qp.m), dim3(256), 0, st, qp);
    HIPC(hipGetLastError());
    if (S) {
      // quotient values of every rank's points -> all ranks, natural order; the 4 quotient columns are then committed on
      // every rank alike (no exchange: 4 columns)
      if ((rc = split_all_gather_device(P, 2 * S->ml))) return rc;
      hipLaunchKernelGGL(split_unpack_rows_kernel, blocks(m), dim3(256), 0, st, S->scratch, m, 2u, S->log_r, P->d_q);
      HIPC(hipGetLastError());
    } else if (Z > 4) { if ((rc = absorb_times(P, Z, EX_Z_ABSORB_MS))) return rc; }   // behind the quotient kernel
  }
  HIPC(hipEventRecord(P->ev[ST_QUOTIENT_COMMIT], st));
  // coset_ifft(7) of the 2 quotient value vectors (size m), in place via tmp; the result viewed as
  // [4][n] is exactly quotient_poly.chunks(degree) in the order q0_lo, q0_hi, q1_lo, q1_hi.
  if ((rc = ntt_columns(P, P->d_q, m, P->d_q, m, P->d_tmp, m, 2, P->lde_log, true, m, nullptr, P->d_shift_inv, host_inv_pow2(P->lde_log)))) return rc;
  if ((rc = lde_coeffs(P, P->d_q, P->d_qlde, 4))) return rc;
  if ((rc = tree_from_matrix(P, P->tree_q, P->d_qlde, 4))) return rc;
  HIPC(hipEventRecord(P->ev[ST_OPENINGS], st));
  if ((rc = tree_cap_to_host(P, P->tree_q, q_cap))) return rc;
  ch.observe_words(q_cap.data(), capw);

  // P4 openings -------------------------------------------------------------------------------------
  E2 zeta = ch.ext_challenge();
  F g = f_root_of_unity(P->degree_bits);
  if (e2_exp_pow2(zeta, P->degree_bits) == E2(F(1), F(0))) return fail(SBN_ERR_HIP, "opening point is in the subgroup");
  E2 zeta_next = zeta * g;
  hipLaunchKernelGGL(ext_pow_table_kernel, blocks(n), dim3(256), 0, st, P->d_zpow, P->d_zpow + n, n, zeta.a.v, zeta.b.v);
  hipLaunchKernelGGL(ext_pow_table_kernel, blocks(n), dim3(256), 0, st, P->d_zpow + 2 * n, P->d_zpow + 3 * n, n, zeta_next.a.v, zeta_next.b.v);
  // Transcript order: local (trace at zeta), Z at zeta, quotient at zeta, then next (trace at g*zeta), Z at g*zeta.  The
  // 1,220 sequential permutations that hash them are the longest host stretch of a proof, so the device produces the
  // values in that order behind three events: trace at zeta alone first (the host starts hashing after half an opening
  // pass), then Z and quotient at both points, then trace at g*zeta -- the coefficients are read twice, in time the
  // device would otherwise spend waiting for the host.
  const u64* open = P->h_open;
  const u64* open2 = P->h_open2;
  if (S && S->comm.world > 1) {
    // this rank evaluates its own trace / Z columns at both points (the quotient columns everywhere; own columns sit
    // compactly in local order at the start of the trace / Z sections of d_open), the values are all-gathered on the host,
    // scattered to their global positions and observed in transcript order
    const u64 *zp0 = P->d_zpow, *zp1 = P->d_zpow + n, *zp2 = P->d_zpow + 2 * n, *zp3 = P->d_zpow + 3 * n;
    auto open_k = n % 1024 == 0 ? openings_kernel<4> : openings_kernel<1>;
    if (S->cr) hipLaunchKernelGGL(open_k, dim3((unsigned)S->cr), dim3(256), 0, st, P->d_coef, n, zp0, zp1, zp2, zp3, P->d_open);
    if (S->zr) hipLaunchKernelGGL(open_k, dim3((unsigned)S->zr), dim3(256), 0, st, P->d_zcoef, n, zp0, zp1, zp2, zp3, P->d_open + C * 4);
    hipLaunchKernelGGL(open_k, dim3(4), dim3(256), 0, st, P->d_q, n, zp0, zp1, zp2, zp3, P->d_open + (C + Z) * 4);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(P->h_open, P->d_open, (C + Z + 4) * 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPC(hipEventRecord(P->ev[ST_FRI_COMBINE], st));
    HIPC(stream_wait(st));
    const u32 R = S->comm.world;
    const size_t cmax = S->cs.max_own(), zmax = S->zs.max_own(), per = (cmax + zmax) * 4;
    std::vector<u64> mine(per, 0), all((size_t)R * per);
    memcpy(mine.data(), P->h_open, S->cr * 4 * sizeof(u64));
    memcpy(mine.data() + cmax * 4, P->h_open + C * 4, S->zr * 4 * sizeof(u64));
    if ((rc = split_all_gather_host(P, mine.data(), all.data(), per * sizeof(u64)))) return rc;
    for (u32 r = 0; r < R; r++) {
      const u64* src = all.data() + (size_t)r * per;
      for (size_t l = 0, cnt = S->cs.own_cols(r); l < cnt; l++) memcpy(P->h_open + S->cs.global_col(r, l) * 4, src + l * 4, 4 * sizeof(u64));
      for (size_t l = 0, cnt = S->zs.own_cols(r); l < cnt; l++) memcpy(P->h_open + (C + S->zs.global_col(r, l)) * 4, src + (cmax + l) * 4, 4 * sizeof(u64));
    }
    open2 = open;
    for (size_t p = 0; p < C + Z + 4; p++) { ch.observe(F(open[4 * p])); ch.observe(F(open[4 * p + 1])); }
    for (size_t p = 0; p < C + Z; p++) { ch.observe(F(open[4 * p + 2])); ch.observe(F(open[4 * p + 3])); }
  } else {
    const bool u4 = n % 1024 == 0;
    auto open1_k = u4 ? openings1_kernel<4> : openings1_kernel<1>;
    const u64 *zp0 = P->d_zpow, *zp1 = P->d_zpow + n, *zp2 = P->d_zpow + 2 * n, *zp3 = P->d_zpow + 3 * n;
    // The host hashes ~3 columns per microsecond, the device evaluates ~5: the trace columns at zeta go out in slices that
    // double in size (C/16, C/8, C/4, rest), so the host starts after a sixteenth of an opening pass and is never starved.
    size_t sl[5] = {0, C / 16, C / 16 + C / 8, C / 16 + C / 8 + C / 4, C};
    for (int k = 0; k < 4; k++) {
      const size_t a = sl[k], cnt = sl[k + 1] - sl[k];
      if (cnt) {
        hipLaunchKernelGGL(open1_k, dim3((unsigned)cnt), dim3(256), 0, st, P->d_coef + a * n, n, zp0, zp1, P->d_open + a * 4, 0u);
        HIPC(hipMemcpyAsync(P->h_open + a * 4, P->d_open + a * 4, cnt * 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
      }
      HIPC(hipEventRecord(P->chunk_ready[8 + k], st));  // (the commit pipeline's chunk events are idle here)
    }
    // Z and quotient at zeta, then everything at g*zeta, again in slices: the host (0.86 us per permutation, 4 columns each)
    // is the slower side, and it must never find the next values missing -- with the trace at g*zeta evaluated last in one
    // kernel it idled 0.3 ms before it.
    if (Z) hipLaunchKernelGGL(open1_k, dim3((unsigned)Z), dim3(256), 0, st, P->d_zcoef, n, zp0, zp1, P->d_open + C * 4, 0u);
    hipLaunchKernelGGL(open1_k, dim3(4), dim3(256), 0, st, P->d_q, n, zp0, zp1, P->d_open + (C + Z) * 4, 0u);
    HIPC(hipMemcpyAsync(P->h_open + C * 4, P->d_open + C * 4, (Z + 4) * 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPC(hipEventRecord(P->chunk_ready[1], st));
    size_t sn[5] = {0, C / 4, C / 2, C / 2 + C / 4, C};
    for (int k = 0; k < 4; k++) {
      const size_t a = sn[k], cnt = sn[k + 1] - sn[k];
      if (cnt) {
        hipLaunchKernelGGL(open1_k, dim3((unsigned)cnt), dim3(256), 0, st, P->d_coef + a * n, n, zp2, zp3, P->d_open + a * 4, 2u);
        HIPC(hipMemcpyAsync(P->h_open2 + a * 4, P->d_open + a * 4, cnt * 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
      }
      HIPC(hipEventRecord(P->chunk_ready[12 + k], st));
    }
    if (Z) hipLaunchKernelGGL(open1_k, dim3((unsigned)Z), dim3(256), 0, st, P->d_zcoef, n, zp2, zp3, P->d_open + C * 4, 2u);
    HIPC(hipGetLastError());
    HIPC(hipMemcpyAsync(P->h_open2 + C * 4, P->d_open + C * 4, Z * 4 * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPC(hipEventRecord(P->ev[ST_FRI_COMBINE], st));
    // observe_openings: batch zeta = local ++ perm_zs ++ quotient ; batch g*zeta = next ++ perm_zs_next
    for (int k = 0; k < 4; k++) {
      HIPC(event_wait(P->chunk_ready[8 + k]));
      for (size_t p = sl[k]; p < sl[k + 1]; p++) { ch.observe(F(open[4 * p])); ch.observe(F(open[4 * p + 1])); }
    }
    HIPC(event_wait(P->chunk_ready[1]));
    for (size_t p = C; p < C + Z + 4; p++) { ch.observe(F(open[4 * p])); ch.observe(F(open[4 * p + 1])); }
    for (int k = 0; k < 4; k++) {
      HIPC(event_wait(P->chunk_ready[12 + k]));
      for (size_t p = sn[k]; p < sn[k + 1]; p++) { ch.observe(F(open2[4 * p + 2])); ch.observe(F(open2[4 * p + 3])); }
    }
    HIPC(stream_wait(st));
    for (size_t p = C; p < C + Z; p++) { ch.observe(F(open2[4 * p + 2])); ch.observe(F(open2[4 * p + 3])); }
  }

  // P5 FRI ------------------------------------------------------------------------------------------
  E2 fri_alpha = ch.ext_challenge();
  {
    // F1 = sum_{j < C+Z} alpha^j f_j ; F0 = F1 + alpha^(C+Z) * sum_{j<4} alpha^j q_j
    // polys per group; groups run on grid.y, <= 32 per launch.  The split: a group = one own column block (ColShare), whose
    // weight is alpha^(global position of the block); consecutive own blocks are `world` blocks apart.
    const u32 RW = S ? S->comm.world : 1;
    const u32 GS = RW > 1 ? (u32)SPLIT_BLOCK : 128;
    // all group weights alpha^(offset + g*stride) are known up front: one upload, no host sync in the loop
    std::vector<u64> wall;
    auto plan = [&](u32 npoly, E2 weight0, u64 stride) { size_t off = wall.size(); u32 ng = (npoly + GS - 1) / GS; E2 ag = e2_pow(fri_alpha, stride), cur = weight0;
                                                         for (u32 k = 0; k < ng; k++) { wall.push_back(cur.a.v); wall.push_back(cur.b.v); cur = cur * ag; } return off; };
    // the split: this rank's columns only, with the weights of their global positions
    const size_t tc0 = S ? (size_t)S->comm.rank * GS : 0, tcn = S ? S->cr : C, tzn = S ? S->zr : Z;
    const u64 gstride = (u64)GS * RW;
    size_t w_t = tcn ? plan((u32)tcn, e2_pow(fri_alpha, RW > 1 ? tc0 : 0), gstride) : 0, w_z = tzn ? plan((u32)tzn, e2_pow(fri_alpha, C + (RW > 1 ? tc0 : 0)), gstride) : 0,
           w_q = plan(4, e2_pow(fri_alpha, C + Z), GS);
    if (wall.size() > 4096) return fail(SBN_ERR_UNSUPPORTED, "too many FRI combine groups");
    HIPC(hipMemcpyAsync(P->d_w, wall.data(), wall.size() * sizeof(u64), hipMemcpyHostToDevice, st));
    // alpha^k, k < GS, as two planes in the quotient's alpha-power buffer (idle now; it holds at least 1,025 words per plane)
    hipLaunchKernelGGL(ext_pow_table_kernel, dim3(1), dim3(GS), 0, st, P->d_apow, P->d_apow + GS, (size_t)GS, fri_alpha.a.v, fri_alpha.b.v);
    auto combine = [&](const u64* coeffs, u32 npoly, size_t woff, u64* oa, u64* ob, int accumulate) -> int {
      u32 ng = (npoly + GS - 1) / GS;
      for (u32 g0 = 0; g0 < ng; g0 += 32) {
        u32 gc = std::min<u32>(32, ng - g0);
        hipLaunchKernelGGL(fri_combine_partial_kernel, dim3((unsigned)((n + 255) / 256), gc), dim3(256), 0, st, coeffs + (size_t)g0 * GS * n, n,
                           npoly - g0 * GS, GS, P->d_apow, P->d_apow + GS, P->d_part, P->d_part + 32 * n);
        hipLaunchKernelGGL(fri_combine_reduce_kernel, blocks(n), dim3(256), 0, st, P->d_part, P->d_part + 32 * n, n, gc, P->d_w + woff + 2 * g0, oa, ob,
                           (accumulate || g0 > 0) ? 1 : 0);
      }
      HIPC(hipGetLastError());
      return 0;
    };
    u64 *f1a = P->d_fb, *f1b = P->d_fb + n, *f0a = P->d_fa, *f0b = P->d_fa + n;
    if (!tcn) HIPC(hipMemsetAsync(f1a, 0, 2 * n * sizeof(u64), st));
    if (tcn) if ((rc = combine(P->d_coef, (u32)tcn, w_t, f1a, f1b, 0))) return rc;
    if (tzn) if ((rc = combine(P->d_zcoef, (u32)tzn, w_z, f1a, f1b, 1))) return rc;
    if (S) {
      // F1 = sum over ranks of the partial sums: all-gather + a mod-p add kernel (RCCL has no mod-p reduction); 16 N bytes per rank
      HIPC(hipMemcpyAsync(S->comm.send_buf, f1a, 2 * n * sizeof(u64), hipMemcpyDeviceToDevice, st));
      if ((rc = split_all_gather_device(P, 2 * n))) return rc;   // stream-ordered behind the copy
      hipLaunchKernelGGL(split_modadd_kernel, blocks(2 * n), dim3(256), 0, st, S->scratch, 2 * n, S->comm.world, f1a);
      HIPC(hipGetLastError());
    }
    HIPC(hipMemcpyAsync(f0a, f1a, 2 * n * sizeof(u64), hipMemcpyDeviceToDevice, st));
    if ((rc = combine(P->d_q, 4, w_q, f0a, f0b, 1))) return rc;
    HIPC(stream_wait(st));  // `wall` (pageable host memory) must outlive its upload
    // final_poly = alpha^(C+Z) * (F0 / (X - zeta)) + F1 / (X - g zeta), n-1 coefficients each; times X (a zero in front, plonky2
    // 0.1.x, sbn_config.fri_variant) or zero-padded at the end; then lde -> m
    HIPC(hipMemsetAsync(P->d_fcoef, 0, 2 * m * sizeof(u64), st));
    E2 shift2 = e2_pow(fri_alpha, C + Z);
    auto divide = [&](const u64* ca, const u64* cb, E2 z, E2 mul, int accumulate) {
      const size_t nch = n / DBL_CHUNK;  // a power of two >= 32
      u64 *ha = P->d_part, *hb = P->d_part + nch;
      hipLaunchKernelGGL(divide_by_linear_pass1, blocks(nch), dim3(256), 0, st, ca, cb, nch, z.a.v, z.b.v, ha, hb);
      hipLaunchKernelGGL(divide_by_linear_pass2, dim3(1), dim3(256), 0, st, ha, hb, nch, z.a.v, z.b.v);
      hipLaunchKernelGGL(divide_by_linear_pass3, blocks(nch), dim3(256), 0, st, ca, cb, nch, z.a.v, z.b.v, ha, hb, mul.a.v, mul.b.v, P->d_fcoef, P->d_fcoef + m, accumulate, fri_times_x(cfg));
    };
    divide(f0a, f0b, zeta, E2{F(0), F(0)}, 0);
    divide(f1a, f1b, zeta_next, shift2, 1);
    HIPC(hipGetLastError());
  }
  HIPC(hipEventRecord(P->ev[ST_FRI_LAYERS], st));
  std::vector<std::vector<u64>> fri_caps;
  std::vector<u64> final_poly;
  {
    // fri_committed_trees
    u32 bits = P->lde_log;
    u64* coef = P->d_fcoef; u64* coef_alt = P->d_fcoef2;
    size_t clen = m;  // coefficient vector length (planes at coef, coef + clen)
    F shift(GL_GEN);
    for (size_t li = 0; li < P->fri.arity_bits.size(); li++) {
      u32 ab = P->fri.arity_bits[li];
      u64* va = P->fri_vals[li]; u64* vb = va + clen;
      // values = coeffs.coset_fft(shift): scale by shift^i then NTT (2 base planes)
      HIPC(hipMemcpyAsync(va, coef, 2 * clen * sizeof(u64), hipMemcpyDeviceToDevice, st));
      hipLaunchKernelGGL(scale_pow_kernel, blocks(clen), dim3(256), 0, st, va, clen, shift.v);
      hipLaunchKernelGGL(scale_pow_kernel, blocks(clen), dim3(256), 0, st, vb, clen, shift.v);
      if ((rc = ntt_columns(P, va, clen, va, clen, P->d_tmp, m, 2, bits, false, clen, nullptr, nullptr, 1))) return rc;
      DevTree& t = P->fri_trees[li];
      if (t.nleaf <= 16384) hipLaunchKernelGGL(fri_leaf_hash_coop_kernel, blocks(t.nleaf * 16), dim3(256), 0, st, va, vb, bits, ab, t.d);
      else hipLaunchKernelGGL(fri_leaf_hash_kernel, blocks(t.nleaf), dim3(256), 0, st, va, vb, bits, ab, t.d);
      if ((rc = tree_build_inner(P, t, st))) return rc;
      std::vector<u64> cap;
      if ((rc = tree_cap_to_host(P, t, cap))) return rc;
      ch.observe_words(cap.data(), capw);
      fri_caps.push_back(cap);
      E2 beta = ch.ext_challenge();
      size_t nout = clen >> ab;
      hipLaunchKernelGGL(fri_fold_kernel, blocks(nout), dim3(256), 0, st, coef, coef + clen, nout, 1u << ab, beta.a.v, beta.b.v, coef_alt, coef_alt + nout);
      HIPC(hipGetLastError());
      std::swap(coef, coef_alt);
      clen = nout; bits -= ab;
      shift = f_pow(shift, (u64)1 << ab);
    }
    size_t fl = clen >> cfg.rate_bits;  // coefficients beyond are zero
    final_poly.resize(2 * fl);
    std::vector<u64> fa(fl), fb(fl);
    HIPC(hipMemcpyAsync(fa.data(), coef, fl * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPC(hipMemcpyAsync(fb.data(), coef + clen, fl * sizeof(u64), hipMemcpyDeviceToHost, st));
    HIPC(stream_wait(st));
    for (size_t i = 0; i < fl; i++) { final_poly[2 * i] = fa[i]; final_poly[2 * i + 1] = fb[i]; ch.observe(F(fa[i])); ch.observe(F(fb[i])); }
  }
  HIPC(hipEventRecord(P->ev[ST_POW], st));
  // fri_proof_of_work: smallest witness
  u64 pow_witness = ~0ULL;
  {
    PowParams pp{};
    for (int i = 0; i < 12; i++) pp.state[i] = ch.st[i].v;
    for (size_t i = 0; i < ch.in.size(); i++) pp.state[i] = ch.in[i].v;
    pp.wpos = (u32)ch.in.size(); pp.min_lz = cfg.proof_of_work_bits; pp.result = P->d_pow;
    // candidates are scanned in increasing windows (2^17, then 2^20 each): a 16-bit PoW is found in the
    // first window with probability 1 - e^-2; the smallest witness over the scanned prefix is kept.
    u64 base = 0;
    for (int round = 0; pow_witness == ~0ULL; round++) {
      const u64 BATCH = round == 0 ? (1ULL << 17) : (1ULL << 20);
      if (base >= GLP - BATCH) return fail(SBN_ERR_HIP, "proof of work failed");
      u64 init = ~0ULL;
      HIPC(hipMemcpyAsync(P->d_pow, &init, sizeof(u64), hipMemcpyHostToDevice, st));
      pp.base = base; pp.count = BATCH;
      hipLaunchKernelGGL(pow_kernel, blocks(BATCH), dim3(256), 0, st, pp);
      HIPC(hipMemcpyAsync(&pow_witness, P->d_pow, sizeof(u64), hipMemcpyDeviceToHost, st));
      HIPC(stream_wait(st));
      base += BATCH;
    }
    ch.observe(F(pow_witness));
    F resp = ch.challenge();
    u32 lz = resp.v ? (u32)__builtin_clzll(resp.v) : 64;
    if (lz < cfg.proof_of_work_bits) return fail(SBN_ERR_HIP, "proof-of-work witness check failed");
  }
  HIPC(hipEventRecord(P->ev[ST_QUERIES], st));
  // fri_prover_query_rounds
  const u32 nq = cfg.num_query_rounds;
  std::vector<u64> qwords(P->qstride * nq);
  std::vector<u32> idx(nq), idx_local(nq);  // pageable sources of async uploads: they live until the stream_wait below
  size_t split_section = 0;                 // the split: words per query that come from the row-sharded trees
  {
    for (u32 q = 0; q < nq; q++) idx[q] = (u32)(ch.challenge().v % m);
    HIPC(hipMemcpyAsync(P->d_idx, idx.data(), nq * sizeof(u32), hipMemcpyHostToDevice, st));
    size_t off = 0;
    const u32 nsib = P->lde_log - cfg.cap_height;
    // a row-sharded matrix (the split): `rows` local rows, leaf index inside this rank's subtrees; the rank that owns a
    // query's leaf holds its whole row and its Merkle path up to the cap (other ranks gather a row that is dropped below)
    auto initial = [&](const u64* mat, size_t ncols, const DevTree& t, size_t rows, u32 log_rows, const u32* d_index) {
      hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((ncols + 255) / 256), nq), dim3(256), 0, st, mat, rows, log_rows, (u32)ncols, d_index, P->d_qbuf, P->qstride, off);
      off += ncols;
      hipLaunchKernelGGL(gather_siblings_kernel, dim3(nq), dim3(128), 0, st, t.d, t.nleaf, nsib, d_index, 0u, P->d_qbuf, P->qstride, off);
      off += (size_t)nsib * 4;
    };
    if (S) {
      for (u32 q = 0; q < nq; q++) idx_local[q] = idx[q] & (u32)(S->ml - 1);
      HIPC(hipMemcpyAsync(S->d_idx_local, idx_local.data(), nq * sizeof(u32), hipMemcpyHostToDevice, st));
      initial(S->lde_l, C, P->tree_t, S->ml, P->lde_log - S->log_r, S->d_idx_local);
      if (Z) initial(S->zlde_l, Z, P->tree_z, S->ml, P->lde_log - S->log_r, S->d_idx_local);
      split_section = off;
    } else {
      initial(P->d_lde, C, P->tree_t, m, P->lde_log, P->d_idx);
      if (Z) initial(P->d_zlde, Z, P->tree_z, m, P->lde_log, P->d_idx);
    }
    initial(P->d_qlde, 4, P->tree_q, m, P->lde_log, P->d_idx);
    u32 bits = P->lde_log, shift = 0;
    for (size_t li = 0; li < P->fri.arity_bits.size(); li++) {
      u32 ab = P->fri.arity_bits[li];
      shift += ab;
      size_t len = (size_t)1 << bits;
      hipLaunchKernelGGL(gather_fri_leaf_kernel, dim3(nq), dim3(64), 0, st, P->fri_vals[li], P->fri_vals[li] + len, bits, ab, P->d_idx, shift, P->d_qbuf, P->qstride, off);
      off += 2 * ((size_t)1 << ab);
      const DevTree& t = P->fri_trees[li];
      hipLaunchKernelGGL(gather_siblings_kernel, dim3(nq), dim3(128), 0, st, t.d, t.nleaf, t.nlevels, P->d_idx, shift, P->d_qbuf, P->qstride, off);
      off += (size_t)t.nlevels * 4;
      bits -= ab;
    }
    HIPC(hipGetLastError());
    if (off != P->qstride) return fail(SBN_ERR_HIP, "internal: query layout mismatch (%zu vs %zu)", off, P->qstride);
    HIPC(hipMemcpyAsync(qwords.data(), P->d_qbuf, qwords.size() * sizeof(u64), hipMemcpyDeviceToHost, st));
  }
  HIPC(hipEventRecord(P->ev[ST_COUNT], st));
  HIPC(stream_wait(st));
  for (int i = 0; i < ST_COUNT; i++) HIPC(hipEventElapsedTime(&P->stage_ms[i], P->ev[i], P->ev[i + 1]));
  if (S) {
    // the trace / Z sections of every query come from the rank that owns the leaf: all-gather on the host and pick
    const u32 R = S->comm.world;
    std::vector<u64> mine((size_t)nq * split_section), all((size_t)R * nq * split_section);
    for (u32 q = 0; q < nq; q++) memcpy(mine.data() + (size_t)q * split_section, qwords.data() + (size_t)q * P->qstride, split_section * sizeof(u64));
    if ((rc = split_all_gather_host(P, mine.data(), all.data(), mine.size() * sizeof(u64)))) return rc;
    for (u32 q = 0; q < nq; q++) {
      const u32 owner = idx[q] >> (P->lde_log - S->log_r);
      memcpy(qwords.data() + (size_t)q * P->qstride, all.data() + ((size_t)owner * nq + q) * split_section, split_section * sizeof(u64));
    }
    float xms = 0;   // device time inside the exchanges (waiting for the peers' blocks included)
    for (size_t i = 0; i + 1 < S->tev_used; i += 2) { float t = 0; HIPC(hipEventElapsedTime(&t, S->tev[i], S->tev[i + 1])); xms += t; }
    P->stage_ms[ST_COUNT + EX_COMM_MS] = xms;
  }

  // assemble canonical proof words (layout: include/sbn.h) -----------------------------------------
  sbn_proof* pr = new sbn_proof();
  pr->degree_bits = P->degree_bits;
  std::vector<u64>& w = pr->words;
  w.reserve(12 + 3 * capw + (C + Z + 4) * 4 + fri_caps.size() * capw + qwords.size() + final_poly.size() + 1 + P->pi.size());
  u64 hdr[12] = {PROOF_MAGIC, P->degree_bits, C, Z, 4, P->pi.size(), cfg.cap_height, cfg.rate_bits, fri_caps.size(), cfg.fri_arity_bits,
                 final_poly.size() / 2, nq};
  w.insert(w.end(), hdr, hdr + 12);
  w.insert(w.end(), trace_cap.begin(), trace_cap.end());
  if (Z) w.insert(w.end(), z_cap.begin(), z_cap.end());
  w.insert(w.end(), q_cap.begin(), q_cap.end());
  for (size_t p = 0; p < C; p++) { w.push_back(open[4 * p]); w.push_back(open[4 * p + 1]); }          // local_values
  for (size_t p = 0; p < C; p++) { w.push_back(open2[4 * p + 2]); w.push_back(open2[4 * p + 3]); }    // next_values
  for (size_t p = C; p < C + Z; p++) { w.push_back(open[4 * p]); w.push_back(open[4 * p + 1]); }      // permutation_zs
  for (size_t p = C; p < C + Z; p++) { w.push_back(open2[4 * p + 2]); w.push_back(open2[4 * p + 3]); }  // permutation_zs_next
  for (size_t p = C + Z; p < C + Z + 4; p++) { w.push_back(open[4 * p]); w.push_back(open[4 * p + 1]); }  // quotient_polys
  for (auto& cap : fri_caps) w.insert(w.end(), cap.begin(), cap.end());
  w.insert(w.end(), qwords.begin(), qwords.end());
  w.insert(w.end(), final_poly.begin(), final_poly.end());
  w.push_back(pow_witness);
  w.insert(w.end(), P->pi.begin(), P->pi.end());
  *out = pr;
  return SBN_OK;
}

// ---- building blocks for parity tests ---------------------------------------------------------------
extern "C" int sbn_commit_values(const uint64_t* cols, size_t ncols, size_t n, uint32_t rate_bits, uint32_t cap_height, uint64_t* cap_out,
                                 uint64_t* coeffs_out, uint64_t* lde_out) {
  if (!cols || !cap_out || ncols == 0) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (rate_bits != 1 || n < 512 || (n & (n - 1))) return fail(SBN_ERR_UNSUPPORTED, "need rate_bits=1 and n a power of two >= 512");
  u32 lg = 0; while (((size_t)1 << lg) < n) lg++;
  // a throw-away prover-like context built on the G1_OP shape would waste memory; build a minimal one
  sbn_prover P{};
  sbn_standard_fast_config(&P.cfg); P.cfg.cap_height = cap_height;
  P.degree_bits = lg; P.lde_log = lg + 1; P.n = n; P.m = 2 * n; P.ntt_chunk = 64; P.device = g_device;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SBN_ERR_NO_DEVICE, "no HIP device available: no CPU fallback");
  HIPC(hipSetDevice(g_device));
  { int rc0 = ntt_fast_setup(); if (rc0) return rc0; }
  { std::string serr; if (!P.set.load(serr)) return fail(SBN_ERR_BAD_ARG, "%s", serr.c_str()); }
  P.fast_ntt = P.set.fast_ntt; P.ntt_xcd = P.set.ntt_xcd;
  // the transform kernels the prover itself picks at this size (the fused middle passes), so that the parity test of this
  // entry point covers them
  P.ntt_fused = P.fast_ntt && (lg == 16 || lg == 17) && P.set.ntt_fused;
  P.ntt_fused512 = P.fast_ntt && lg == 18 && P.set.ntt_split1024 && P.set.ntt_fused;
  HIPC(hipStreamCreate(&P.stream));
  u64 *d_vals = nullptr, *d_coef = nullptr, *d_lde = nullptr;
  int rc = 0;
  rc |= dmalloc(&d_vals, ncols * n); rc |= dmalloc(&d_coef, ncols * n); rc |= dmalloc(&d_lde, ncols * P.m); rc |= dmalloc(&P.d_tmp, 64 * P.m);
  if (P.ntt_fused || P.ntt_fused512) rc |= dmalloc(&P.d_tmp2, 64 * P.m);
  if (lg == 18 && P.set.ntt_split1024) rc |= dmalloc(&P.d_shift_odd, n);
  rc |= dmalloc(&P.d_tw_f, P.m); rc |= dmalloc(&P.d_tw_i, P.m); rc |= dmalloc(&P.d_shift, P.m);
  rc |= tree_alloc(P.tree_t, P.m, cap_height);
  if (!rc) {
    auto blocks = [](size_t k) { return dim3((unsigned)((k + 255) / 256)); };
    F w = f_root_of_unity(P.lde_log);
    hipLaunchKernelGGL(pow_table_kernel, blocks(P.m), dim3(256), 0, P.stream, P.d_tw_f, P.m, w.v);
    hipLaunchKernelGGL(pow_table_kernel, blocks(P.m), dim3(256), 0, P.stream, P.d_tw_i, P.m, f_inv(w).v);
    hipLaunchKernelGGL(pow_table_kernel, blocks(P.m), dim3(256), 0, P.stream, P.d_shift, P.m, (u64)GL_GEN);
    if (P.d_shift_odd) hipLaunchKernelGGL(shift_odd_table_kernel, blocks(n), dim3(256), 0, P.stream, P.d_shift_odd, n, P.d_shift, P.d_tw_f, 9u);
    if (hipMemcpy(d_vals, cols, ncols * n * sizeof(u64), hipMemcpyHostToDevice) != hipSuccess) rc = fail(SBN_ERR_HIP, "H2D failed");
  }
  for (size_t c0 = 0; !rc && c0 < ncols; c0 += P.ntt_chunk) rc = intt_then_lde_chunk(&P, d_vals, d_coef, d_lde, c0, std::min(P.ntt_chunk, ncols - c0));
  if (!rc && ncols <= 4) rc = tree_from_matrix(&P, P.tree_t, d_lde, ncols);
  if (!rc && ncols > 4) {  // same chunked sponge as the prover (chunks of 64 columns), on one stream
    rc = dmalloc(&P.d_sponge, 12 * P.m);
    for (size_t c0 = 0; !rc && c0 < ncols; c0 += 64) {
      size_t nc = std::min<size_t>(64, ncols - c0);
      hipLaunchKernelGGL(leaf_absorb_kernel, dim3((unsigned)((P.m + 255) / 256)), dim3(256), 0, P.stream, d_lde + c0 * P.m, P.m, P.lde_log, (u32)nc, P.d_sponge,
                         c0 == 0 ? 1 : 0, c0 + 64 >= ncols ? 1 : 0, P.tree_t.d, (c0 + 64 < ncols && ncols - (c0 + 64) >= 8) ? 1 : 0);
    }
    if (!rc) rc = tree_build_inner(&P, P.tree_t, P.stream);
  }
  std::vector<u64> cap;
  if (!rc) rc = tree_cap_to_host(&P, P.tree_t, cap);
  if (!rc) memcpy(cap_out, cap.data(), cap.size() * sizeof(u64));
  if (!rc && coeffs_out && hipMemcpy(coeffs_out, d_coef, ncols * n * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SBN_ERR_HIP, "D2H failed");
  if (!rc && lde_out && hipMemcpy(lde_out, d_lde, ncols * P.m * sizeof(u64), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(SBN_ERR_HIP, "D2H failed");
  for (u64* b : {d_vals, d_coef, d_lde, P.d_tmp, P.d_tmp2, P.d_shift_odd, P.d_tw_f, P.d_tw_i, P.d_shift, P.tree_t.d, P.d_sponge}) if (b) (void)hipFree(b);
  (void)hipStreamDestroy(P.stream);
  return rc;
}

extern "C" int sbn_poseidon_permute_batch(uint64_t* states, size_t count) {
  if (!states) return fail(SBN_ERR_BAD_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SBN_ERR_NO_DEVICE, "no HIP device available: no CPU fallback");
  HIPC(hipSetDevice(g_device));
  for (size_t i = 0; i < count * 12; i++) if (states[i] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "state word %zu is not canonical", i);
  u64* d = nullptr;
  HIPC(hipMalloc((void**)&d, count * 12 * sizeof(u64)));
  HIPC(hipMemcpy(d, states, count * 12 * sizeof(u64), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(poseidon_batch_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d, count);
  HIPC(hipMemcpy(states, d, count * 12 * sizeof(u64), hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return SBN_OK;
}

extern "C" int sbn_field_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, int mode) {
  if (!a || !b || !out) return fail(SBN_ERR_BAD_ARG, "null argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(SBN_ERR_NO_DEVICE, "no HIP device available: no CPU fallback");
  HIPC(hipSetDevice(g_device));
  if (count == 0) return SBN_OK;
  u64* d = nullptr;
  HIPC(hipMalloc((void**)&d, 3 * count * sizeof(u64)));
  hipError_t e = hipMemcpy(d, a, count * sizeof(u64), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(d + count, b, count * sizeof(u64), hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(field_mul_batch_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, 0, d, d + count, d + 2 * count, count, mode);
    e = hipMemcpy(out, d + 2 * count, count * sizeof(u64), hipMemcpyDeviceToHost);
  }
  (void)hipFree(d);
  if (e != hipSuccess) return fail(SBN_ERR_HIP, "sbn_field_mul_batch: %s", hipGetErrorString(e));
  return SBN_OK;
}

// The transcript's host permutation (sparse partial rounds) or, with use_definition != 0, the plain round-by-round
// form it must agree with.  No device involved: this is the Fiat-Shamir hasher of prove() / verify().
extern "C" int sbn_poseidon_permute_host(uint64_t* states, size_t count, int use_definition) {
  if (!states) return fail(SBN_ERR_BAD_ARG, "null argument");
  for (size_t i = 0; i < count * 12; i++) if (states[i] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "state word %zu is not canonical", i);
  for (size_t k = 0; k < count; k++) {
    F st[12];
    for (int i = 0; i < 12; i++) st[i] = F(states[12 * k + i]);
    if (use_definition) poseidon_permute_generic(st); else poseidon_permute(st);
    for (int i = 0; i < 12; i++) states[12 * k + i] = st[i].v;
  }
  return SBN_OK;
}

extern "C" int sbn_set_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(SBN_ERR_NO_DEVICE, "device %d not available", device);
  HIPC(hipSetDevice(device));
  t_device = device; g_default_device.store(device);
  return SBN_OK;
}
extern "C" int sbn_set_thread_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return fail(SBN_ERR_NO_DEVICE, "device %d not available", device);
  HIPC(hipSetDevice(device));
  t_device = device;
  return SBN_OK;
}
// The resolved switches of a prover as one line of key=value pairs (settings.hpp): what bench.py prints into `config`.
extern "C" int sbn_prover_describe(const sbn_prover* P, char* out, size_t cap) {
  if (!P || !out || cap == 0) return fail(SBN_ERR_BAD_ARG, "null argument");
  const Settings& s = P->set;
  static const char* const CH[] = {"host_pool", "device_lane", "device_wave"};
  const char* chain = P->chain_mode == 0 && tracegen_host_chains_vectorized() ? "host_pool_ifma_x8" : CH[P->chain_mode < 0 || P->chain_mode > 2 ? 0 : P->chain_mode];
  char buf[1024];
  snprintf(buf, sizeof buf,
           "abi=%d device=%d ntt_chunk=%zu fast_ntt=%d ntt_xcd=%d ntt_fused=%d ntt_sub=%zu ntt_streams=%d ntt_split1024=%d merkle_fuse=%d quotient_tail=%d "
           "curve_chains=%s host_threads=%u fq12_host_chain=%d fq12_row_kernel=%d range_check=%d range_async=%d perm_z=%d quotient_lookups=%d comm_timeout_s=%g experimental=%d ignored=[%s]",
           SBN_ABI_VERSION, P->device, P->ntt_chunk, (int)P->fast_ntt, (int)P->ntt_xcd, (int)(P->ntt_fused || P->ntt_fused512), P->ntt_sub, P->ntt_two_streams ? 2 : 1, P->d_shift_odd ? 1 : 0,
           (int)s.merkle_fuse, s.quotient_tail, chain, tracegen_host_threads(), (int)s.fq12_host_chain,
           (int)s.fq12_row_kernel, s.range_check, (int)s.range_async, s.perm_z, s.quotient_lookups, s.comm_timeout_s, (int)s.experimental, s.ignored.c_str());
  snprintf(out, cap, "%s", buf);
  return SBN_OK;
}
extern "C" int sbn_device_count(void) { int n = 0; return hipGetDeviceCount(&n) == hipSuccess ? n : 0; }

// ---- oversized-trace split: C ABI (include/sbn.h) -----------------------------------------------------------------------
struct sbn_split_prover { sbn_prover* P; };
extern "C" int sbn_split_exchange_bytes(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, uint32_t world, uint64_t* send_bytes,
                                        uint64_t* recv_bytes) {
  if (!air || !cfg || !send_bytes || !recv_bytes || world == 0) return fail(SBN_ERR_BAD_ARG, "null argument");
  AirShape as;
  if (!air_shape(air, cfg, as)) return fail(SBN_ERR_BAD_ARG, "unknown air kind / num_io");
  size_t sw;
  return split_sizes(as, degree_bits, cfg->rate_bits, world, SPLIT_BLOCK, send_bytes, recv_bytes, &sw, nullptr);
}
extern "C" int sbn_split_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, const sbn_comm* comm, sbn_split_prover** out) {
  if (!out || !comm) return fail(SBN_ERR_BAD_ARG, "null argument");
  *out = nullptr;
  sbn_prover* P = nullptr;
  int rc = create_ctx(air, cfg, degree_bits, comm, &P);
  if (rc) return rc;
  *out = new sbn_split_prover{P};
  return SBN_OK;
}
extern "C" void sbn_split_prover_destroy(sbn_split_prover* sp) {
  if (!sp) return;
  sbn_prover_destroy(sp->P);
  delete sp;
}
extern "C" int sbn_split_prover_generate_trace(sbn_split_prover* sp, const uint32_t* ios, size_t num_io, uint64_t* pi_out) {
  if (!sp) return fail(SBN_ERR_BAD_ARG, "null argument");
  return sbn_prover_generate_trace(sp->P, ios, num_io, pi_out);
}
extern "C" int sbn_split_prover_load_trace(sbn_split_prover* sp, const uint64_t* trace, const uint64_t* pi, size_t n_pi) {
  if (!sp) return fail(SBN_ERR_BAD_ARG, "null argument");
  return sbn_prover_load_trace(sp->P, trace, pi, n_pi);
}
extern "C" int sbn_split_prover_prove(sbn_split_prover* sp, sbn_proof** out) {
  if (!sp) return fail(SBN_ERR_BAD_ARG, "null argument");
  return sbn_prover_prove(sp->P, out);
}
extern "C" int sbn_split_prover_stage_times(const sbn_split_prover* sp, float* ms_out, int cap) {
  if (!sp) return 0;
  return sbn_prover_stage_times(sp->P, ms_out, cap);
}
