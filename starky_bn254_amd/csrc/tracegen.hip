// Host witness generation for the G1 / G2 tables: replaces G1ExpStark::generate_trace (and G2ExpStark's,
// src/curves/g2/exp.rs:271-342) /
// generate_public_inputs (src/curves/g1/exp.rs:255-327) and G1Stark::generate_trace
// (src/curves/g1/muladd.rs:481-546), which the reference runs serially with arkworks + num-bigint.
//
// Native design (independent of the test oracle's restatement):
//  * Fq in 4x64 Montgomery form; all instances advance in LOCKSTEP so the affine-slope divisions of
//    one row step are served by ONE field inversion (Montgomery's batch trick; binary extended GCD).
//  * new_x / new_y come from the curve formulas in Fq; the signed quotient of each modular gadget is
//    an exact division by p done limb-by-limb from the low end (Hensel), no big-integer division.
//  * pulse witnesses 1/(row - pos) are table lookups into inv[1..N) (one batch inversion per trace).
//  * range-check columns use a counting sort (values < 2^16) and the reference's merge
//    (src/utils/lookup.rs:60-111) to place unused table values.
#include "host_common.hpp"
#include <algorithm>
#include "bn254w.cuh"
#include "chains_ifma.hpp"
#include <thread>
#include <atomic>
#include <cstring>
#include <functional>
#include <chrono>
#include <mutex>
#include <condition_variable>
#include <unistd.h>

using namespace sbn;

namespace {
using namespace bnw;

struct Ctx { Fq one_m; Ctx() { one_m = fq_one(); } };
const Ctx& ctx() { static const Ctx c; return c; }

// Batch inversion of Montgomery-form elements (all nonzero), in place.
void batch_inv(std::vector<Fq>& v) {
  size_t n = v.size(); if (!n) return;
  std::vector<Fq> pre(n);
  Fq acc = ctx().one_m;
  for (size_t i = 0; i < n; i++) { pre[i] = acc; acc = mmul(acc, v[i]); }
  u64 s[4], si[4]; from_m(acc, s); inv_std(s, si);
  Fq inv = to_m(si);
  for (size_t i = n; i-- > 0;) { Fq t = v[i]; v[i] = mmul(inv, pre[i]); inv = mmul(inv, t); }
}

// batch inversion: E = 1 directly; E = 2 through the norms a^2 + b^2 (one Fq inversion for the whole batch)
// (Co<E>, cadd/csub/cmul/czero, conv16_fq2 and g2_output_row live in bn254w.cuh, shared with the device generator)
inline void cbatch_inv(std::vector<Co<1>>& v) { std::vector<Fq> t(v.size()); for (size_t i = 0; i < v.size(); i++) t[i] = v[i].c[0]; batch_inv(t); for (size_t i = 0; i < v.size(); i++) v[i].c[0] = t[i]; }
inline void cbatch_inv(std::vector<Co<2>>& v) {
  std::vector<Fq> nrm(v.size());
  for (size_t i = 0; i < v.size(); i++) nrm[i] = fadd(mmul(v[i].c[0], v[i].c[0]), mmul(v[i].c[1], v[i].c[1]));
  batch_inv(nrm);
  Fq z = {{0, 0, 0, 0}};
  for (size_t i = 0; i < v.size(); i++) { Fq a = mmul(v[i].c[0], nrm[i]), b = fsub(z, mmul(v[i].c[1], nrm[i])); v[i].c[0] = a; v[i].c[1] = b; }
}

// Persistent worker pool: witness generation issues several short parallel loops per trace, and spawning threads
// for each costs more than the loops themselves.  Workers sleep on a condition variable between loops; the calling
// thread takes part.  The pool is rebuilt after a fork (the child has none of the parent's threads).
class WorkerPool {
 public:
  static WorkerPool& get() {
    static std::mutex gm; static WorkerPool* pool = nullptr; static pid_t owner = 0;
    std::lock_guard<std::mutex> g(gm);
    if (!pool || owner != getpid()) { pool = new WorkerPool(); owner = getpid(); }  // a forked child leaks the stale one
    return *pool;
  }
  void run(size_t n, const std::function<void(size_t)>& f) {
    if (in_worker() || n <= 1 || workers_.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
    std::lock_guard<std::mutex> serial(run_m_);
    { std::lock_guard<std::mutex> lk(m_); job_ = &f; n_ = n; next_.store(0); active_ = (unsigned)workers_.size(); gen_++; }
    cv_.notify_all();
    for (size_t i; (i = next_.fetch_add(1)) < n;) f(i);
    std::unique_lock<std::mutex> lk(m_);
    done_.wait(lk, [&] { return active_ == 0; });
  }
 private:
  static bool& in_worker() { static thread_local bool w = false; return w; }
  WorkerPool() {
    unsigned nt = std::thread::hardware_concurrency(); if (nt == 0) nt = 4; if (nt > 64) nt = 64;
    const Settings set = Settings::from_env_or_default();   // SBN_HOST_THREADS (settings.hpp), read when the pool is first used
    if (set.host_threads) nt = (unsigned)set.host_threads;
    nthreads_ = nt;
    for (unsigned t = 1; t < nt; t++) workers_.emplace_back([this] { worker(); });
    for (auto& t : workers_) t.detach();  // they sleep until process exit; nothing to join at unload
  }
  void worker() {
    in_worker() = true;
    size_t seen = 0;
    for (;;) {
      std::unique_lock<std::mutex> lk(m_);
      cv_.wait(lk, [&] { return gen_ != seen; });
      seen = gen_;
      const std::function<void(size_t)>* f = job_; const size_t n = n_;
      lk.unlock();
      for (size_t i; (i = next_.fetch_add(1)) < n;) (*f)(i);
      lk.lock();
      if (--active_ == 0) done_.notify_one();
    }
  }
  std::vector<std::thread> workers_;
  unsigned nthreads_ = 1;
 public:
  unsigned threads() const { return nthreads_; }
 private:
  std::mutex m_, run_m_;
  std::condition_variable cv_, done_;
  const std::function<void(size_t)>* job_ = nullptr;
  size_t n_ = 0, gen_ = 0;
  std::atomic<size_t> next_{0};
  unsigned active_ = 0;
};
void parallel_for(size_t n, const std::function<void(size_t)>& f) { WorkerPool::get().run(n, f); }
unsigned pool_threads() { return WorkerPool::get().threads(); }   // the caller's thread included
// contiguous index ranges of `chunk` per task: consecutive rows written by one thread stay cache-friendly
// in the column-major trace
void parallel_for_chunks(size_t n, size_t chunk, const std::function<void(size_t, size_t)>& f) {
  size_t tasks = (n + chunk - 1) / chunk;
  parallel_for(tasks, [&](size_t t) { size_t a = t * chunk, b = a + chunk < n ? a + chunk : n; f(a, b); });
}

// permuted_cols (lookup.rs:60-111) for values < range (counting sort); table = 0..range-1 then range-1 repeated.
void permuted_cols_u(const u64* col, size_t n, u32 range, u64* sorted_out, u64* perm_table_out) {
  std::vector<u32> hist(range, 0);
  for (size_t i = 0; i < n; i++) hist[col[i]]++;
  { size_t k = 0; for (u32 v = 0; v < range; v++) for (u32 c = 0; c < hist[v]; c++) sorted_out[k++] = v; }
  auto table_at = [&](size_t j) -> u64 { return j < range ? j : range - 1; };
  std::vector<size_t> unused_inds; std::vector<u64> unused_vals;
  size_t i = 0, j = 0;
  while (j < n && i < n) {
    u64 iv = sorted_out[i], tv = table_at(j);
    if (iv > tv) { unused_vals.push_back(tv); j++; }
    else if (iv < tv) { if (!unused_vals.empty()) { perm_table_out[i] = unused_vals.back(); unused_vals.pop_back(); } else unused_inds.push_back(i); i++; }
    else { perm_table_out[i] = tv; i++; j++; }
  }
  for (; j < n; j++) unused_vals.push_back(table_at(j));
  for (; i < n; i++) unused_inds.push_back(i);
  for (size_t k = 0; k < unused_inds.size(); k++) perm_table_out[unused_inds[k]] = unused_vals[k];
}

inline void from_u32(const uint32_t* w, u64* out) { for (int i = 0; i < 4; i++) out[i] = (u64)w[2 * i] | ((u64)w[2 * i + 1] << 32); }
inline void put_limbs(u64* dst, size_t stride, const u64* v) { for (int i = 0; i < 16; i++) dst[i * stride] = (v[i / 4] >> (16 * (i % 4))) & 0xffff; }

// Goldilocks inverses of 1..n-1 (batch).
std::vector<u64> small_inverses(size_t n) {
  std::vector<u64> pre(n), inv(n);
  F acc(1);
  for (size_t k = 1; k < n; k++) { pre[k] = acc.v; acc = acc * F(k); }
  F ia = f_inv(acc);
  for (size_t k = n; k-- > 1;) { inv[k] = (ia * F(pre[k])).v; ia = ia * F(k); }
  inv[0] = 0;
  return inv;
}

}  // namespace

// flags columns (flags.rs:46-134) for every instance: closed form per row.  exp(k) = the 8 u32 limbs.
static void fill_flags(uint64_t* trace, size_t n, int sf, size_t num_io, const std::function<const uint32_t*(size_t)>& exp) {
  auto col = [&](int c) { return trace + (size_t)c * n; };
  const size_t RPB = 512;
  for (size_t k = 0; k < num_io; k++) {
    const uint32_t* e = exp(k);
    u64 limbs[8]; for (int i = 0; i < 8; i++) limbs[i] = e[i];
    u64 bit = limbs[0] & 1; limbs[0] >>= 1;  // first row
    for (size_t r = 0; r < RPB; r++) {
      size_t row = k * RPB + r;
      u64 a = r & 1, b = 1 - a;
      col(sf)[row] = r == RPB - 1; col(sf + 1)[row] = (r % 64) == 62;
      col(sf + 2)[row] = a; col(sf + 3)[row] = b; col(sf + 4)[row] = bit * b; col(sf + 5)[row] = bit;
      for (int i = 0; i < 8; i++) col(sf + 6 + i)[row] = limbs[i];
      bool split = a == 1, rotate = (r % 64) == 62;
      if (split) { bit = limbs[0] & 1; limbs[0] >>= 1; }
      if (rotate) { for (int i = 0; i < 7; i++) limbs[i] = limbs[i + 1]; limbs[7] = 0; }
    }
  }
}
// periodic pulse (pulse.rs:100-144: counter starts at 1, period 64, witness 1/(counter-63); absent in the u64 table) and
// io pulses (pulse.rs:20-43) at the first and last row of every instance
static void fill_pulses_at(uint64_t* trace, size_t n, int start_periodic, int start_io_pulses, size_t num_io, size_t RPB) {
  auto col = [&](int c) { return trace + (size_t)c * n; };
  std::vector<u64> inv = small_inverses(n);
  if (start_periodic >= 0) {
    u64* cnt = col(start_periodic); u64* wit = col(start_periodic + 1);
    for (size_t i = 0; i < n; i++) { u64 c = (i + 1) % 64; cnt[i] = c; wit[i] = c == 63 ? 0 : (-F(inv[63 - c])).v; }
  }
  u64* cnt2 = col(start_io_pulses);
  for (size_t i = 0; i < n; i++) cnt2[i] = i;
  parallel_for(2 * num_io, [&](size_t q) {
    size_t pos = (q >> 1) * RPB + ((q & 1) ? RPB - 1 : 0);
    u64* w = col(start_io_pulses + 1 + 2 * (int)q); u64* pul = col(start_io_pulses + 2 + 2 * (int)q);   // pulse.rs:10-16
    for (size_t i = 0; i < n; i++) w[i] = i > pos ? inv[i - pos] : (i < pos ? (-F(inv[pos - i])).v : 0);
    memset(pul, 0, n * sizeof(u64));
    pul[pos] = 1;
  });
}
static void fill_pulses(uint64_t* trace, size_t n, const ExpShape& sh) {
  fill_pulses_at(trace, n, sh.start_periodic, sh.start_io_pulses, (size_t)sh.num_io, (size_t)sh.rpb);
}
// split range check (range_check.rs:116-160): table 0..255 then 255; per target: lo, perm(lo), table', hi, perm(hi), table'
static bool fill_split_range_check(uint64_t* trace, size_t n, int table_col, int first_target, int num_targets) {
  auto col = [&](int c) { return trace + (size_t)c * n; };
  std::atomic<int> bad(0);
  u64* table = col(table_col);
  for (size_t i = 0; i < n; i++) table[i] = i < 256 ? i : 255;
  parallel_for((size_t)num_targets, [&](size_t k) {
    const u64* c = col(first_target + (int)k);
    int o = table_col + 1 + 6 * (int)k;
    for (size_t i = 0; i < n; i++) { if (c[i] >= 65536) { bad = 1; return; } col(o)[i] = c[i] & 0xff; col(o + 3)[i] = c[i] >> 8; }
    permuted_cols_u(col(o), n, 256, col(o + 1), col(o + 2));
    permuted_cols_u(col(o + 3), n, 256, col(o + 4), col(o + 5));
  });
  return !bad;
}

// Both curve chains of every instance on host threads (the sequential 0.1% of witness generation; the device does the
// rest, see sbn_prover_generate_trace).  ja / jb: [K][257][3][E][4] u64 (bn254w.cuh jac_at).
namespace sbn {
// form: 0 = eight instances per AVX-512 IFMA register when the CPU has it (chains_ifma.hpp), else one instance per task;
// 1 = always one instance per task (exp_chains); 2 = IFMA or fail.  Both forms write the same words.
int tracegen_host_chains(int E, const uint32_t* ios, size_t K, u64* ja, u64* jb, int form) {
  std::atomic<int> bad(0);
  const size_t iow = 8 * (4 * (size_t)E + 1);
#if defined(SBN_HAVE_IFMA)
  if (form != 1 && ifma::available()) {
    parallel_for((K + 7) / 8, [&](size_t g) {
      const size_t k0 = 8 * g; const int lanes = (int)std::min<size_t>(8, K - k0);
      const int b = E == 1 ? ifma::exp_chains_x8<1>(ios, k0, lanes, ja, jb) : ifma::exp_chains_x8<2>(ios, k0, lanes, ja, jb);
      if (b) bad |= b;
    });
    return bad.load();
  }
#endif
  if (form == 2) return -1;
  parallel_for(K, [&](size_t k) { int b = E == 1 ? exp_chains<1>(ios + iow * k, k, ja, jb) : exp_chains<2>(ios + iow * k, k, ja, jb); if (b) bad |= b; });
  return bad.load();
}
}  // namespace sbn

namespace sbn {
unsigned tracegen_host_threads() { return pool_threads(); }
bool tracegen_host_chains_vectorized() {
#if defined(SBN_HAVE_IFMA)
  return ifma::available();
#else
  return false;
#endif
}
}  // namespace sbn
// The SBN_* switches of the calling process as a prover created now would resolve them (no device needed): SBN_OK and one line
// of key=value pairs, or SBN_ERR_BAD_ARG naming the value that is not understood.
extern "C" int sbn_settings_check(char* out, size_t cap) {
  if (!out || cap == 0) return fail(SBN_ERR_BAD_ARG, "null argument");
  Settings s; std::string err;
  if (!s.load(err)) return fail(SBN_ERR_BAD_ARG, "%s", err.c_str());
  snprintf(out, cap, "host_threads=%u curve_chains_on_host=%s device_chain=%d comm_timeout_s=%g trace_timing=%d no_avx512=%d experimental=%d ntt_chunk=%d fast_ntt=%d ntt_xcd=%d "
           "ntt_fused=%d ntt_sub=%d ntt_streams=%d ntt_split1024=%d merkle_fuse=%d fq12_host_chain=%d fq12_row_kernel=%d quotient_tail=%d range_check=%d perm_z=%d quotient_lookups=%d ignored=[%s]",
           tracegen_host_threads(), tracegen_host_chains_vectorized() ? "ifma_x8" : "scalar", s.device_chain, s.comm_timeout_s, (int)s.trace_timing, (int)s.no_avx512, (int)s.experimental,
           s.ntt_chunk, (int)s.fast_ntt, (int)s.ntt_xcd, (int)s.ntt_fused, s.ntt_sub, s.ntt_streams, (int)s.ntt_split1024, (int)s.merkle_fuse, (int)s.fq12_host_chain,
           (int)s.fq12_row_kernel, s.quotient_tail, s.range_check, s.perm_z, s.quotient_lookups, s.ignored.c_str());
  return SBN_OK;
}
// Parity hook (include/sbn.h): the Jacobian chains of the curve witness as the device generator consumes them.
extern "C" int sbn_host_curve_chains(int E, const uint32_t* ios, size_t num_io, uint64_t* ja, uint64_t* jb, int form) {
  if (!ios || !ja || !jb || (E != 1 && E != 2) || num_io == 0 || form < 0 || form > 2) return fail(SBN_ERR_BAD_ARG, "bad arguments");
  const size_t iow = 8 * (4 * (size_t)E + 1);
  for (size_t k = 0; k < num_io; k++)
    for (int v = 0; v < 4 * E; v++) { u64 t[4]; u32x8_to_u64x4(ios + iow * k + 8 * v, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coordinate >= p (instance %zu)", k); }
  const int rc = tracegen_host_chains(E, ios, num_io, ja, jb, form);
  if (rc < 0) return fail(SBN_ERR_UNSUPPORTED, "this CPU has no AVX-512 IFMA");
  if (rc) return fail(SBN_ERR_WITNESS, "degenerate affine operation (x1 == x2 or y == 0)");
  return SBN_OK;
}

struct PhaseTimer {  // SBN_TRACE_TIMING=1 prints per-phase wall time to stderr
  bool on; std::chrono::steady_clock::time_point t;
  PhaseTimer() : on(Settings::from_env_or_default().trace_timing), t(std::chrono::steady_clock::now()) {}
  void lap(const char* name) {
    auto n = std::chrono::steady_clock::now();
    if (on) fprintf(stderr, "[tracegen] %-18s %8.3f s\n", name, std::chrono::duration<double>(n - t).count());
    t = n;
  }
};

template <int E>
static int generate_exp_trace(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  if (!ios || !trace || !pi_out || num_io == 0 || num_io > (size_t)G1EXP_MAX_IO) return fail(SBN_ERR_BAD_ARG, "bad arguments");
  const ExpShape sh(E, (int)num_io);
  const size_t RPB = 512, n = RPB * num_io;
  const size_t IOW = 8 * (4 * E + 1);  // u32 words per instance: x and offset (2E Fq each) + exp_val
  if (n < 65536) return fail(SBN_ERR_UNSUPPORTED, "the table needs >= 2^16 rows (u16 range check, range_check.rs:26)");
  const int sf = sh.start_flags, GC = sh.gadget_col, GW = 320 * E;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  PhaseTimer tm;  // every column is written in full below (pulse columns are zeroed in fill_pulses): no global memset
  fill_flags(trace, n, sf, num_io, [&](size_t k) { return ios + IOW * k + 32 * E; });
  tm.lap("flags");
  // --- phase 1: the double-and-add chain of every instance (g1/exp.rs:165-230, g2/exp.rs:180-246), one instance per
  //     task: 512 sequential steps, each with one field inversion (binary extended GCD; Fq2 through the norm).
  //     Per row the standard-form values ax ay bx by [lam nx ny] are kept for phase 2.
  typedef Co<E> C;
  const size_t SV = 28 * E;
  std::vector<u64> std_vals(n * SV);
  std::vector<unsigned char> row_op(n);  // 0 none, 1 add, 2 double
  std::vector<C> out_x(num_io), out_y(num_io);
  std::atomic<int> bad(0);
  for (size_t k = 0; k < num_io; k++)
    for (int v = 0; v < 4 * E; v++) { u64 t[4]; from_u32(ios + IOW * k + 8 * v, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coordinate >= p (instance %zu)", k); }
  parallel_for(num_io, [&](size_t k) {
    C ax, ay, bx, by, lam, nx, ny;
    C* dst[4] = {&ax, &ay, &bx, &by};
    for (int v = 0; v < 4; v++) for (int q = 0; q < E; q++) { u64 t[4]; from_u32(ios + IOW * k + 8 * (v * E + q), t); dst[v]->c[q] = to_m(t); }
    int prev = 0;
    std::vector<C> one(1);
    for (size_t r = 0; r < RPB; r++) {
      const size_t row = k * RPB + r;
      if (prev == 2) { ax = nx; ay = ny; } else if (prev == 1) { bx = nx; by = ny; }
      const bool dbl = r & 1;
      const int op = dbl ? 2 : (col(sf + 4)[row] ? 1 : 0);
      if (op) {
        C d = dbl ? cadd(ay, ay) : csub(bx, ax);
        if (czero(d)) { bad = 2; return; }
        one[0] = d; cbatch_inv(one);
        C num;
        if (dbl) { C x2 = cmul(ax, ax); num = cadd(cadd(x2, x2), x2); } else num = csub(by, ay);
        lam = cmul(num, one[0]);
        C l2 = cmul(lam, lam);
        nx = dbl ? csub(csub(l2, ax), ax) : csub(csub(l2, ax), bx);
        ny = csub(cmul(lam, csub(ax, nx)), ay);
      }
      u64* sv = &std_vals[row * SV];
      const C* src[7] = {&ax, &ay, &bx, &by, &lam, &nx, &ny};
      for (int v = 0; v < (op ? 7 : 4); v++) for (int q = 0; q < E; q++) from_m(src[v]->c[q], sv + 4 * (v * E + q));
      row_op[row] = (unsigned char)op;
      prev = op;
    }
    out_x[k] = bx; out_y[k] = by;  // b at the last row
  });
  tm.lap("curve chains");
  if (bad == 2) return fail(SBN_ERR_WITNESS, "degenerate affine operation (x1 == x2 or y == 0)");
  // --- phase 2: limb columns and the modular-gadget witnesses of every row, rows in contiguous chunks
  parallel_for_chunks(n, 64, [&](size_t r0, size_t r1) {
    u64 lv[640];
    for (size_t row = r0; row < r1; row++) {
      const u64* sv = &std_vals[row * SV];
      for (int v = 0; v < 4; v++) for (int q = 0; q < E; q++) put_limbs(col(16 * (v * E + q)) + row, n, sv + 4 * (v * E + q));
      const int op = row_op[row];
      if (op) {
        bool ok = E == 1 ? g1_output_row(op == 2, sv, sv + 4, sv + 8, sv + 12, sv + 16, sv + 20, sv + 24, lv)
                         : g2_output_row(op == 2, sv, sv + 8, sv + 16, sv + 24, sv + 32, sv + 40, sv + 48, lv);
        if (!ok) { bad = 1; return; }
      } else {  // G1Output::default / G2Output::default: zeros, quot signs = 1
        for (int i = 0; i < GW - 3 * E; i++) lv[i] = 0;
        for (int i = GW - 3 * E; i < GW; i++) lv[i] = 1;
      }
      for (int c = 0; c < GW; c++) col(GC + c)[row] = lv[c];
    }
  });
  tm.lap("row witnesses");
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  // --- public inputs: x, offset, exp_val, output as u32 limbs (g1/exp.rs:124-135, g2/exp.rs:139-156)
  for (size_t k = 0; k < num_io; k++) {
    u64* p = pi_out + (size_t)sh.pi_per_io * k;
    const int L = sh.L;
    for (int i = 0; i < 2 * L + 8; i++) p[i] = ios[IOW * k + i];
    const C* o[2] = {&out_x[k], &out_y[k]};  // b at the last row
    for (int v = 0; v < 2; v++) for (int q = 0; q < E; q++) {
      u64 t[4]; from_m(o[v]->c[q], t);
      for (int i = 0; i < 8; i++) p[2 * L + 8 + 8 * (v * E + q) + i] = (t[i / 2] >> (32 * (i % 2))) & 0xffffffffULL;
    }
  }
  fill_pulses(trace, n, sh);
  tm.lap("pulses");
  // --- u16 range check (range_check.rs:20-47)
  {
    u64* table = col(sh.start_lookups);
    for (size_t i = 0; i < n; i++) table[i] = i < 65536 ? i : 65535;
    parallel_for((size_t)sh.num_rc, [&](size_t k) {
      const u64* c = col((int)k);
      for (size_t i = 0; i < n; i++) if (c[i] >= 65536) { bad = 1; return; }
      permuted_cols_u(c, n, 65536, col(sh.start_lookups + 1 + 2 * (int)k), col(sh.start_lookups + 2 + 2 * (int)k));
    });
    if (bad) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  }
  tm.lap("range check");
  return SBN_OK;
}

// permuted_cols (lookup.rs:60-111) on arbitrary columns: sorted inputs, and the table permuted so that every looked-up value
// sits beside its first occurrence and unused table values fill the gaps (a LIFO pool while the merge runs, the rest in order).
static bool permuted_cols_any(const u64* inputs, const u64* table, size_t n, u64* sorted_out, u64* perm_table_out) {
  std::vector<u64> si(inputs, inputs + n), st(table, table + n);
  std::sort(si.begin(), si.end());
  std::sort(st.begin(), st.end());
  std::vector<size_t> unused_inds; std::vector<u64> unused_vals;
  size_t i = 0, j = 0;
  while (j < n && i < n) {
    if (si[i] > st[j]) { unused_vals.push_back(st[j]); j++; }
    else if (si[i] < st[j]) { if (!unused_vals.empty()) { perm_table_out[i] = unused_vals.back(); unused_vals.pop_back(); } else unused_inds.push_back(i); i++; }
    else { perm_table_out[i] = st[j]; i++; j++; }
  }
  for (; j < n; j++) unused_vals.push_back(st[j]);
  for (; i < n; i++) unused_inds.push_back(i);
  if (unused_inds.size() != unused_vals.size()) return false;
  for (size_t k = 0; k < unused_inds.size(); k++) perm_table_out[unused_inds[k]] = unused_vals[k];
  memcpy(sorted_out, si.data(), n * sizeof(u64));
  return true;
}
// MyStark::generate_trace (src/utils/lookup.rs:151-166)
extern "C" int sbn_generate_trace_lookup(const uint64_t* inputs, const uint64_t* table, size_t rows, uint64_t* trace) {
  if (!inputs || !table || !trace) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (rows < 2 || (rows & (rows - 1))) return fail(SBN_ERR_BAD_ARG, "rows must be a power of two");
  for (size_t i = 0; i < rows; i++) if (inputs[i] >= GLP || table[i] >= GLP) return fail(SBN_ERR_NON_CANONICAL, "value %zu is not canonical", i);
  memcpy(trace, inputs, rows * sizeof(u64));
  memcpy(trace + rows, table, rows * sizeof(u64));
  if (!permuted_cols_any(inputs, table, rows, trace + 2 * rows, trace + 3 * rows)) return fail(SBN_ERR_WITNESS, "permuted_cols: the merge does not close");
  return SBN_OK;
}
// FlagStark::generate_trace (src/utils/flags.rs:392-440): the flags of every input, the rotation pulse, the block-boundary pulses
extern "C" int sbn_generate_trace_flags(const uint32_t* limbs, size_t num_io, uint64_t* trace) {
  if (!limbs || !trace) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (num_io == 0 || num_io > (size_t)G1EXP_MAX_IO || (num_io & (num_io - 1))) return fail(SBN_ERR_BAD_ARG, "num_io must be a power of two <= %d", G1EXP_MAX_IO);
  const size_t n = 512 * num_io;
  const FlagShape sh((int)num_io);
  memset(trace, 0, (size_t)sh.num_cols() * n * sizeof(u64));
  fill_flags(trace, n, 0, num_io, [&](size_t k) { return limbs + 8 * k; });
  fill_pulses_at(trace, n, FlagShape::START_PERIODIC, FlagShape::START_IO_PULSES, num_io, 512);
  return SBN_OK;
}

// the FlagStark of src/fields/fq12_u64/flags_u64.rs:316-337: flags_u64 of every input (closed form: bit t of e is consumed on rows
// 2t, 2t+1), then the block-boundary pulses; no rotation pulse
extern "C" int sbn_generate_trace_flags_u64(const uint64_t* exps, size_t num_io, uint64_t* trace) {
  if (!exps || !trace) return fail(SBN_ERR_BAD_ARG, "null argument");
  if (num_io < 4 || num_io > 4 * (size_t)G1EXP_MAX_IO || (num_io & (num_io - 1))) return fail(SBN_ERR_BAD_ARG, "num_io must be a power of two between 4 and %d", 4 * G1EXP_MAX_IO);
  const size_t n = 128 * num_io;
  const FlagU64Shape sh((int)num_io);
  auto col = [&](int c) { return trace + (size_t)c * n; };
  memset(trace, 0, (size_t)sh.num_cols() * n * sizeof(u64));
  for (size_t k = 0; k < num_io; k++) {
    const u64 e = exps[k];
    if (e >= GLP) return fail(SBN_ERR_NON_CANONICAL, "exponent %zu is not a canonical field element", k);
    for (size_t r = 0; r < 128; r++) {
      const size_t row = k * 128 + r;
      const unsigned t = (unsigned)(r >> 1);
      const u64 fa = r & 1, fb = 1 - fa, bit = (e >> t) & 1, val = t == 63 ? 0 : e >> (t + 1);
      col(0)[row] = r == 127; col(1)[row] = fa; col(2)[row] = fb; col(3)[row] = bit * fb; col(4)[row] = bit; col(5)[row] = val;
    }
  }
  fill_pulses_at(trace, n, -1, FlagU64Shape::MAIN_COLS, num_io, 128);
  return SBN_OK;
}

extern "C" int sbn_generate_trace_g1_exp(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  return generate_exp_trace<1>(ios, num_io, trace, pi_out);
}
extern "C" int sbn_generate_trace_g2_exp(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  return generate_exp_trace<2>(ios, num_io, trace, pi_out);
}

extern "C" int sbn_generate_trace_g1_op(const uint32_t* pts, size_t rows, uint64_t* trace) {
  typedef G1OpShape S;
  if (!pts || !trace || rows < 256 || (rows & (rows - 1))) return fail(SBN_ERR_BAD_ARG, "rows must be a power of two >= 256");
  const size_t n = rows;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  std::vector<Fq> den(n);
  std::vector<u64> a(n * 16);  // ax ay bx by standard
  for (size_t r = 0; r < n; r++) {
    for (int q = 0; q < 4; q++) { from_u32(pts + 32 * r + 8 * q, &a[r * 16 + 4 * q]); if (geq_p(&a[r * 16 + 4 * q])) return fail(SBN_ERR_BAD_ARG, "coordinate >= p"); }
    den[r] = fsub(to_m(&a[r * 16 + 8]), to_m(&a[r * 16]));
    if (fzero(den[r])) return fail(SBN_ERR_WITNESS, "degenerate affine add at row %zu", r);
  }
  batch_inv(den);
  std::atomic<int> bad(0);
  parallel_for(n, [&](size_t r) {
    const u64* s = &a[r * 16];
    Fq x1 = to_m(s), y1 = to_m(s + 4), x2 = to_m(s + 8), y2 = to_m(s + 12);
    Fq lam = mmul(fsub(y2, y1), den[r]);
    Fq nx = fsub(fsub(mmul(lam, lam), x1), x2), ny = fsub(mmul(lam, fsub(x1, nx)), y1);
    u64 ls[4], nxs[4], nys[4]; from_m(lam, ls); from_m(nx, nxs); from_m(ny, nys);
    put_limbs(col(0) + r, n, s); put_limbs(col(16) + r, n, s + 4); put_limbs(col(32) + r, n, s + 8); put_limbs(col(48) + r, n, s + 12);
    u64 lv[320];
    if (!g1_output_row(false, s, s + 4, s + 8, s + 12, ls, nxs, nys, lv)) { bad = 1; return; }
    for (int c = 0; c < 320; c++) col(64 + c)[r] = lv[c];
    col(S::MAIN_COLS - 2)[r] = 1;  // is_add
    col(S::MAIN_COLS - 1)[r] = 0;  // is_double
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (!fill_split_range_check(trace, n, S::MAIN_COLS, S::START_RC, S::NUM_RC)) bad = 1;
  if (bad) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  return SBN_OK;
}

// ---- ModularStark (src/modular/modular.rs:383-437) and Fq12Stark (src/fields/fq12/mul.rs:375-419): one product per row ----
extern "C" int sbn_generate_trace_modular(const uint32_t* ops, size_t rows, uint64_t* trace) {
  const OpShape S(SBN_AIR_MODULAR);
  if (!ops || !trace || rows < 256 || (rows & (rows - 1))) return fail(SBN_ERR_BAD_ARG, "rows must be a power of two >= 256");
  const size_t n = rows;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  for (size_t r = 0; r < n; r++)
    for (int q = 0; q < 2; q++) { u64 t[4]; from_u32(ops + 16 * r + 8 * q, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "operand >= p (row %zu)", r); }
  std::atomic<int> bad(0);
  parallel_for(n, [&](size_t r) {
    u64 as[4], bs[4], ps[4], lv[112];
    from_u32(ops + 16 * r, as); from_u32(ops + 16 * r + 8, bs);
    from_m(mmul(to_m(as), to_m(bs)), ps);
    put_limbs(col(0) + r, n, as); put_limbs(col(16) + r, n, bs);
    if (!fq_output_row(as, bs, ps, lv)) { bad = 1; return; }
    for (int c = 0; c < 112; c++) col(32 + c)[r] = lv[c];   // output, aux, quot_sign
    col(S.main_cols - 1)[r] = 1;                            // filter
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (!fill_split_range_check(trace, n, S.main_cols, S.start_rc, S.num_rc)) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  return SBN_OK;
}
extern "C" int sbn_generate_trace_fq12_mul(const uint32_t* ops, size_t rows, uint64_t* trace) {
  const OpShape S(SBN_AIR_FQ12_MUL);
  if (!ops || !trace || rows < 256 || (rows & (rows - 1))) return fail(SBN_ERR_BAD_ARG, "rows must be a power of two >= 256");
  const size_t n = rows;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  for (size_t r = 0; r < n; r++)
    for (int q = 0; q < 24; q++) { u64 t[4]; from_u32(ops + 192 * r + 8 * q, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coefficient >= p (row %zu)", r); }
  std::atomic<int> bad(0);
  parallel_for(n, [&](size_t r) {
    Fq a[12], b[12], prod[12];
    u64 as[12][4], bs[12][4], ps[12][4];
    for (int c = 0; c < 12; c++) { from_u32(ops + 192 * r + 8 * c, as[c]); a[c] = to_m(as[c]); from_u32(ops + 192 * r + 96 + 8 * c, bs[c]); b[c] = to_m(bs[c]); }
    fq12_mul_m(a, b, prod);
    for (int c = 0; c < 12; c++) { from_m(prod[c], ps[c]); put_limbs(col(16 * c) + r, n, as[c]); put_limbs(col(192 + 16 * c) + r, n, bs[c]); }
    if (!fq12_output_row(as, bs, ps, [&](int i, u64 v) { col(384 + i)[r] = v; })) { bad = 1; return; }
    col(S.main_cols - 1)[r] = 1;   // filter
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  if (!fill_split_range_check(trace, n, S.main_cols, S.start_rc, S.num_rc)) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  return SBN_OK;
}

// ---- Fq12ExpStark (src/fields/fq12/exp.rs:229-319) --------------------------------------------------------------
// (fq12_mul_m and fq12_output_row live in bn254w.cuh, shared with the device generator)

// Square-and-multiply chains of every Fq12ExpStark instance on host threads, in standard form:
// A[t] = x^(2^t), B[0] = offset, B[t+1] = bit_t ? A[t] * B[t] : B[t]  (fq12/exp.rs:165-221: even rows multiply when the
// bit is set, odd rows square).  ca / cb: [K][steps+1][12][4] u64; steps = 256 (iow = 200) or, for the u64 table, 64 (iow = 194).
namespace sbn {
int tracegen_host_chains_fq12(const uint32_t* ios, size_t iow, int steps, size_t K, u64* ca, u64* cb) {
  parallel_for(K, [&](size_t k) {
    const uint32_t* io = ios + iow * k;
    Fq a[12], b[12], prod[12];
    for (int c = 0; c < 12; c++) { u64 t[4]; u32x8_to_u64x4(io + 8 * c, t); a[c] = to_m(t); u32x8_to_u64x4(io + 96 + 8 * c, t); b[c] = to_m(t); }
    for (int t = 0;; t++) {
      u64* pa = ca + ((k * (steps + 1) + t) * 12) * 4; u64* pb = cb + ((k * (steps + 1) + t) * 12) * 4;
      for (int c = 0; c < 12; c++) { from_m(a[c], pa + 4 * c); from_m(b[c], pb + 4 * c); }
      if (t == steps) break;
      if ((io[192 + (t >> 5)] >> (t & 31)) & 1) { fq12_mul_m(a, b, prod); memcpy(b, prod, sizeof b); }
      fq12_mul_m(a, a, prod); memcpy(a, prod, sizeof a);
    }
  });
  return 0;
}
}  // namespace sbn

extern "C" int sbn_generate_trace_fq12_exp(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  if (!ios || !trace || !pi_out || num_io == 0 || num_io > (size_t)G1EXP_MAX_IO || (num_io & (num_io - 1))) return fail(SBN_ERR_BAD_ARG, "bad arguments");
  const ExpShape sh(12, (int)num_io);
  const size_t RPB = 512, n = RPB * num_io, IOW = 200;
  const int sf = sh.start_flags;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  for (size_t k = 0; k < num_io; k++)
    for (int c = 0; c < 24; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coefficient >= p (instance %zu)", k); }
  fill_flags(trace, n, sf, num_io, [&](size_t k) { return ios + IOW * k + 192; });
  std::atomic<int> bad(0);
  parallel_for(num_io, [&](size_t k) {
    Fq a[12], b[12], prod[12];
    u64 as[12][4], bs[12][4], ps[12][4];
    for (int c = 0; c < 12; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); a[c] = to_m(t); from_u32(ios + IOW * k + 96 + 8 * c, t); b[c] = to_m(t); }
    int prev_op = 0;  // 0 none, 1 square (a <- prod), 2 multiply (b <- prod)
    std::vector<u64> lv(1344);
    for (size_t r = 0; r < RPB; r++) {
      size_t row = k * RPB + r;
      if (prev_op == 1) memcpy(a, prod, sizeof a); else if (prev_op == 2) memcpy(b, prod, sizeof b);
      for (int c = 0; c < 12; c++) { from_m(a[c], as[c]); from_m(b[c], bs[c]); put_limbs(col(16 * c) + row, n, as[c]); put_limbs(col(192 + 16 * c) + row, n, bs[c]); }
      int op = (r & 1) ? 1 : (col(sf + 4)[row] ? 2 : 0);  // odd rows square, even rows multiply when the bit is set
      if (op) {
        fq12_mul_m(a, op == 1 ? a : b, prod);
        for (int c = 0; c < 12; c++) from_m(prod[c], ps[c]);
        if (!fq12_output_row(as, op == 1 ? as : bs, ps, [&](int i, u64 v) { lv[i] = v; })) { bad = 1; return; }
      } else {  // Fq12Output::default (mul.rs:179-187)
        for (int i = 0; i < 1332; i++) lv[i] = 0;
        for (int i = 1332; i < 1344; i++) lv[i] = 1;
      }
      for (int c = 0; c < 1344; c++) col(384 + c)[row] = lv[c];
      prev_op = op;
    }
    // public inputs: x, offset as 16-bit limbs, exp_val, output = b at the last row (fq12/exp.rs:95-117)
    u64* p = pi_out + (size_t)sh.pi_per_io * k;
    for (int c = 0; c < 24; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); for (int i = 0; i < 16; i++) p[16 * c + i] = (t[i / 4] >> (16 * (i % 4))) & 0xffff; }
    for (int i = 0; i < 8; i++) p[384 + i] = ios[IOW * k + 192 + i];
    for (int c = 0; c < 12; c++) for (int i = 0; i < 16; i++) p[392 + 16 * c + i] = (bs[c][i / 4] >> (16 * (i % 4))) & 0xffff;
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  fill_pulses(trace, n, sh);
  if (!fill_split_range_check(trace, n, sh.start_lookups, sh.rc_start, sh.num_rc)) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  return SBN_OK;
}

// ---- FqExpStark (src/fields/fq/exp.rs:128-284) ---------------------------------------------------------------------
// (fq_output_row lives in bn254w.cuh, shared with the device generator)

// Square-and-multiply chains of every FqExpStark instance on host threads, standard form: A[t] = x^(2^t), B[0] = offset,
// B[t+1] = bit_t ? A[t] * B[t] : B[t].  ca / cb: [K][257][4] u64.
namespace sbn {
int tracegen_host_chains_fq(const uint32_t* ios, size_t K, u64* ca, u64* cb) {
  parallel_for_chunks(K, 8, [&](size_t k0, size_t k1) {
    for (size_t k = k0; k < k1; k++) {
      const uint32_t* io = ios + 24 * k;
      u64 t4[4];
      u32x8_to_u64x4(io, t4); Fq a = to_m(t4);
      u32x8_to_u64x4(io + 8, t4); Fq b = to_m(t4);
      for (int t = 0;; t++) {
        from_m(a, ca + (k * 257 + t) * 4); from_m(b, cb + (k * 257 + t) * 4);
        if (t == 256) break;
        if ((io[16 + (t >> 5)] >> (t & 31)) & 1) b = mmul(a, b);
        a = mmul(a, a);
      }
    }
  });
  return 0;
}
}  // namespace sbn

extern "C" int sbn_generate_trace_fq_exp(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  if (!ios || !trace || !pi_out || num_io == 0 || num_io > (size_t)G1EXP_MAX_IO || (num_io & (num_io - 1))) return fail(SBN_ERR_BAD_ARG, "bad arguments");
  const ExpShape sh(0, (int)num_io);
  const size_t RPB = 512, n = RPB * num_io, IOW = 24;
  if (n < 65536) return fail(SBN_ERR_UNSUPPORTED, "the table needs >= 2^16 rows (u16 range check, range_check.rs:26)");
  const int sf = sh.start_flags;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  for (size_t k = 0; k < num_io; k++)
    for (int c = 0; c < 2; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "value >= p (instance %zu)", k); }
  fill_flags(trace, n, sf, num_io, [&](size_t k) { return ios + IOW * k + 16; });
  std::atomic<int> bad(0);
  parallel_for(num_io, [&](size_t k) {
    u64 t[4];
    from_u32(ios + IOW * k, t); Fq a = to_m(t);
    from_u32(ios + IOW * k + 8, t); Fq b = to_m(t);
    Fq prod = a;
    u64 as[4], bs[4], ps[4], lv[112];
    int prev_op = 0;  // 0 none, 1 square (a <- prod), 2 multiply (b <- prod)
    for (size_t r = 0; r < RPB; r++) {
      size_t row = k * RPB + r;
      if (prev_op == 1) a = prod; else if (prev_op == 2) b = prod;
      from_m(a, as); from_m(b, bs);
      put_limbs(col(0) + row, n, as); put_limbs(col(16) + row, n, bs);
      int op = (r & 1) ? 1 : (col(sf + 4)[row] ? 2 : 0);  // odd rows square, even rows multiply when the bit is set
      if (op) {
        prod = mmul(a, op == 1 ? a : b);
        from_m(prod, ps);
        if (!fq_output_row(as, op == 1 ? as : bs, ps, lv)) { bad = 1; return; }
      } else {  // FqOutput::default (mul.rs:24-32)
        for (int i = 0; i < 111; i++) lv[i] = 0;
        lv[111] = 1;
      }
      for (int c = 0; c < 112; c++) col(32 + c)[row] = lv[c];
      prev_op = op;
    }
    // public inputs: x, offset, exp_val, output = b at the last row, as u32 limbs (fq/exp.rs:98-108)
    u64* p = pi_out + (size_t)sh.pi_per_io * k;
    for (int i = 0; i < 24; i++) p[i] = ios[IOW * k + i];
    for (int i = 0; i < 8; i++) p[24 + i] = (bs[i / 2] >> (32 * (i % 2))) & 0xffffffffULL;
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  fill_pulses(trace, n, sh);
  {
    u64* table = col(sh.start_lookups);
    for (size_t i = 0; i < n; i++) table[i] = i < 65536 ? i : 65535;
    parallel_for((size_t)sh.num_rc, [&](size_t k) {
      const u64* c = col((int)k);
      for (size_t i = 0; i < n; i++) if (c[i] >= 65536) { bad = 1; return; }
      permuted_cols_u(c, n, 65536, col(sh.start_lookups + 1 + 2 * (int)k), col(sh.start_lookups + 2 + 2 * (int)k));
    });
    if (bad) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  }
  return SBN_OK;
}

// ---- Fq12ExpU64Stark (src/fields/fq12_u64/exp_u64.rs:147-313, flags_u64.rs:34-94) -----------------------------------
extern "C" int sbn_generate_trace_fq12_exp_u64(const uint32_t* ios, size_t num_io, uint64_t* trace, uint64_t* pi_out) {
  if (!ios || !trace || !pi_out || num_io == 0 || num_io > (size_t)G1EXP_MAX_IO || (num_io & (num_io - 1))) return fail(SBN_ERR_BAD_ARG, "bad arguments");
  const ExpShape sh(13, (int)num_io);
  const size_t RPB = 128, n = RPB * num_io, IOW = 194;
  if (n < 512) return fail(SBN_ERR_UNSUPPORTED, "the prover needs >= 2^9 rows (num_io >= 4)");
  const int sf = sh.start_flags;
  auto col = [&](int c) { return trace + (size_t)c * n; };
  for (size_t k = 0; k < num_io; k++) {
    for (int c = 0; c < 24; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); if (geq_p(t)) return fail(SBN_ERR_BAD_ARG, "coefficient >= p (instance %zu)", k); }
    const u64 e = (u64)ios[IOW * k + 192] | ((u64)ios[IOW * k + 193] << 32);
    if (e >= GLP) return fail(SBN_ERR_NON_CANONICAL, "exponent of instance %zu is not a canonical field element", k);
  }
  std::atomic<int> bad(0);
  parallel_for(num_io, [&](size_t k) {
    const u64 e = (u64)ios[IOW * k + 192] | ((u64)ios[IOW * k + 193] << 32);
    Fq a[12], b[12], prod[12];
    u64 as[12][4], bs[12][4], ps[12][4];
    for (int c = 0; c < 12; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); a[c] = to_m(t); from_u32(ios + IOW * k + 96 + 8 * c, t); b[c] = to_m(t); }
    int prev_op = 0;  // 0 none, 1 square (a <- prod), 2 multiply (b <- prod)
    std::vector<u64> lv(1344);
    for (size_t r = 0; r < RPB; r++) {
      const size_t row = k * RPB + r;
      // flags (flags_u64.rs:34-94) in closed form: bit t of e is consumed on rows 2t, 2t+1
      const unsigned t = (unsigned)(r >> 1);
      const u64 fa = r & 1, fb = 1 - fa, bit = (e >> t) & 1, val = t == 63 ? 0 : e >> (t + 1);
      col(sf)[row] = r == RPB - 1; col(sf + 1)[row] = fa; col(sf + 2)[row] = fb; col(sf + 3)[row] = bit * fb; col(sf + 4)[row] = bit; col(sf + 5)[row] = val;
      if (prev_op == 1) memcpy(a, prod, sizeof a); else if (prev_op == 2) memcpy(b, prod, sizeof b);
      for (int c = 0; c < 12; c++) { from_m(a[c], as[c]); from_m(b[c], bs[c]); put_limbs(col(16 * c) + row, n, as[c]); put_limbs(col(192 + 16 * c) + row, n, bs[c]); }
      const int op = fa ? 1 : (bit ? 2 : 0);  // odd rows square, even rows multiply when the bit is set
      if (op) {
        fq12_mul_m(a, op == 1 ? a : b, prod);
        for (int c = 0; c < 12; c++) from_m(prod[c], ps[c]);
        if (!fq12_output_row(as, op == 1 ? as : bs, ps, [&](int i, u64 v) { lv[i] = v; })) { bad = 1; return; }
      } else {  // Fq12Output::default (mul.rs:179-187)
        for (int i = 0; i < 1332; i++) lv[i] = 0;
        for (int i = 1332; i < 1344; i++) lv[i] = 1;
      }
      for (int c = 0; c < 1344; c++) col(384 + c)[row] = lv[c];
      prev_op = op;
    }
    // public inputs: x, offset as 16-bit limbs, exp_val (one element), output = b at the last row (exp_u64.rs:100-128)
    u64* p = pi_out + (size_t)sh.pi_per_io * k;
    for (int c = 0; c < 24; c++) { u64 t[4]; from_u32(ios + IOW * k + 8 * c, t); for (int i = 0; i < 16; i++) p[16 * c + i] = (t[i / 4] >> (16 * (i % 4))) & 0xffff; }
    p[384] = e;
    for (int c = 0; c < 12; c++) for (int i = 0; i < 16; i++) p[385 + 16 * c + i] = (bs[c][i / 4] >> (16 * (i % 4))) & 0xffff;
  });
  if (bad) return fail(SBN_ERR_WITNESS, "modular witness generation failed");
  fill_pulses(trace, n, sh);
  if (!fill_split_range_check(trace, n, sh.start_lookups, sh.rc_start, sh.num_rc)) return fail(SBN_ERR_WITNESS, "range-checked column holds a value >= 2^16");
  return SBN_OK;
}

