// Every SBN_* environment switch of the library, read in ONE place.
//
// A prover reads the environment once, when it is created (create_ctx -> Settings::load), checks every value and keeps the
// resolved set for its lifetime (sbn_prover_describe prints it; bench.py copies it into `config`).  Two classes:
//   * production switches (host thread count, collective timeout, RCCL path, timing print, AVX-512 opt-out, chain placement of
//     the curve witness) are always honoured;
//   * EXPERIMENT switches (A/B forms of kernels and pipelines kept for measurement: every one of them yields the same proof
//     bytes, profiles/r3_v10_switch_parity.txt) are honoured only when SBN_EXPERIMENTAL=1 is set as well -- a stale export in
//     a production shell can then not change which kernels run or how long they take; without it they are reported as
//     ignored.
// A value that is not understood is an error (SBN_ERR_BAD_ARG), never a silent default.
#pragma once
#include <cstdlib>
#include <cstring>
#include <string>

namespace sbn {

struct Settings {
  // ---- production ----
  int host_threads = 0;          // SBN_HOST_THREADS: worker threads of the host pool (0 = one per visible CPU, at most 64)
  double comm_timeout_s = 600;   // SBN_COMM_TIMEOUT_S: deadline of every wait on another rank of a split proof
  std::string rccl_lib;          // SBN_RCCL_LIB: path of librccl (default: the loader's search path, then /opt/rocm/lib)
  bool trace_timing = false;     // SBN_TRACE_TIMING: print per-kernel times of the device witness generation
  bool no_avx512 = false;        // SBN_NO_AVX512: scalar host transcript permutation
  int device_chain = -1;         // SBN_TRACEGEN_DEVICE_CHAIN: -1 auto (host pool with >= 8 threads, else 2), 0 host pool,
                                 //   1 one lane per instance, 2 one wave per instance (tg::chain_coop_kernel)
  bool experimental = false;     // SBN_EXPERIMENTAL=1
  // ---- experiments (need SBN_EXPERIMENTAL=1) ----
  int ntt_chunk = 0;             // SBN_NTT_CHUNK: columns per commit-pipeline chunk (0 = size-dependent default)
  bool fast_ntt = true;          // SBN_FAST_NTT=0: generic radix-2 passes everywhere
  bool ntt_xcd = true;           // SBN_NTT_XCD=0: (tiles, columns) grid order of the register passes
  bool ntt_fused = true;         // SBN_NTT_FUSED=0: four separate passes per chunk at 2^16 / 2^17 rows
  int ntt_sub = 0;               // SBN_NTT_SUB: transform a chunk in sub-chunks of this many columns
  int ntt_streams = 0;           // SBN_NTT_STREAMS: 0 auto (two from 2^19 LDE rows up), 1, 2
  bool ntt_split1024 = true;     // SBN_NTT_SPLIT1024=0: generic first pass of the 2^19-point LDE
  bool merkle_fuse = true;       // SBN_MERKLE_FUSE=0: one launch per narrow Merkle level
  bool fq12_host_chain = false;  // SBN_FQ12_HOST_CHAIN=1: Fq12 square-and-multiply chains on the host pool
  bool fq12_row_kernel = false;  // SBN_FQ12_ROW_KERNEL=1: round 2's one lane per row
  int quotient_tail = 0;         // SBN_QUOTIENT_TAIL: 0, 1, 2 (placement of the AIR tail segment)
  int range_check = 0;           // SBN_RANGE_CHECK: 0 default, 1 = the round-3 kernel (one lane per 64 consecutive values)
  int perm_z = 0;                // SBN_PERM_Z=1: one workgroup per Z column, two sweeps (rounds 1-3) instead of the three chunk passes
  int quotient_lookups = 0;      // SBN_QUOTIENT_LOOKUPS=1: the lookup constraints of the u16 range check beside the permutation checks instead of in the AIR tail segment (measured slower)
  bool range_async = false;      // SBN_RANGE_ASYNC=1: the u16 range check of the curve witness behind generate_trace, beside the first half of the trace commitment (measured: no gain)
  int tracegen_skip = 0;         // SBN_TRACEGEN_SKIP: MEASUREMENT ONLY -- bit mask of curve-witness kernels NOT launched (1 flags / pulses,
                                 //   2 chains + affine_lambda, 4 gadget witness, 8 range check): the trace keeps what an earlier call wrote
  std::string ignored;           // experiment switches that were set without SBN_EXPERIMENTAL=1

  // Reads the environment.  false: `err` names the variable whose value is not understood.
  bool load(std::string& err) {
    auto get = [](const char* n) -> const char* { const char* v = getenv(n); return v && *v ? v : nullptr; };
    auto integer = [&](const char* n, long lo, long hi, long* out) -> bool {   // true: unset or valid
      const char* v = get(n);
      if (!v) return true;
      char* end = nullptr;
      const long x = strtol(v, &end, 10);
      if (!end || *end || x < lo || x > hi) { err = std::string(n) + "=" + v + " is not an integer in [" + std::to_string(lo) + ", " + std::to_string(hi) + "]"; return false; }
      *out = x;
      return true;
    };
    long x;
    x = 0; if (!integer("SBN_HOST_THREADS", 1, 256, &x)) return false; host_threads = (int)x;
    if (const char* v = get("SBN_COMM_TIMEOUT_S")) {
      char* end = nullptr;
      const double t = strtod(v, &end);
      if (!end || *end || !(t > 0)) { err = std::string("SBN_COMM_TIMEOUT_S=") + v + " is not a positive number of seconds"; return false; }
      comm_timeout_s = t;
    }
    if (const char* v = get("SBN_RCCL_LIB")) rccl_lib = v;
    trace_timing = get("SBN_TRACE_TIMING") != nullptr;
    no_avx512 = get("SBN_NO_AVX512") != nullptr;
    x = -1; if (!integer("SBN_TRACEGEN_DEVICE_CHAIN", 0, 2, &x)) return false; device_chain = (int)x;
    x = 0; if (!integer("SBN_EXPERIMENTAL", 0, 1, &x)) return false; experimental = x == 1;
    static const char* const EXP[] = {"SBN_NTT_CHUNK", "SBN_FAST_NTT", "SBN_NTT_XCD", "SBN_NTT_FUSED", "SBN_NTT_SUB", "SBN_NTT_STREAMS", "SBN_NTT_SPLIT1024",
                                      "SBN_MERKLE_FUSE", "SBN_FQ12_HOST_CHAIN", "SBN_FQ12_ROW_KERNEL", "SBN_QUOTIENT_TAIL", "SBN_RANGE_CHECK", "SBN_TRACEGEN_SKIP", "SBN_PERM_Z", "SBN_QUOTIENT_LOOKUPS", "SBN_RANGE_ASYNC"};
    if (!experimental) {
      for (const char* n : EXP) if (get(n)) { if (!ignored.empty()) ignored += ","; ignored += n; }
      return true;
    }
    x = 0; if (!integer("SBN_NTT_CHUNK", 8, 256, &x)) return false;
    if (x % 8) { err = "SBN_NTT_CHUNK must be a multiple of 8 between 8 and 256"; return false; }
    ntt_chunk = (int)x;
    x = 1; if (!integer("SBN_FAST_NTT", 0, 1, &x)) return false; fast_ntt = x != 0;
    x = 1; if (!integer("SBN_NTT_XCD", 0, 1, &x)) return false; ntt_xcd = x != 0;
    x = 1; if (!integer("SBN_NTT_FUSED", 0, 1, &x)) return false; ntt_fused = x != 0;
    x = 0; if (!integer("SBN_NTT_SUB", 8, 256, &x)) return false;
    if (x % 8) { err = "SBN_NTT_SUB must be a multiple of 8 between 8 and 256"; return false; }
    ntt_sub = (int)x;
    x = 0; if (!integer("SBN_NTT_STREAMS", 1, 2, &x)) return false; ntt_streams = (int)x;
    x = 1; if (!integer("SBN_NTT_SPLIT1024", 0, 1, &x)) return false; ntt_split1024 = x != 0;
    x = 1; if (!integer("SBN_MERKLE_FUSE", 0, 1, &x)) return false; merkle_fuse = x != 0;
    x = 0; if (!integer("SBN_FQ12_HOST_CHAIN", 0, 1, &x)) return false; fq12_host_chain = x != 0;
    x = 0; if (!integer("SBN_FQ12_ROW_KERNEL", 0, 1, &x)) return false; fq12_row_kernel = x != 0;
    x = 0; if (!integer("SBN_QUOTIENT_TAIL", 0, 2, &x)) return false; quotient_tail = (int)x;
    x = 0; if (!integer("SBN_RANGE_CHECK", 0, 1, &x)) return false; range_check = (int)x;
    x = 0; if (!integer("SBN_TRACEGEN_SKIP", 0, 15, &x)) return false; tracegen_skip = (int)x;
    x = 0; if (!integer("SBN_PERM_Z", 0, 1, &x)) return false; perm_z = (int)x;
    x = 0; if (!integer("SBN_QUOTIENT_LOOKUPS", 0, 1, &x)) return false; quotient_lookups = (int)x;
    x = 0; if (!integer("SBN_RANGE_ASYNC", 0, 1, &x)) return false; range_async = x != 0;
    return true;
  }
  // The environment of a caller that has no prover (library-level entry points): invalid values fall back to the defaults.
  static Settings from_env_or_default() { Settings s; std::string e; if (!s.load(e)) s = Settings(); return s; }
};

}  // namespace sbn
