// HIP kernels of the prover path (gfx950).  One header, included once by prover.hip.
// Each kernel states the reference stage it replaces, its access
// pattern and its algorithmic bytes.  All matrices are COLUMN-MAJOR [col][row] u64; LDE matrices
// are kept in NATURAL row order (row i = evaluation at 7*w^i); Merkle leaf index = bitrev(row).
#pragma once
#include "poseidon.cuh"
#include "air.cuh"

#include "kernels_ntt.cuh"   // K1: the transform kernels (their own translation unit, ntt.hip)

// table[i] = base^i (i < n); one thread per entry (square-and-multiply), used once per prover.
__global__ void pow_table_kernel(u64* out, size_t n, u64 base) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = f_pow(F(base), i).v;
}
// out[i] = shift[i] * w_m^((i >> log_s) << log_s), i < n = m/2: the input scale of the odd half of a split first pass
// (NttPassParams::pre2; tw = the forward table w_m^e, e < m/2).
__global__ void shift_odd_table_kernel(u64* out, size_t n, const u64* __restrict__ shift, const u64* __restrict__ tw, u32 log_s) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (F(shift[i]) * F(tw[(i >> log_s) << log_s])).v;
}
// v[i] *= base^i, for the small FRI layers.
__global__ void scale_pow_kernel(u64* v, size_t n, u64 base) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (F(v[i]) * f_pow(F(base), i)).v;
}

// =================================================================================================
// K2  Poseidon leaf hashing (MerkleTree::new leaves, `hash_or_noop`; P1/P2/P3)
// One thread per LDE row; lane l reads lde[c][row0+l] for every column c => each column read is a
// 512-byte coalesced segment per wave.  Sponge: overwrite rate lanes with 8 columns, permute.
// Digest stored at leaf index bitrev(row).  Algorithmic bytes: 8*M*C read + 32*M written.
// This kernel is ALU-bound (ceil(C/8) permutations per row), not HBM-bound.
// =================================================================================================
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void leaf_hash_kernel(const u64* __restrict__ lde, size_t m, u32 lde_log, u32 ncols,
                                                        u64* __restrict__ digests) {
  size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  F s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = F(0);
  size_t leaf = bitrev32((u32)row, lde_log);
  if (ncols <= 4) {  // hash_or_noop: digest = the elements, zero padded
    for (u32 c = 0; c < 4; c++) digests[leaf * 4 + c] = c < ncols ? lde[(size_t)c * m + row] : 0;
    return;
  }
  // one call site (the permutation is inlined once); a block that is followed by a FULL block only hands its capacity on
  for (u32 c = 0; c < ncols; c += 8) {
#pragma unroll
    for (u32 i = 0; i < 8; i++)
      if (c + i < ncols) s[i] = F(lde[(size_t)(c + i) * m + row]);
    poseidon_permute_keep(s, c + 8 >= ncols ? P_KEEP_DIGEST : (c + 16 <= ncols ? P_KEEP_CAPACITY : P_KEEP_ALL));
  }
#pragma unroll
  for (int i = 0; i < 4; i++) digests[leaf * 4 + i] = s[i].v;
}

// K3  Merkle inner level: parent[i] = two_to_one(child[2i], child[2i+1]).
__global__ __launch_bounds__(256) void merkle_level_kernel(const u64* __restrict__ child, u64* __restrict__ parent, size_t nparent) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nparent) return;
  F s[12];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = F(child[i * 8 + k]);
#pragma unroll
  for (int k = 8; k < 12; k++) s[k] = F(0);
  poseidon_permute_keep(s, P_KEEP_DIGEST);
#pragma unroll
  for (int k = 0; k < 4; k++) parent[i * 4 + k] = s[k].v;
}

// K2' chunked sponge absorption: the leaf hash of a wide matrix is a sequential sponge over its columns,
// so it can run per COLUMN CHUNK right behind the NTT that produced the chunk (second HIP stream): the
// LDE chunk is then read from the Infinity Cache instead of HBM and the HBM-bound NTT of chunk k+1
// overlaps the ALU-bound hashing of chunk k.  Sponge state: [12][m] words, column-major (coalesced).
// ncols_chunk is a multiple of 8 except for the last chunk; `first` starts from the zero state, `last`
// writes the digest to leaf bitrev(row).
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void leaf_absorb_kernel(const u64* __restrict__ lde_chunk, size_t m, u32 lde_log, u32 ncols_chunk,
                                                          u64* __restrict__ state, int first, int last, u64* __restrict__ digests, int next_full) {
  size_t row = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= m) return;
  F s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = F(0);
  if (!first) {
    // a launch that starts with a full block needs the capacity only (its first block overwrites state[0..7], and the launch before
    // stored nothing else: next_full); a launch of fewer than eight columns (the ragged end of the row) gets the whole state
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = F(state[(size_t)i * m + row]);
    if (ncols_chunk < 8) {
#pragma unroll
      for (int i = 0; i < 8; i++) s[i] = F(state[(size_t)i * m + row]);
    }
  }
  // Overwrite-mode sponge: a block of eight inputs replaces state[0..7], so a permutation that is followed by a FULL block only
  // hands its capacity state[8..11] on, and the last one of the row only its digest: the last matrix layer then computes four rows
  // instead of twelve (poseidon_permute_keep).  `next_full`: the next launch of this row starts with a full block (the host knows the
  // next chunk's width); a ragged block (the last of the row) keeps state[len..7], so the permutation before it keeps everything.
  // One call site: the permutation is inlined once.
  for (u32 c = 0; c < ncols_chunk; c += 8) {
#pragma unroll
    for (u32 i = 0; i < 8; i++)
      if (c + i < ncols_chunk) s[i] = F(lde_chunk[(size_t)(c + i) * m + row]);
    const bool end = c + 8 >= ncols_chunk;
    const int keep = end ? (last ? P_KEEP_DIGEST : (next_full ? P_KEEP_CAPACITY : P_KEEP_ALL)) : (c + 16 <= ncols_chunk ? P_KEEP_CAPACITY : P_KEEP_ALL);
    poseidon_permute_keep(s, keep);
  }
  // the store addresses are recomputed from an opaque copy of the lane index instead of being kept live across the permutations
  // (the kernel sits exactly at 128 VGPRs: two more live pairs would go to scratch)
  u32 tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const size_t row2 = (size_t)blockIdx.x * blockDim.x + tid;
  if (last) {
    size_t leaf = bitrev32((u32)row2, lde_log);
#pragma unroll
    for (int i = 0; i < 4; i++) digests[leaf * 4 + i] = s[i].v;
  } else {
    // with next_full only the capacity is meaningful, and only the capacity is stored and loaded again: 8 + 8 words per row and
    // launch less of carried state (the buffer layout [12][m] stays)
#pragma unroll
    for (int i = 8; i < 12; i++) state[(size_t)i * m + row2] = s[i].v;
    if (!next_full) {
#pragma unroll
      for (int i = 0; i < 8; i++) state[(size_t)i * m + row2] = s[i].v;
    }
  }
}

// Measured (tools/microbench/coop_latency.hip): one permutation takes 49 us on one lane and 16 us on 16 lanes, and a
// single wave already saturates its SIMD's issue slots, so more waves per SIMD only queue up: 256 lanes (one wave per
// SIMD) is the best workgroup size for the latency-bound levels (A/B against 512 and 1024 lanes on the G1 proof).
static constexpr u32 MERKLE_SUBTREE_THREADS = 256;
// K3' fused Merkle levels: a workgroup owns up to 512 consecutive digests of level `l0` and hashes
// up to `nlev` levels above them through LDS, writing every level to the tree (levels concatenated,
// level l at word offset 4*(2*nleaf - (2*nleaf >> l))).  Two launches build a 2^17-leaf tree.  Levels with few
// nodes per workgroup switch to 16 lanes per node (poseidon_permute_coop16): they are latency-bound.
__global__ __launch_bounds__(MERKLE_SUBTREE_THREADS) void merkle_subtree_kernel(u64* __restrict__ tree, size_t nleaf, u32 l0, u32 nlev, u32 nchild) {
  __shared__ u64 buf[2 * 256 * 4];
  const u32 tid = threadIdx.x, nt = blockDim.x;   // nchild = digests of level l0 owned by this workgroup (<= 512)
  const u64* child = tree + (2 * nleaf - ((2 * nleaf) >> l0)) * 4;
  size_t base = (size_t)blockIdx.x * nchild;
  for (u32 e = tid; e < nchild * 4; e += nt) buf[e] = child[base * 4 + e];
  __syncthreads();
  u32 active = nchild / 2;
  const u32 groups = nt / 16, grp = tid / 16, lane = tid & 15;
  for (u32 lv = 1; lv <= nlev; lv++) {
    u64* out = tree + (2 * nleaf - ((2 * nleaf) >> (l0 + lv))) * 4;
    const size_t idx0 = ((size_t)blockIdx.x * nchild) >> lv;
    if (active > 4 * groups) {
      // many nodes: one thread per parent
      F s[12];
      if (tid < active) {
#pragma unroll
        for (int k = 0; k < 8; k++) s[k] = F(buf[tid * 8 + k]);
#pragma unroll
        for (int k = 8; k < 12; k++) s[k] = F(0);
        poseidon_permute_keep(s, P_KEEP_DIGEST);
      }
      __syncthreads();
      if (tid < active) {
#pragma unroll
        for (int k = 0; k < 4; k++) { buf[tid * 4 + k] = s[k].v; out[(idx0 + tid) * 4 + k] = s[k].v; }
      }
      __syncthreads();
    } else {
      // few nodes: 16 lanes per parent (latency-bound part of the tree)
      u64 res[4];  // results of this lane's parents (written after all children have been read)
      u32 nit = (active + groups - 1) / groups;
      for (u32 it = 0; it < nit; it++) {
        u32 parent = it * groups + grp;
        if (it * groups + (tid / 64) * 4 >= active) break;  // wave-uniform: none of this wave's four groups has a parent
        const u32 e = lane < 12 ? lane : lane - 12;  // lanes 12..15 mirror elements 0..3
        u64 v = (parent < active && e < 8) ? buf[parent * 8 + e] : 0;
        res[it] = poseidon_permute_coop16(v, lane);
      }
      __syncthreads();
      for (u32 it = 0; it < nit; it++) {
        u32 parent = it * groups + grp;
        if (parent < active && lane < 4) { buf[parent * 4 + lane] = res[it]; out[(idx0 + parent) * 4 + lane] = res[it]; }
      }
      __syncthreads();
    }
    active >>= 1;
  }
}

// One Merkle level across the whole GPU, for the levels the fused kernel would keep inside a few workgroups:
// `merkle_level_thread_kernel` (one lane per parent) for wide levels, `merkle_level_coop_kernel` (16 lanes per parent,
// 16 parents per workgroup = one wave per SIMD) for the narrow, latency-bound ones.  l = level of the children.
__global__ __launch_bounds__(256) void merkle_level_thread_kernel(u64* __restrict__ tree, size_t nleaf, u32 l) {
  const size_t parents = nleaf >> (l + 1);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= parents) return;
  const u64* child = tree + (2 * nleaf - ((2 * nleaf) >> l)) * 4;
  u64* out = tree + (2 * nleaf - ((2 * nleaf) >> (l + 1))) * 4;
  F s[12];
#pragma unroll
  for (int k = 0; k < 8; k++) s[k] = F(child[i * 8 + k]);
#pragma unroll
  for (int k = 8; k < 12; k++) s[k] = F(0);
  poseidon_permute_keep(s, P_KEEP_DIGEST);
#pragma unroll
  for (int k = 0; k < 4; k++) out[i * 4 + k] = s[k].v;
}
__global__ __launch_bounds__(256) void merkle_level_coop_kernel(u64* __restrict__ tree, size_t nleaf, u32 l) {
  const size_t parents = nleaf >> (l + 1);
  const size_t parent = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const u32 lane = threadIdx.x & 15, e = lane < 12 ? lane : lane - 12;  // lanes 12..15 mirror elements 0..3
  const u64* child = tree + (2 * nleaf - ((2 * nleaf) >> l)) * 4;
  u64* out = tree + (2 * nleaf - ((2 * nleaf) >> (l + 1))) * 4;
  const u64 v = (parent < parents && e < 8) ? child[parent * 8 + e] : 0;
  const u64 r = poseidon_permute_coop16(v, lane);   // every lane of the row takes part (DPP rotations)
  if (parent < parents && lane < 4) out[parent * 4 + lane] = r;
}

// FRI layer leaves: leaf l = the 16 extension values at bit-reversed positions 16l..16l+15 of the
// layer's evaluation vector (planes va/vb in natural order), flattened c0,c1 (fri/prover.rs).
__global__ __launch_bounds__(256) void fri_leaf_hash_kernel(const u64* __restrict__ va, const u64* __restrict__ vb, u32 log_m, u32 arity_bits,
                                                            u64* __restrict__ digests) {
  size_t leaf = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t nleaf = (size_t)1 << (log_m - arity_bits);
  if (leaf >= nleaf) return;
  F s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = F(0);
  u32 arity = 1u << arity_bits;
  for (u32 t = 0; t < arity; t += 4) {
#pragma unroll
    for (u32 u = 0; u < 4; u++) {
      u32 nat = bitrev32((u32)(leaf * arity + t + u), log_m);
      s[2 * u] = F(va[nat]); s[2 * u + 1] = F(vb[nat]);
    }
    poseidon_permute_keep(s, t + 4 >= arity ? P_KEEP_DIGEST : P_KEEP_CAPACITY);
  }
#pragma unroll
  for (int i = 0; i < 4; i++) digests[leaf * 4 + i] = s[i].v;
}

// The same with 16 lanes per leaf (poseidon_permute_coop16): a FRI layer has few leaves and four dependent permutations
// per leaf, so the one-lane-per-leaf form is bound by the latency of a permutation (34 us) rather than by work.
__global__ __launch_bounds__(256) void fri_leaf_hash_coop_kernel(const u64* __restrict__ va, const u64* __restrict__ vb, u32 log_m, u32 arity_bits,
                                                                 u64* __restrict__ digests) {
  const size_t leaf = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const size_t nleaf = (size_t)1 << (log_m - arity_bits);
  const u32 lane = threadIdx.x & 15, e = lane < 12 ? lane : lane - 12;  // lanes 12..15 mirror elements 0..3
  const u32 arity = 1u << arity_bits;
  u64 st = 0;
  for (u32 t = 0; t < arity; t += 4) {
    if (leaf < nleaf && e < 8) {
      const u32 nat = bitrev32((u32)(leaf * arity + t + (e >> 1)), log_m);
      st = (e & 1) ? vb[nat] : va[nat];
    }
    st = poseidon_permute_coop16(st, lane);   // every lane of the row takes part (DPP rotations)
  }
  if (leaf < nleaf && lane < 4) digests[leaf * 4 + lane] = st;
}

// test hook: the device multiply on arbitrary 64-bit representatives (sbn_field_mul_batch)
__global__ void field_mul_batch_kernel(const u64* a, const u64* b, u64* out, size_t count, int mode) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
#if defined(__HIP_DEVICE_COMPILE__)
  out[i] = mode ? nw::canon(nw::canon(nw::mul(a[i], b[i]))) : (F(a[i]) * F(b[i])).v;
#endif
}
__global__ void poseidon_batch_kernel(u64* states, size_t count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  F s[12];
  for (int k = 0; k < 12; k++) s[k] = F(states[i * 12 + k]);
  poseidon_permute(s);
  for (int k = 0; k < 12; k++) states[i * 12 + k] = s[k].v;
}

// =================================================================================================
// K4  permutation Z columns (starky permutation.rs `compute_permutation_z_poly`; P2)
// One workgroup per Z column.  Z[i] = prod_{k<i} num_k / den_k with num = (lhs+g0)(lhs+g1),
// den = (rhs+g0)(rhs+g1).  Computed as  prefix(num)[i] * suffix(den)[i] / prod(den): two coalesced
// block-scan sweeps and ONE field inversion per column (exact arithmetic => identical to the
// reference's batch-inverse + running product).  Algorithmic bytes: reads 2 trace columns twice,
// writes/reads/writes Z once: 8*N*(4+3) per column.
// =================================================================================================
struct PairCols { int lhs, rhs; };

__device__ __forceinline__ F wave_scan_mul(F x, int lane) {  // inclusive prefix product within a wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u64 o = __shfl_up((unsigned long long)x.v, d, 64);
    if (lane >= d) x = x * F(o);
  }
  return x;
}

// Every lane takes PZ_E consecutive rows per sweep step (a serial product inside the lane, then one wave scan over the
// lanes' totals), which amortises the scan and the two barriers over PZ_E rows.  n must be a multiple of 256 * PZ_E.
// The kernel moves 7 words per row (lhs, rhs twice, Z written, read and written) = 2.8 GB for G1ExpStark(128) in 0.53 ms: it runs
// at HBM speed, not at the latency of its chain of scan steps.  Measured and dropped in round 4 (profiles/r4_perm_z_experiments.txt):
// loading the rows of step k + 1 in front of the barriers of step k (0.53 -> 0.66 ms: sixteen more live 64-bit values per lane),
// and four workgroups per column in two passes with a fix-up by the segment products (0.54 -> 0.58 ms: the same traffic).
template <int PZ_E>
__global__ __launch_bounds__(256) void permutation_z_kernel(const u64* __restrict__ trace, size_t n, const PairCols* __restrict__ pairs,
                                                            u64 gamma0, u64 gamma1, u64* __restrict__ zout) {
  __shared__ u64 wtot[4];
  __shared__ u64 carry_s;
  const int z = blockIdx.x;
  const u64* lhs = trace + (size_t)pairs[z].lhs * n;
  const u64* rhs = trace + (size_t)pairs[z].rhs * n;
  u64* zc = zout + (size_t)z * n;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const F g0(gamma0), g1(gamma1);
  // forward sweep: exclusive prefix product of num -> zc ; running product of den
  F carry(1), dprod(1);
  for (size_t base = 0; base < n; base += 256 * PZ_E) {
    const size_t i0 = base + (size_t)tid * PZ_E;
    u64 lv[PZ_E], rv[PZ_E];
#pragma unroll
    for (int e = 0; e < PZ_E; e++) { lv[e] = lhs[i0 + e]; rv[e] = rhs[i0 + e]; }
    F pr[PZ_E];   // inclusive products of num inside the lane
#pragma unroll
    for (int e = 0; e < PZ_E; e++) {
      const F l(lv[e]), r(rv[e]);
      const F num = (l + g0) * (l + g1);
      pr[e] = e ? pr[e - 1] * num : num;
      dprod = dprod * ((r + g0) * (r + g1));
    }
    F inc = wave_scan_mul(pr[PZ_E - 1], lane);
    if (lane == 63) wtot[wv] = inc.v;
    __syncthreads();
    F pre = carry;
    for (int w = 0; w < wv; w++) pre = pre * F(wtot[w]);
    const u64 prev = __shfl_up((unsigned long long)inc.v, 1, 64);
    const F excl = lane == 0 ? pre : pre * F(prev);   // product of everything before this lane's first row
    zc[i0] = excl.v;
#pragma unroll
    for (int e = 1; e < PZ_E; e++) zc[i0 + e] = (excl * pr[e - 1]).v;
    carry = carry * F(wtot[0]) * F(wtot[1]) * F(wtot[2]) * F(wtot[3]);
    __syncthreads();
  }
  // total den product -> inverse
  {
    F w = dprod;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) w = w * F(__shfl_xor((unsigned long long)w.v, d, 64));
    if (lane == 0) wtot[wv] = w.v;
    __syncthreads();
    if (tid == 0) carry_s = f_inv(F(wtot[0]) * F(wtot[1]) * F(wtot[2]) * F(wtot[3])).v;
    __syncthreads();
  }
  // backward sweep: inclusive suffix product of den, times 1/prod(den); lane t takes rows base-1-t*PZ_E-e, scanning from the top
  carry = F(carry_s);
  __syncthreads();
  for (size_t base = n; base > 0; base -= 256 * PZ_E) {
    const size_t i0 = base - 1 - (size_t)tid * PZ_E;
    u64 rv[PZ_E], zv[PZ_E];
#pragma unroll
    for (int e = 0; e < PZ_E; e++) { rv[e] = rhs[i0 - e]; zv[e] = zc[i0 - e]; }
    F pr[PZ_E];   // inclusive products of den inside the lane, from the top
#pragma unroll
    for (int e = 0; e < PZ_E; e++) {
      const F r(rv[e]);
      const F den = (r + g0) * (r + g1);
      pr[e] = e ? pr[e - 1] * den : den;
    }
    F inc = wave_scan_mul(pr[PZ_E - 1], lane);
    if (lane == 63) wtot[wv] = inc.v;
    __syncthreads();
    F pre = carry;
    for (int w = 0; w < wv; w++) pre = pre * F(wtot[w]);
    const u64 prev = __shfl_up((unsigned long long)inc.v, 1, 64);
    const F excl = lane == 0 ? pre : pre * F(prev);
#pragma unroll
    for (int e = 0; e < PZ_E; e++) zc[i0 - e] = (F(zv[e]) * (excl * pr[e])).v;   // = Dinv * prod_{k>=i} den_k
    carry = carry * F(wtot[0]) * F(wtot[1]) * F(wtot[2]) * F(wtot[3]);
    __syncthreads();
  }
}

// =================================================================================================
// K5  constraint / quotient evaluation (starky prover.rs `compute_quotient_polys`; P3)
// One thread per LDE point i; local row = i, next row = (i + 2^quotient_degree_bits) mod M.
// Lanes are consecutive points, so every column access is a coalesced 512-byte wave segment and the
// "next" access re-hits the same lines.  Output: quotient values acc_j / Z_H(x_i) for j < 2.
// Algorithmic bytes: 8*M*(C + Zc) read once, 16*M written.
// =================================================================================================
// `base` holds the local rows, `nbase` the next rows (the same matrix on one GPU; in the oversized-trace split a rank
// holds the LDE rows i = j * R + rho of its Merkle subtrees and, from R = 4 ranks up, a second plane with the rows i + 2).
struct DevRow {
  const u64* base; const u64* nbase; size_t m; size_t i, inext;
  __device__ __forceinline__ F l(int c) const { return F(base[(size_t)c * m + i]); }
  __device__ __forceinline__ F n(int c) const { return F(nbase[(size_t)c * m + inext]); }
};
struct DevZRow {
  const u64* base; const u64* nbase; size_t m; size_t i, inext;
  __device__ __forceinline__ F zl(int z) const { return F(base[(size_t)z * m + i]); }
  __device__ __forceinline__ F zn(int z) const { return F(nbase[(size_t)z * m + inext]); }
};
struct QuotientParams {
  const u64* lde; const u64* zlde; size_t m; u32 next_step;
  // Row sharding (all zero / equal to lde, zlde on one GPU): m = LOCAL point count, local point j is LDE point
  // (j << row_shift) | row_rho (tables xs / lag_* / zh_inv are indexed by the LDE point), its next row is local row
  // (j + next_step) mod m of lde_next / zlde_next.
  const u64* lde_next; const u64* zlde_next; u32 row_shift, row_rho;
  u32 seg_mask;     // segments whose kernels are launched (all four; a diagnostic switch for per-segment counter passes)
  const u64* xs; const u64* lag_first; const u64* lag_last;  // per LDE point
  u64 zh_inv[2];   // 1/Z_H on the two residues of i mod 2
  u64 last;        // g^-1
  u64 alpha[SBN_NCH];
  const u64* apow[SBN_NCH];
  u64 gamma0, gamma1;
  int num_zs, num_io;
  const void* pic;  // ExpPiConsts<F>*
  u64* qout;        // [SBN_NCH][m]
  u64* part;        // [QSEG segments][SBN_NCH][m] partial accumulators
  u64 seg_shift[4][SBN_NCH];  // alpha_j^(number of constraints that follow the segment)
  int seg_count[4];           // constraints of each segment (its first one is weighted alpha^(count-1) inside the segment)
  int zsplit;       // both permutation constraints of the Z columns [0, zsplit) go with segment 2, those of [zsplit, num_zs) with segment 3
  int lookups_in_perm;   // u16-range-check tables: the lookup constraints go with the permutation segments (their columns are loaded there anyway)
};

// The constraint stream is one Horner sum in alpha, so it splits exactly into four segments: 0 = AIR sections [1]-[8]
// (public inputs, transitions, flags, the add / double gadget), 1 = AIR sections [9]-[10] (io pulses, range check),
// 2 / 3 = the permutation checks of the first / second half of the Z columns (first-row constraint and transition of a column together);
// quotient_combine_kernel joins them as sum_s acc_s * alpha^(constraints after segment s).  With one lane per LDE point
// there are only two waves per SIMD at 2^17 points and one long dependent chain per lane (2.65 ms); four segments in
// flight give eight waves and quarter the chain (1.56 ms).  Each PART is its own kernel (0: segment 0, 1: segment 1,
// 2: segments 2 and 3 on grid.y) so that it gets its own register allocation -- as one kernel the gadget code of segment
// 0 set the VGPR count, and with it the occupancy, of the permutation checks too -- and the prover launches PART 0 + 1
// on its main stream and PART 2 on its second stream, so the parts still overlap.
// Tried and measured (profiles/r2_quotient_ab.txt): an XCD-aware order that runs the four segments of a 256-point block
// back to back on one XCD: traffic 7.50 -> 6.81 GB per launch but 1.56 -> 2.13 ms (tail imbalance); the re-reads were
// not between segments but inside segment 0 (3.0 GB for 0.45 GB of columns: every limb re-read for each convolution
// coefficient it feeds), which the factored gadgets of air.cuh removed (profiles/r2_quotient_segments.txt).
static constexpr u32 QSEG = 4;
// The alpha-power tables and the public-input constants come in as `const __restrict__` kernel arguments of their own (not
// inside the parameter struct): only then does the compiler know that the kernel never writes them and fetches the
// uniformly indexed entries with SCALAR loads (s_load through the scalar cache, operand straight into the multiply-add);
// through the struct they were vector loads + v_readfirstlane with a memory latency in front of every term.
template <int KIND, int PART>
__global__ __launch_bounds__(256, 2) void quotient_kernel(QuotientParams p, const u64* __restrict__ apow0, const u64* __restrict__ apow1,
                                                          const void* __restrict__ pic_arg) {   // at least two waves per SIMD: at most 256 VGPRs
  const u32 seg = PART == 2 ? 2 + blockIdx.y : (u32)PART;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.m || !((p.seg_mask >> seg) & 1)) return;
  const size_t inext = (i + p.next_step) & (p.m - 1);
  const size_t ig = (i << p.row_shift) | p.row_rho;   // LDE point of local row i
  Cons<F> cs;
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) cs.alpha[j] = F(p.alpha[j]);
  cs.apow[0] = (const F*)apow0; cs.apow[1] = (const F*)apow1;
  cs.start(p.seg_count[seg]);
  cs.z_last = F(p.xs[ig]) - F(p.last);
  cs.l_first = F(p.lag_first[ig]);
  cs.l_last = F(p.lag_last[ig]);
  DevRow row{p.lde, p.lde_next, p.m, i, inext};
  DevZRow zrow{p.zlde, p.zlde_next, p.m, i, inext};
  if (KIND == 1) {
    if (PART == 0) g1op_eval(cs, row);
    else if (PART == 2) {
      if (seg == 2) permutation_checks(cs, row, zrow, G1OpShape(), p.num_zs, F(p.gamma0), F(p.gamma1), 0, p.zsplit);
      else permutation_checks(cs, row, zrow, G1OpShape(), p.num_zs, F(p.gamma0), F(p.gamma1), p.zsplit, p.num_zs);
    }
  } else if (KIND == 9) {   // MyStark: two lookup constraints, two permutation pairs
    if (PART == 0) lookup_eval(cs, row);
    else if (PART == 2) {
      if (seg == 2) permutation_checks(cs, row, zrow, LookupShape(), p.num_zs, F(p.gamma0), F(p.gamma1), 0, p.zsplit);
      else permutation_checks(cs, row, zrow, LookupShape(), p.num_zs, F(p.gamma0), F(p.gamma1), p.zsplit, p.num_zs);
    }
  } else if (KIND == 10) {  // FlagStark: no permutation pairs, segments 1-3 are empty
    if (PART == 0) flag_eval(cs, row, FlagShape(p.num_io));
  } else if (KIND == 11) {  // the u64 FlagStark
    if (PART == 0) flag_u64_eval(cs, row, FlagU64Shape(p.num_io));
  } else if (KIND == 7 || KIND == 8) {   // ModularStark / Fq12Stark: everything but the permutation checks is the head segment
    const OpShape sh(KIND);
    if (PART == 0) op_eval<KIND>(cs, row, sh);
    else if (PART == 2) {
      if (seg == 2) permutation_checks(cs, row, zrow, sh, p.num_zs, F(p.gamma0), F(p.gamma1), 0, p.zsplit);
      else permutation_checks(cs, row, zrow, sh, p.num_zs, F(p.gamma0), F(p.gamma1), p.zsplit, p.num_zs);
    }
  } else {
    constexpr int E = KIND == 4 ? 12 : (KIND == 6 ? 13 : (KIND == 3 ? 2 : (KIND == 5 ? 0 : 1)));
    ExpShape sh(E, p.num_io);
    // u16 range check (G1 / G2 / Fq tables), SBN_QUOTIENT_LOOKUPS=1 (experiment switch): the lookup constraints beside the permutation
    // checks, which load the same columns (air.cuh lookups_beside_permutation).  Measured in round 4: the tail segment loses 0.8 GB of
    // reads, but the permutation segments -- the longest of the three concurrent kernels -- get the work: 1.18 -> 1.25 ms for the stage.
    const bool moved = (E == 0 || E == 1 || E == 2) && p.lookups_in_perm;
    if (PART < 2) exp_eval<E>(cs, row, sh, (const ExpPiConsts<F>*)pic_arg, (PART == 1 && moved) ? 3 : 1 + PART);
    else {
      const int z0 = seg == 2 ? 0 : p.zsplit, z1 = seg == 2 ? p.zsplit : p.num_zs;
      permutation_checks(cs, row, zrow, sh, p.num_zs, F(p.gamma0), F(p.gamma1), z0, z1);
      if (moved) lookups_beside_permutation(cs, row, sh, p.num_zs, (z0 + 1) / 2, (z1 + 1) / 2);
    }
  }
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) p.part[((size_t)seg * SBN_NCH + j) * p.m + i] = cs.result(j).v;
}
__global__ __launch_bounds__(256) void quotient_combine_kernel(QuotientParams p) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.m) return;
  F dinv(p.zh_inv[((i << p.row_shift) | p.row_rho) & 1]);
#pragma unroll
  for (int j = 0; j < SBN_NCH; j++) {
    F acc(p.part[((size_t)(QSEG - 1) * SBN_NCH + j) * p.m + i]);   // nothing follows the last segment
#pragma unroll
    for (u32 sgm = 0; sgm + 1 < QSEG; sgm++) acc += F(p.part[((size_t)sgm * SBN_NCH + j) * p.m + i]) * F(p.seg_shift[sgm][j]);
    p.qout[(size_t)j * p.m + i] = (acc * dinv).v;
  }
}

// Coset points and Lagrange selectors on the LDE domain (prover.rs: lagrange_first/last
// `.lde_onto_coset`, coset = cyclic_subgroup_coset_known_order).  L_0(x) = (x^N-1)/(N(x-1)),
// L_last(x) = (x^N-1)/(N(g x-1)); one Fermat inversion per point, run once per prover.
__global__ void domain_tables_kernel(u64* xs, u64* lag_first, u64* lag_last, size_t m, u32 lde_log, u32 degree_bits) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  F w = f_root_of_unity(lde_log);
  F x = F(GL_GEN) * f_pow(w, i);
  F g = f_root_of_unity(degree_bits);
  F nn = F((u64)1 << degree_bits);
  F zx = f_exp_pow2(x, degree_bits) - F(1);
  xs[i] = x.v;
  lag_first[i] = (zx * f_inv(nn * (x - F(1)))).v;
  lag_last[i] = (zx * f_inv(nn * (g * x - F(1)))).v;
}

// ---- permutation Z without reading Z back (round 4) ---------------------------------------------------------------------------------
// permutation_z_kernel moves 7 words per row (the rhs column twice, Z written, read and written again) and runs at HBM speed.  Cut into
// CHUNKS of 256 * PZ_E rows, a column needs no second sweep: pass A multiplies up num and den of every chunk (2 words per row), pass M
// turns the chunk products of a column into  P[c] = prod of num of the chunks before c  and  Q[c] = (prod of den of the chunks after c)
// / (prod of every den)  (one inversion per column), and pass B reads a chunk's two columns again and writes
//   Z[i] = P[c] * prod_{k < i, k in chunk} num_k  *  Q[c] * prod_{k >= i, k in chunk} den_k        (5 words per row, every pass parallel
// over columns x chunks).  Exact field arithmetic: the same Z.  tot / pq: [column][chunk][2].
__device__ __forceinline__ F wave_scan_mul_rev(F x, int lane) {  // inclusive SUFFIX product within a wave
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    u64 o = __shfl_down((unsigned long long)x.v, d, 64);
    if (lane + d < 64) x = x * F(o);
  }
  return x;
}
template <int PZ_E>
__global__ __launch_bounds__(256) void permz_chunk_products_kernel(const u64* __restrict__ trace, size_t n, const PairCols* __restrict__ pairs, u64 gamma0, u64 gamma1,
                                                                   u64* __restrict__ tot) {
  __shared__ u64 wn[4], wd[4];
  const int z = blockIdx.y;
  const size_t c = blockIdx.x, chunks = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t i0 = c * 256 * PZ_E + (size_t)tid * PZ_E;
  const u64* lhs = trace + (size_t)pairs[z].lhs * n + i0;
  const u64* rhs = trace + (size_t)pairs[z].rhs * n + i0;
  const F g0(gamma0), g1(gamma1);
  u64 lv[PZ_E], rv[PZ_E];
#pragma unroll
  for (int e = 0; e < PZ_E; e++) { lv[e] = lhs[e]; rv[e] = rhs[e]; }
  F pn(1), pd(1);
#pragma unroll
  for (int e = 0; e < PZ_E; e++) { const F l(lv[e]), r(rv[e]); pn = pn * ((l + g0) * (l + g1)); pd = pd * ((r + g0) * (r + g1)); }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { pn = pn * F(__shfl_xor((unsigned long long)pn.v, d, 64)); pd = pd * F(__shfl_xor((unsigned long long)pd.v, d, 64)); }
  if (lane == 0) { wn[wv] = pn.v; wd[wv] = pd.v; }
  __syncthreads();
  if (tid == 0) {
    tot[((size_t)z * chunks + c) * 2] = (F(wn[0]) * F(wn[1]) * F(wn[2]) * F(wn[3])).v;
    tot[((size_t)z * chunks + c) * 2 + 1] = (F(wd[0]) * F(wd[1]) * F(wd[2]) * F(wd[3])).v;
  }
}
// one wave per column: P[c], Q[c] from the chunk products, 64 chunks per step
__global__ __launch_bounds__(64) void permz_chunk_scan_kernel(const u64* __restrict__ tot, u32 chunks, u64* __restrict__ pq) {
  const int z = blockIdx.x, lane = threadIdx.x;
  const u64* t = tot + (size_t)z * chunks * 2;
  u64* o = pq + (size_t)z * chunks * 2;
  F carry(1), all(1);
  for (u32 b = 0; b < chunks; b += 64) {
    const u32 idx = b + lane;
    const bool on = idx < chunks;
    const F inc = wave_scan_mul(on ? F(t[2 * idx]) : F(1), lane);
    const u64 pv = __shfl_up((unsigned long long)inc.v, 1, 64);
    if (on) o[2 * idx] = (lane == 0 ? carry : carry * F(pv)).v;
    carry = carry * F(__shfl((unsigned long long)inc.v, 63, 64));
    F d = on ? F(t[2 * idx + 1]) : F(1);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) d = d * F(__shfl_xor((unsigned long long)d.v, k, 64));
    all = all * d;
  }
  F carry_s = f_inv(all);
  for (u32 b = ((chunks - 1) / 64) * 64;; b -= 64) {
    const u32 idx = b + lane;
    const bool on = idx < chunks;
    const F inc = wave_scan_mul_rev(on ? F(t[2 * idx + 1]) : F(1), lane);
    const u64 sv = __shfl_down((unsigned long long)inc.v, 1, 64);
    if (on) o[2 * idx + 1] = (lane == 63 ? carry_s : carry_s * F(sv)).v;
    carry_s = carry_s * F(__shfl((unsigned long long)inc.v, 0, 64));
    if (b == 0) break;
  }
}
template <int PZ_E>
__global__ __launch_bounds__(256) void permz_chunk_write_kernel(const u64* __restrict__ trace, size_t n, const PairCols* __restrict__ pairs, u64 gamma0, u64 gamma1,
                                                                const u64* __restrict__ pq, u64* __restrict__ zout) {
  __shared__ u64 wn[4], wd[4];
  const int z = blockIdx.y;
  const size_t c = blockIdx.x, chunks = gridDim.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const size_t i0 = c * 256 * PZ_E + (size_t)tid * PZ_E;
  const u64* lhs = trace + (size_t)pairs[z].lhs * n + i0;
  const u64* rhs = trace + (size_t)pairs[z].rhs * n + i0;
  u64* zc = zout + (size_t)z * n + i0;
  const F g0(gamma0), g1(gamma1);
  u64 lv[PZ_E], rv[PZ_E];
#pragma unroll
  for (int e = 0; e < PZ_E; e++) { lv[e] = lhs[e]; rv[e] = rhs[e]; }
  F pn[PZ_E], sd[PZ_E];   // inclusive prefix of num / inclusive suffix of den inside the lane
#pragma unroll
  for (int e = 0; e < PZ_E; e++) { const F l(lv[e]); const F num = (l + g0) * (l + g1); pn[e] = e ? pn[e - 1] * num : num; }
#pragma unroll
  for (int e = PZ_E - 1; e >= 0; e--) { const F r(rv[e]); const F den = (r + g0) * (r + g1); sd[e] = e < PZ_E - 1 ? sd[e + 1] * den : den; }
  const F incn = wave_scan_mul(pn[PZ_E - 1], lane), incd = wave_scan_mul_rev(sd[0], lane);
  if (lane == 63) wn[wv] = incn.v;
  if (lane == 0) wd[wv] = incd.v;
  __syncthreads();
  F pre(pq[((size_t)z * chunks + c) * 2]), suf(pq[((size_t)z * chunks + c) * 2 + 1]);
  for (int w = 0; w < wv; w++) pre = pre * F(wn[w]);
  for (int w = 3; w > wv; w--) suf = suf * F(wd[w]);
  const u64 pv = __shfl_up((unsigned long long)incn.v, 1, 64), sv = __shfl_down((unsigned long long)incd.v, 1, 64);
  const F exn = lane == 0 ? pre : pre * F(pv);     // prod of num of every row before this lane's first
  const F exd = lane == 63 ? suf : suf * F(sv);    // Q[c] * prod of den of every row after this lane's last
  const F both = exn * exd;
  zc[0] = (both * sd[0]).v;
#pragma unroll
  for (int e = 1; e < PZ_E; e++) zc[e] = (both * pn[e - 1] * sd[e]).v;
}

// =================================================================================================
// K6  openings (StarkOpeningSet::new; P4): out[p] = poly_p(z) over the extension, for two points.
// Block per polynomial; threads stride the coefficients against a precomputed table z^i (two planes).
// Algorithmic bytes: 8*N per polynomial per point (the power table stays in L2).
// =================================================================================================
__global__ void ext_pow_table_kernel(u64* pa, u64* pb, size_t n, u64 z0, u64 z1) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  E2 r = e2_pow(E2(F(z0), F(z1)), i);
  pa[i] = r.a.v; pb[i] = r.b.v;
}
__device__ __forceinline__ F block_sum(F v, u64* sh) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = v + F(__shfl_xor((unsigned long long)v.v, d, 64));
  __syncthreads();
  if (lane == 0) sh[wv] = v.v;
  __syncthreads();
  F t(0);
  for (unsigned w = 0; w < blockDim.x / 64; w++) t = t + F(sh[w]);
  return t;
}
// evaluates at two points at once: tables (pa0,pb0) for z and (pa1,pb1) for g*z; out: [npoly][4].  The kernel is bound
// by load latency, not bandwidth or ALU (one coefficient + four table words per ~110 instructions), so every lane has
// OPEN_UNROLL iterations' loads in flight before it multiplies (4 from 1,024 rows up).
template <u32 OPEN_UNROLL>   // n must be a multiple of 256 * OPEN_UNROLL
__global__ __launch_bounds__(256) void openings_kernel(const u64* __restrict__ coeffs, size_t n, const u64* __restrict__ pa0, const u64* __restrict__ pb0,
                                                       const u64* __restrict__ pa1, const u64* __restrict__ pb1, u64* __restrict__ out) {
  __shared__ u64 sh[4];
  const u64* c = coeffs + (size_t)blockIdx.x * n;
  Acc<F> s0, r0, s1, r1;   // unreduced 192-bit sums (air.cuh): 8 instructions per term instead of a multiply, a reduction and an add
  s0.clear(); r0.clear(); s1.clear(); r1.clear();
  for (size_t i = threadIdx.x; i < n; i += 256 * OPEN_UNROLL) {
    u64 cv[OPEN_UNROLL], t0[OPEN_UNROLL], t1[OPEN_UNROLL], t2[OPEN_UNROLL], t3[OPEN_UNROLL];
#pragma unroll
    for (u32 k = 0; k < OPEN_UNROLL; k++) {
      const size_t ik = i + 256 * k;
      cv[k] = c[ik]; t0[k] = pa0[ik]; t1[k] = pb0[ik]; t2[k] = pa1[ik]; t3[k] = pb1[ik];
    }
    __builtin_amdgcn_sched_barrier(0);   // all the loads first
#pragma unroll
    for (u32 k = 0; k < OPEN_UNROLL; k++) {
      const F v{cv[k]};
      s0.macv(v, F(t0[k])); r0.macv(v, F(t1[k]));
      s1.macv(v, F(t2[k])); r1.macv(v, F(t3[k]));
    }
  }
  const F a0 = block_sum(s0.value(), sh), b0 = block_sum(r0.value(), sh), a1 = block_sum(s1.value(), sh), b1 = block_sum(r1.value(), sh);
  if (threadIdx.x == 0) {
    u64* o = out + (size_t)blockIdx.x * 4;
    o[0] = a0.v; o[1] = b0.v; o[2] = a1.v; o[3] = b1.v;
  }
}

// One point only (tables pa, pb), written to out[4 * poly + off .. + 1]: the trace openings at zeta go first so that the
// host can start hashing them into the transcript while the device evaluates everything else.
template <u32 OPEN_UNROLL>
__global__ __launch_bounds__(256) void openings1_kernel(const u64* __restrict__ coeffs, size_t n, const u64* __restrict__ pa, const u64* __restrict__ pb,
                                                        u64* __restrict__ out, u32 off) {
  __shared__ u64 sh[4];
  const u64* c = coeffs + (size_t)blockIdx.x * n;
  Acc<F> s0, r0;
  s0.clear(); r0.clear();
  for (size_t i = threadIdx.x; i < n; i += 256 * OPEN_UNROLL) {
    u64 cv[OPEN_UNROLL], t0[OPEN_UNROLL], t1[OPEN_UNROLL];
#pragma unroll
    for (u32 k = 0; k < OPEN_UNROLL; k++) {
      const size_t ik = i + 256 * k;
      cv[k] = c[ik]; t0[k] = pa[ik]; t1[k] = pb[ik];
    }
    __builtin_amdgcn_sched_barrier(0);   // all the loads first
#pragma unroll
    for (u32 k = 0; k < OPEN_UNROLL; k++) {
      const F v{cv[k]};
      s0.macv(v, F(t0[k])); r0.macv(v, F(t1[k]));
    }
  }
  const F a0 = block_sum(s0.value(), sh), b0 = block_sum(r0.value(), sh);
  if (threadIdx.x == 0) {
    u64* o = out + (size_t)blockIdx.x * 4 + off;
    o[0] = a0.v; o[1] = b0.v;
  }
}

// =================================================================================================
// K7  FRI batch combine (fri/oracle.rs `prove_openings`: alpha.reduce_polys_base; P5)
// part[g][i] = sum_{j in group g} alpha^(j - j0_g) * f_j[i]  (Horner from the group's last poly);
// the host-side combine multiplies by alpha^(j0_g).  Thread per coefficient index, groups on
// blockIdx.y for parallelism.  Algorithmic bytes: 8*N per polynomial, read once.
// =================================================================================================
// (pa, pb)[k] = the two coordinates of alpha^k, k < group_size (ext_pow_table_kernel): a base-field coefficient times an
// extension power is two multiplies, where the Horner step acc * alpha + f is a full extension product; the table index is
// uniform, so the powers arrive through scalar loads.
__global__ __launch_bounds__(256) void fri_combine_partial_kernel(const u64* __restrict__ coeffs, size_t n, u32 npoly, u32 group_size,
                                                                  const u64* __restrict__ pa, const u64* __restrict__ pb, u64* __restrict__ part_a,
                                                                  u64* __restrict__ part_b) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const u32 g = blockIdx.y;
  const u32 j0 = g * group_size, cnt = j0 + group_size < npoly ? group_size : npoly - j0;
  const u64* c = coeffs + (size_t)j0 * n + i;
  Acc<F> a, b;   // unreduced sums, uniform weights (scalar operands)
  a.clear(); b.clear();
  u32 k = 0;
  for (; k + 8 <= cnt; k += 8) {
    u64 f[8];
#pragma unroll
    for (u32 u = 0; u < 8; u++) f[u] = c[(size_t)(k + u) * n];
#pragma unroll
    for (u32 u = 0; u < 8; u++) { const F v(f[u]); a.mac(v, F(pa[k + u])); b.mac(v, F(pb[k + u])); }
  }
  for (; k < cnt; k++) { const F v(c[(size_t)k * n]); a.mac(v, F(pa[k])); b.mac(v, F(pb[k])); }
  part_a[(size_t)g * n + i] = a.value().v; part_b[(size_t)g * n + i] = b.value().v;
}
// out[i] (+)= sum_g w_g * part[g][i]   (w_g given as ext pairs)
__global__ void fri_combine_reduce_kernel(const u64* part_a, const u64* part_b, size_t n, u32 ngroups, const u64* w, u64* out_a, u64* out_b, int accumulate) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  E2 acc = accumulate ? E2(F(out_a[i]), F(out_b[i])) : E2(F(0), F(0));
  for (u32 g = 0; g < ngroups; g++) acc += E2(F(part_a[(size_t)g * n + i]), F(part_b[(size_t)g * n + i])) * E2(F(w[2 * g]), F(w[2 * g + 1]));
  out_a[i] = acc.a.v; out_b[i] = acc.b.v;
}

// Synthetic division by (X - z) (PolynomialCoeffs::divide_by_linear):
// B_i = c_i + z*B_{i+1}; quotient q_{i-1} = B_i, q_{n-1} = 0; then out = out*mul + q (ext).
// The suffix recurrence is split into chunks of DBL_CHUNK coefficients:
//   pass 1 (one lane per chunk)  h_t = sum_k c_{tL+k} z^k, the chunk's own Horner value;
//   pass 2 (one workgroup)       carry_t = B_{(t+1)L} from B_{tL} = h_t + z^L B_{(t+1)L}, scanned from the top;
//   pass 3 (one lane per chunk)  replays the chunk from its carry and writes the quotient.
static constexpr u32 DBL_CHUNK = 16;
__global__ __launch_bounds__(256) void divide_by_linear_pass1(const u64* __restrict__ ca, const u64* __restrict__ cb, size_t nchunks, u64 z0, u64 z1,
                                                              u64* __restrict__ ha, u64* __restrict__ hb) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks) return;
  const E2 z{F(z0), F(z1)};
  const size_t s = t * DBL_CHUNK;
  E2 h{F(0), F(0)};
  for (u32 k = DBL_CHUNK; k-- > 0;) h = h * z + E2(F(ca[s + k]), F(cb[s + k]));
  ha[t] = h.a.v; hb[t] = h.b.v;
}
// in place: (ha, hb)[t] <- carry_t.  One workgroup; nchunks is a power of two, min(256, nchunks) lanes own
// nchunks/lanes consecutive chunks each: a lane folds its own chunks (loads issued eight at a time: the kernel is
// latency-bound), the lanes' values go through a log-step suffix scan of x -> g + zg x in LDS, then every lane replays
// its chunks from its carry.
template <u32 U> __device__ __forceinline__ void dbl_fold(const u64* ha, const u64* hb, size_t base, size_t per, const E2 zl, E2& g) {
  for (size_t j0 = per; j0 > 0; j0 -= U) {
    u64 va[U], vb[U];
#pragma unroll
    for (u32 k = 0; k < U; k++) { va[k] = ha[base + j0 - 1 - k]; vb[k] = hb[base + j0 - 1 - k]; }
#pragma unroll
    for (u32 k = 0; k < U; k++) g = E2(F(va[k]), F(vb[k])) + zl * g;
  }
}
template <u32 U> __device__ __forceinline__ void dbl_replay(u64* ha, u64* hb, size_t base, size_t per, const E2 zl, E2 carry) {
  for (size_t j0 = per; j0 > 0; j0 -= U) {
    u64 va[U], vb[U];
#pragma unroll
    for (u32 k = 0; k < U; k++) { va[k] = ha[base + j0 - 1 - k]; vb[k] = hb[base + j0 - 1 - k]; }
#pragma unroll
    for (u32 k = 0; k < U; k++) {
      ha[base + j0 - 1 - k] = carry.a.v; hb[base + j0 - 1 - k] = carry.b.v;
      carry = E2(F(va[k]), F(vb[k])) + zl * carry;
    }
  }
}
__global__ __launch_bounds__(256) void divide_by_linear_pass2(u64* __restrict__ ha, u64* __restrict__ hb, size_t nchunks, u64 z0, u64 z1) {
  __shared__ u64 sa[2][256], sb[2][256];
  const int tid = threadIdx.x;
  const int lanes = nchunks < 256 ? (int)nchunks : 256;
  const size_t per = nchunks / lanes, base = (size_t)tid * per;
  const bool live = tid < lanes;
  const E2 zl = e2_pow(E2{F(z0), F(z1)}, DBL_CHUNK);
  E2 g{F(0), F(0)};  // value of this lane's chunks with a zero carry-in
  if (live) { if (per % 8 == 0) dbl_fold<8>(ha, hb, base, per, zl, g); else dbl_fold<1>(ha, hb, base, per, zl, g); }
  // inclusive suffix scan S_t = g_t + zg S_{t+1} over the lanes (Hillis-Steele; the multiplier zg^(2^k) is uniform)
  E2 m = e2_pow(zl, per);
  int cur = 0;
  sa[0][tid] = g.a.v; sb[0][tid] = g.b.v;
  __syncthreads();
  for (int d = 1; d < lanes; d <<= 1) {
    E2 v{F(sa[cur][tid]), F(sb[cur][tid])};
    if (tid + d < lanes) v = v + m * E2(F(sa[cur][tid + d]), F(sb[cur][tid + d]));
    sa[cur ^ 1][tid] = v.a.v; sb[cur ^ 1][tid] = v.b.v;
    m = m * m;
    cur ^= 1;
    __syncthreads();
  }
  // carry into lane t = S_{t+1}
  E2 carry{F(0), F(0)};
  if (tid + 1 < lanes) carry = E2(F(sa[cur][tid + 1]), F(sb[cur][tid + 1]));
  if (live) { if (per % 8 == 0) dbl_replay<8>(ha, hb, base, per, zl, carry); else dbl_replay<1>(ha, hb, base, per, zl, carry); }
}
__global__ __launch_bounds__(256) void divide_by_linear_pass3(const u64* __restrict__ ca, const u64* __restrict__ cb, size_t nchunks, u64 z0, u64 z1,
                                                              const u64* __restrict__ ha, const u64* __restrict__ hb, u64 mul0, u64 mul1,
                                                              u64* __restrict__ oa, u64* __restrict__ ob, int accumulate, u32 times_x) {
  // times_x = 1: quotient coefficient q_i lands at index i + 1 (plonky2 0.1.x multiplies the final polynomial by X:
  // `final_poly.coeffs.insert(0, ZERO)` on the n-1 coefficient quotients); q_{n-1} = 0 is the padding and is dropped,
  // index 0 keeps the zero of the caller's memset.
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nchunks) return;
  const E2 z{F(z0), F(z1)}, mul{F(mul0), F(mul1)};
  const size_t s = t * DBL_CHUNK, n = nchunks * DBL_CHUNK;
  E2 b{F(ha[t]), F(hb[t])};
  for (u32 k = DBL_CHUNK; k-- > 0;) {
    const size_t i = s + k;
    const E2 q = b;  // q_i = B_{i+1}: the value of b BEFORE absorbing c_i
    b = b * z + E2(F(ca[i]), F(cb[i]));
    const size_t j = i + times_x;
    if (j >= n) continue;
    const E2 o = accumulate ? E2(F(oa[j]), F(ob[j])) * mul + q : q;
    oa[j] = o.a.v; ob[j] = o.b.v;
  }
}

// FRI fold in coefficient form (fri/prover.rs: chunks(arity).map(reduce_with_powers(chunk, beta))).
__global__ void fri_fold_kernel(const u64* ca, const u64* cb, size_t nout, u32 arity, u64 b0, u64 b1, u64* oa, u64* ob) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nout) return;
  E2 beta{F(b0), F(b1)}, acc(F(0), F(0));
  for (u32 t = arity; t-- > 0;) acc = acc * beta + E2(F(ca[i * arity + t]), F(cb[i * arity + t]));
  oa[i] = acc.a.v; ob[i] = acc.b.v;
}

// K9  proof-of-work grind (fri/prover.rs `fri_proof_of_work`): candidates base..base+count, the
// smallest hit is kept (deterministic witness).
struct PowParams { u64 state[12]; u32 wpos; u32 min_lz; u64 base; u64 count; u64* result; };
__global__ __launch_bounds__(256) void pow_kernel(PowParams p) {
  u64 gid = (u64)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= p.count) return;
  F s[12];
#pragma unroll
  for (int i = 0; i < 12; i++) s[i] = F(p.state[i]);
  s[p.wpos] = F(p.base + gid);
  poseidon_permute(s);
  u64 resp = s[P_RATE - 1].v;
  u32 lz = resp ? (u32)__clzll((long long)resp) : 64;
  if (lz >= p.min_lz) atomicMin((unsigned long long*)p.result, (unsigned long long)(p.base + gid));
}

// K10 query gathers, written straight in the proof's word layout.
// rows: out[q*qstride + off + c] = mat[c*m + bitrev(idx[q])]
// ---- oversized-trace split (BASELINE config[4]: one trace over R GPUs) ---------------------------------------------
// Rank s owns the LDE rows whose Merkle leaf index bitrev(i) has top log R bits = s, i.e. i = j * R + rho(s) with
// rho = bit reversal on log R bits: complete cap subtrees, so hashing needs no exchange.  split_pack_kernel scatters one
// column block [nc][m] of a rank's LDE by destination: plane L (row i goes to its owner at local row i >> log R) and, from
// 4 ranks up, plane N (row i is the NEXT row of point i - 2, whose owner gets it at the local row of that point; with 2
// ranks the next row of local row j is local row j + 1).  Rows for another rank go to the send slot [dest][ob][m/R]; the
// rank's OWN rows go straight into its receive matrix (self_l / self_n = the block's columns there), so no self block
// ever passes through the transport.
__global__ __launch_bounds__(256) void split_pack_kernel(const u64* __restrict__ lde, size_t m, u32 nc, u32 ob, u32 log_r, u32 me,
                                                         u64* __restrict__ send_l, u64* __restrict__ send_n, u64* __restrict__ self_l, u64* __restrict__ self_n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const u32 c = blockIdx.y;
  if (i >= m || c >= nc) return;
  const u64 v = lde[(size_t)c * m + i];
  const size_t ml = m >> log_r, rmask = ((size_t)1 << log_r) - 1;
  const u32 s = bitrev32((u32)(i & rmask), log_r);
  if (s == me) self_l[(size_t)c * ml + (i >> log_r)] = v;
  else send_l[((size_t)s * ob + c) * ml + (i >> log_r)] = v;
  if (send_n) {
    const size_t p = (i + m - 2) & (m - 1);   // the point whose next row this is
    const u32 s2 = bitrev32((u32)(p & rmask), log_r);
    if (s2 == me) self_n[(size_t)c * ml + (p >> log_r)] = v;
    else send_n[((size_t)s2 * ob + c) * ml + (p >> log_r)] = v;
  }
}
// gathered [rank][nplanes][m/R] -> natural order [nplanes][m]
__global__ void split_unpack_rows_kernel(const u64* __restrict__ in, size_t m, u32 nplanes, u32 log_r, u64* __restrict__ out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= m) return;
  const size_t ml = m >> log_r;
  const u32 s = bitrev32((u32)(i & (((size_t)1 << log_r) - 1)), log_r);
  for (u32 p = 0; p < nplanes; p++) out[(size_t)p * m + i] = in[((size_t)s * nplanes + p) * ml + (i >> log_r)];
}
// out[k] = sum over ranks of in[rank][k] (mod p): the FRI batch-combine partial sums (RCCL has no mod-p reduction)
__global__ void split_modadd_kernel(const u64* __restrict__ in, size_t len, u32 nranks, u64* __restrict__ out) {
  const size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= len) return;
  F acc(in[k]);
  for (u32 r = 1; r < nranks; r++) acc += F(in[(size_t)r * len + k]);
  out[k] = acc.v;
}

__global__ void gather_rows_kernel(const u64* mat, size_t m, u32 lde_log, u32 ncols, const u32* idx, u64* out, size_t qstride, size_t off) {
  u32 c = blockIdx.x * blockDim.x + threadIdx.x;
  u32 q = blockIdx.y;
  if (c >= ncols) return;
  size_t row = bitrev32(idx[q], lde_log);
  out[(size_t)q * qstride + off + c] = mat[(size_t)c * m + row];
}
// siblings: tree levels concatenated (level l has nleaf>>l digests); leaf index = idx[q] >> shift.
__global__ void gather_siblings_kernel(const u64* tree, size_t nleaf, u32 nlevels, const u32* idx, u32 shift, u64* out, size_t qstride, size_t off) {
  u32 q = blockIdx.x;
  u32 e = threadIdx.x;
  if (e >= nlevels * 4) return;
  u32 l = e >> 2, k = e & 3;
  size_t x = idx[q] >> shift;
  size_t level_off = 2 * nleaf - ((2 * nleaf) >> l);  // sum_{k<l} nleaf>>k
  size_t sib = (x >> l) ^ 1;
  out[(size_t)q * qstride + off + e] = tree[(level_off + sib) * 4 + k];
}
// FRI layer leaf: 2*arity words (c0,c1 interleaved) of leaf (idx>>shift).
__global__ void gather_fri_leaf_kernel(const u64* va, const u64* vb, u32 log_m, u32 arity_bits, const u32* idx, u32 shift, u64* out, size_t qstride, size_t off) {
  u32 q = blockIdx.x;
  u32 e = threadIdx.x;
  u32 arity = 1u << arity_bits;
  if (e >= 2 * arity) return;
  size_t leaf = idx[q] >> shift;
  u32 nat = bitrev32((u32)(leaf * arity + (e >> 1)), log_m);
  out[(size_t)q * qstride + off + e] = (e & 1) ? vb[nat] : va[nat];
}
