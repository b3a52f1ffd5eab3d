/*
 * sbn.h -- C ABI of the MI355X-native Starky/BN254 prover path (libsbn254.so).
 *
 * Drop-in boundary for the reference's prove()/verify_stark_proof() calls on the G1 / G2 / Fq12 tables:
 *   reference call sites   src/curves/g1/exp.rs:811-826      (G1ExpStark: trace, pi, prove, verify)
 *                          src/curves/g1/muladd.rs:666-678   (G1Stark)
 *                          src/curves/g1/circuit.rs:187-201  (G1ExpStarkyProofGenerator::run_once)
 *   reference signatures   starky 0.1.1 `prove::<F,C,S,D>(stark, &config, trace_poly_values,
 *                          public_inputs, &mut timing)` and `verify_stark_proof(stark, proof, &config)`
 *                          (un-vendored dependency, Cargo.toml:21).
 *
 * Because the AIR's `eval_packed_generic` is generic Rust that cannot cross a C ABI, a table is
 * named by (kind, num_io) and its constraint code lives natively behind this boundary.
 *
 * Conventions: plain pointers and sizes, no exceptions; 0 = success, negative = sbn_status error.
 * All field elements are canonical Goldilocks u64 (< 2^64 - 2^32 + 1), little-endian.
 * Trace wire format = the reference's Vec<PolynomialValues<F>>: COLUMN-MAJOR [num_columns][N]
 * (src/curves/g1/exp.rs:314-317).  Public inputs: flat [num_public_inputs] (exp.rs:320-327).
 *
 * Proof byte layout ("canonical proof words", LE u64 each; the reference never serialises a proof,
 * so this layout is defined here and shared with the test oracle):
 *   header[12] = { magic "SNBPROV1", degree_bits, n_trace_cols, n_perm_zs, n_quotient_polys,
 *                  n_public_inputs, cap_height, rate_bits, n_fri_layers, arity_bits,
 *                  final_poly_len, n_queries }
 *   trace_cap[2^cap_height][4], permutation_zs_cap[..][4] (iff n_perm_zs>0), quotient_polys_cap[..][4]
 *   openings: local_values[n_trace_cols][2], next_values[..][2], permutation_zs[n_perm_zs][2],
 *             permutation_zs_next[..][2], quotient_polys[n_quotient_polys][2]     (ext elem = c0,c1)
 *   fri commit_phase_merkle_caps[n_fri_layers][2^cap_height][4]
 *   per query (n_queries): per initial oracle (trace, [perm_zs], quotient): leaf row values, then
 *             siblings[lde_bits-cap_height][4]; per FRI layer: evals[2^arity_bits][2], siblings[..][4]
 *   final_poly[final_poly_len][2], pow_witness, public_inputs[n_public_inputs]
 */
#ifndef SBN_H
#define SBN_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sbn_status {
  SBN_OK = 0,
  SBN_ERR_BAD_ARG = -1,       /* null pointer, unknown air kind, size mismatch */
  SBN_ERR_NON_CANONICAL = -2, /* a field element >= p was supplied */
  SBN_ERR_NO_DEVICE = -3,     /* no HIP device / HIP runtime failure at init */
  SBN_ERR_HIP = -4,           /* a HIP call failed (message via sbn_last_error) */
  SBN_ERR_MALFORMED_PROOF = -5,
  SBN_ERR_VERIFY_FAILED = -6, /* proof rejected (reason via sbn_last_error) */
  SBN_ERR_UNSUPPORTED = -7,   /* e.g. G1_EXP with fewer than 2^16 rows (range_check.rs:26) */
  SBN_ERR_WITNESS = -8        /* trace generation hit a degenerate case (x1==x2 in an affine add) */
} sbn_status;

/* Table kinds.  G1_OP = reference `G1Stark` (src/curves/g1/muladd.rs:462-624);
 * G1_EXP = reference `G1ExpStark` (src/curves/g1/exp.rs:232-742). */
typedef enum sbn_air_kind { SBN_AIR_G1_OP = 1, SBN_AIR_G1_EXP = 2, SBN_AIR_G2_EXP = 3, SBN_AIR_FQ12_EXP = 4, SBN_AIR_FQ_EXP = 5, SBN_AIR_FQ12_EXP_U64 = 6,
                              SBN_AIR_MODULAR = 7, SBN_AIR_FQ12_MUL = 8, SBN_AIR_LOOKUP = 9, SBN_AIR_FLAGS = 10, SBN_AIR_FLAGS_U64 = 11 } sbn_air_kind;
/* G2_EXP = reference `G2ExpStark` (src/curves/g2/exp.rs:248-807): the same machine over Fq2 coordinates.
 * FQ12_EXP = reference `Fq12ExpStark` (src/fields/fq12/exp.rs:223-605): offset * x^e in Fq12 (flat basis of
 * plonky2-bn254 `MyFq12`: coefficient of w^k is c[k] + c[k+6]*i, w^6 = 9 + i), 512 rows per instance, num_io a
 * power of two between 1 and 512.
 * FQ_EXP = reference `FqExpStark` (src/fields/fq/exp.rs:193-582): offset * x^e in the base field Fq, the same
 * square-and-multiply machine over one coefficient (960 columns at num_io = 128), u32 public inputs, u16 range check.
 * FQ12_EXP_U64 = reference `Fq12ExpU64Stark` (src/fields/fq12_u64/exp_u64.rs:243-571): Fq12 exponentiation by a u64
 * exponent, 128 rows per instance, 6-column flags (flags_u64.rs), the exponent is ONE public input (< p).
 * MODULAR = reference `ModularStark` (src/modular/modular.rs:361-537) and FQ12_MUL = reference `Fq12Stark`
 * (src/fields/fq12/mul.rs:355-517): its single-operation test tables for the modular gadget (a * b mod p per row, 812
 * columns) and the Fq12 product (9722 columns); no public inputs, num_io ignored, any power-of-two height >= 256.
 * LOOKUP = reference `MyStark` (src/utils/lookup.rs:136-213), its unit-test table of the lookup argument: 4 columns (inputs,
 * table, permuted inputs, permuted table), two permutation pairs, Merkle leaves are the rows themselves (hash_or_noop); any
 * power-of-two height >= 512, num_io ignored (the reference's own 8-row instance is below the device prover's minimum).
 * FLAGS = reference `FlagStark` (src/utils/flags.rs:379-547), the unit-test table of the exponent-bit flags: num_io inputs
 * of 8 u32 limbs, 512 rows each (the reference uses 16), 17 + 4 * num_io columns, NO permutation pairs -- the one table
 * whose proof carries no permutation-Z commitment (header n_perm_zs = 0).
 * FLAGS_U64 = the `FlagStark` of src/fields/fq12_u64/flags_u64.rs:289-420, the same for the u64-exponent flags: num_io inputs
 * (u64), 128 rows each, 7 + 4 * num_io columns, no rotation pulse, no permutation pairs; num_io >= 4 (512 rows). */

typedef struct sbn_air_desc {
  int32_t kind;    /* sbn_air_kind */
  uint32_t num_io; /* Exp tables: number of instances (rows = 512*num_io; FQ12_EXP_U64: 128*num_io); G1_OP: ignored */
} sbn_air_desc;

/* Mirrors starky `StarkConfig` + plonky2 `FriConfig` (reference: stark.config() ->
 * StarkConfig::standard_fast_config, src/curves/g1/exp.rs:250-253). */
typedef struct sbn_config {
  uint32_t security_bits;      /* 100 */
  uint32_t num_challenges;     /* 2   */
  uint32_t rate_bits;          /* 1   */
  uint32_t cap_height;         /* 4   */
  uint32_t proof_of_work_bits; /* 16  */
  uint32_t fri_arity_bits;     /* 4   (FriReductionStrategy::ConstantArityBits(4, 5)) */
  uint32_t fri_final_poly_bits;/* 5   */
  uint32_t num_query_rounds;   /* 84  */
  /* Not a StarkConfig field: which plonky2 FRI this library speaks (sbn_fri_variant).  SBN_FRI_TIMES_X = the 0.1.x line
   * the reference pins (plonky2 0.1.3 @ 541e127, Cargo.lock:529-531): fri/oracle.rs `prove_openings` multiplies the final
   * polynomial by X (`final_poly.coeffs.insert(0, ZERO)`, mir-protocol/plonky2 PR #436) and fri/verifier.rs
   * `fri_combine_initial` returns `sum * subgroup_x`.  SBN_FRI_PLAIN = the form without that step (the quotients are
   * zero-padded at the end), kept selectable.  0 = SBN_FRI_DEFAULT (= SBN_FRI_TIMES_X), so a zero-initialised field
   * selects the default and not the other protocol.  The dependency is un-vendored, so the default is recalled
   * ([DEP-RECALL]); both forms are tested. */
  uint32_t fri_variant;
} sbn_config;
typedef enum sbn_fri_variant { SBN_FRI_DEFAULT = 0, SBN_FRI_TIMES_X = 1, SBN_FRI_PLAIN = 2 } sbn_fri_variant;

typedef struct sbn_prover sbn_prover; /* device context: buffers sized for one (air, degree_bits) */
typedef struct sbn_proof sbn_proof;   /* host-side proof object (canonical words) */

/* Library / device ------------------------------------------------------------------------------ */
const char* sbn_version(void);
/* Bumped whenever a struct or a function signature of this header changes (3: sbn_config.fri_variant replaces
 * fri_final_poly_times_x, sbn_comm carries struct_size and stream-ordered callbacks; 4: sbn_set_thread_device,
 * sbn_prover_describe, sbn_set_device also selects HIP's current device).  Callers compare with SBN_ABI_VERSION. */
#define SBN_ABI_VERSION 4
int sbn_abi_version(void);
const char* sbn_last_error(void);                    /* thread-local message of the last failure */
int sbn_device_count(void);
/* Device of the provers and transports the calling thread creates afterwards, and the default of threads that never chose one.
 * Also makes it HIP's current device of the calling thread (hipSetDevice), so sbn_set_device(r) followed by
 * sbn_rccl_comm_create puts the staging buffers, the communicator and the prover of rank r on GPU r. */
int sbn_set_device(int device);
/* The same for the calling thread ONLY (the process default stays): for rank threads of one process (sbn_local_comm_create). */
int sbn_set_thread_device(int device);
void sbn_standard_fast_config(sbn_config* out);      /* StarkConfig::standard_fast_config */

/* Table shape (ExpStarkConstants, src/curves/g1/exp.rs:6-34) ----------------------------------- */
size_t sbn_air_num_columns(const sbn_air_desc* air);
size_t sbn_air_num_public_inputs(const sbn_air_desc* air);
size_t sbn_air_num_permutation_zs(const sbn_air_desc* air, const sbn_config* cfg);
size_t sbn_air_num_constraints(const sbn_air_desc* air); /* AIR constraints per point (no perm checks) */

/* Witness generation (host): replaces G1ExpStark::generate_trace / generate_public_inputs
 * (src/curves/g1/exp.rs:290-327) and G1Stark::generate_trace (muladd.rs:481-546).
 * ios: num_io x 40 u32 = x.x[8] x.y[8] offset.x[8] offset.y[8] exp_val[8]  (u32 limbs, LE).
 * trace_out: column-major [num_columns][512*num_io]; pi_out: [56*num_io]. */
int sbn_generate_trace_g1_exp(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out);
/* G2ExpStark::generate_trace / generate_public_inputs (src/curves/g2/exp.rs:307-342).
 * ios: num_io x 72 u32 = x.x.c0 x.x.c1 x.y.c0 x.y.c1 offset.x.c0 offset.x.c1 offset.y.c0 offset.y.c1 exp_val
 * (8 u32 limbs each); trace_out: [num_columns][512*num_io]; pi_out: [104*num_io]. */
int sbn_generate_trace_g2_exp(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out);
/* Fq12ExpStark::generate_trace / generate_public_inputs (src/fields/fq12/exp.rs:283-319).
 * ios: num_io x 200 u32 = x[12] offset[12] (flat-basis coefficients, 8 u32 limbs each) exp_val[8];
 * trace_out: [num_columns][512*num_io]; pi_out: [584*num_io]. */
int sbn_generate_trace_fq12_exp(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out);
/* Fq12ExpU64Stark::generate_trace / generate_public_inputs (src/fields/fq12_u64/exp_u64.rs:283-313).
 * ios: num_io x 194 u32 = x[12] offset[12] (8 u32 limbs each) exp_val (low, high u32; value < 2^64 - 2^32 + 1);
 * trace_out: [num_columns][128*num_io]; pi_out: [577*num_io]. */
int sbn_generate_trace_fq12_exp_u64(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out);
/* FqExpStark::generate_trace / generate_public_inputs (src/fields/fq/exp.rs:248-284).
 * ios: num_io x 24 u32 = x[8] offset[8] exp_val[8]; trace_out: [num_columns][512*num_io]; pi_out: [32*num_io]. */
int sbn_generate_trace_fq_exp(const uint32_t* ios, size_t num_io, uint64_t* trace_out, uint64_t* pi_out);
/* pts: rows x 32 u32 = a.x[8] a.y[8] b.x[8] b.y[8]; trace_out: [num_columns][rows]. */
int sbn_generate_trace_g1_op(const uint32_t* pts, size_t rows, uint64_t* trace_out);
/* ModularStark::generate_trace (modular.rs:383-437) with the operands given: ops: rows x 16 u32 = a[8] b[8] (both < p). */
int sbn_generate_trace_modular(const uint32_t* ops, size_t rows, uint64_t* trace_out);
/* Fq12Stark::generate_trace (fq12/mul.rs:375-419): ops: rows x 192 u32 = x[12] y[12] (flat-basis coefficients < p, 8 u32 limbs each). */
int sbn_generate_trace_fq12_mul(const uint32_t* ops, size_t rows, uint64_t* trace_out);

/* MyStark::generate_trace (lookup.rs:151-166) on caller-given columns: inputs[rows], table[rows] (canonical field elements;
 * every input value must occur in the table) -> [4][rows] = inputs, table, permuted inputs, permuted table (permuted_cols,
 * lookup.rs:60-111). */
int sbn_generate_trace_lookup(const uint64_t* inputs, const uint64_t* table, size_t rows, uint64_t* trace_out);
/* FlagStark::generate_trace (flags.rs:392-440): limbs: num_io x 8 u32; trace_out: [17 + 4 * num_io][512 * num_io]. */
int sbn_generate_trace_flags(const uint32_t* limbs, size_t num_io, uint64_t* trace_out);
/* flags_u64.rs FlagStark::generate_trace (:316-337): exps: num_io x u64; trace_out: [7 + 4 * num_io][128 * num_io]. */
int sbn_generate_trace_flags_u64(const uint64_t* exps, size_t num_io, uint64_t* trace_out);

/* Prover ---------------------------------------------------------------------------------------- */
int sbn_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, sbn_prover** out);
void sbn_prover_destroy(sbn_prover* p);
/* Host -> device copy of the trace (PCIe-inclusive path). */
int sbn_prover_load_trace(sbn_prover* p, const uint64_t* trace_col_major, const uint64_t* public_inputs, size_t n_pi);
/* Trace already resident in HBM (device pointer, same layout); copied device-to-device. */
int sbn_prover_load_trace_device(sbn_prover* p, const uint64_t* d_trace_col_major, const uint64_t* public_inputs, size_t n_pi);
/* prove() on the loaded trace; may be called repeatedly (the loaded trace is preserved). */
int sbn_prover_prove(sbn_prover* p, sbn_proof** out);
/* Per-stage device times (ms, HIP events on the prover's stream) of the last prove():
 * names via sbn_prover_stage_name(i); returns the number of stages written. */
int sbn_prover_stage_times(const sbn_prover* p, float* ms_out, int cap);
const char* sbn_prover_stage_name(int i);
/* Parity hook: the two Jacobian curve chains of every G1ExpStark (E = 1) / G2ExpStark (E = 2) instance as the device witness
 * generator consumes them ([num_io][257][3][E][4] u64 each, Montgomery form; csrc/bn254w.cuh exp_chains).  form 0 = the library's
 * choice (eight instances per AVX-512 IFMA register when the CPU has it), 1 = one instance at a time, 2 = IFMA or
 * SBN_ERR_UNSUPPORTED.  The forms write the same words. */
int sbn_host_curve_chains(int E, const uint32_t* ios, size_t num_io, uint64_t* ja_out, uint64_t* jb_out, int form);
/* The SBN_* environment switches this prover was created under, resolved, as one line of key=value pairs (csrc/settings.hpp:
 * the environment is read once, at creation; experiment switches need SBN_EXPERIMENTAL=1 and are listed under ignored=[...]
 * otherwise; a value that is not understood makes sbn_prover_create fail with SBN_ERR_BAD_ARG). */
int sbn_prover_describe(const sbn_prover* p, char* out, size_t cap);
/* The same check without a prover or a device: SBN_OK and the resolved switches of the calling process, or SBN_ERR_BAD_ARG. */
int sbn_settings_check(char* out, size_t cap);
/* Raw device pointer of the loaded trace buffer (for callers that fill it on-device). */
uint64_t* sbn_prover_trace_device_ptr(sbn_prover* p);
/* Witness generation ON THE DEVICE, straight into the prover's trace buffer: G1ExpStark / G2ExpStark / Fq12ExpStark / FqExpStark
 * ::generate_trace + generate_public_inputs (src/curves/g1/exp.rs:255-327, src/curves/g2/exp.rs:271-342,
 * src/fields/fq12/exp.rs:283-319) without the trace ever crossing PCIe.  Same `ios` layout and the same resulting trace /
 * public inputs, bit for bit, as sbn_generate_trace_{g1,g2,fq12}_exp; afterwards the prover is loaded and
 * sbn_prover_prove can run.  pi_out (optional): [num_public_inputs].  G1_EXP / G2_EXP / FQ_EXP: 2^16 .. 2^18 rows (the
 * reference pads to any power of two >= 128 instances, src/curves/g1/circuit.rs:273-277; SBN_ERR_UNSUPPORTED beyond: use
 * the host generators + sbn_prover_load_trace); FQ12_EXP, FQ12_EXP_U64: any size. */
int sbn_prover_generate_trace(sbn_prover* p, const uint32_t* ios, size_t num_io, uint64_t* pi_out);
/* Device -> host copy of the loaded trace, column-major [num_columns][N] (tests, debugging). */
int sbn_prover_read_trace(sbn_prover* p, uint64_t* trace_out);

/* One-shot convenience with the reference's argument list:
 * prove(stark, &config, trace_poly_values, public_inputs) (src/curves/g1/exp.rs:818-825). */
int sbn_prove(const sbn_air_desc* air, const sbn_config* cfg, const uint64_t* trace_col_major, uint32_t degree_bits,
              const uint64_t* public_inputs, size_t n_pi, sbn_proof** out);

/* Batch mode (BASELINE config "batch of independent proofs"): `inflight` prover contexts on the current GPU with one
 * host thread each.  Every unit is one instance list of the table (`num_io` instances, `ios_words_per_unit` u32 words,
 * layouts as for sbn_generate_trace_*): its witness is generated on the device and proved; proofs_out[count] receives
 * the proofs in unit order (all freed and an error returned if any unit fails).  The reference counterpart is the loop
 * of `G1ExpStarkyProofGenerator::run_once` calls over chunks of 128 instances (src/curves/g1/circuit.rs:161-202). */
typedef struct sbn_batch_prover sbn_batch_prover;
int sbn_batch_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, uint32_t inflight, sbn_batch_prover** out);
int sbn_batch_prover_prove_ios(sbn_batch_prover* b, const uint32_t* ios, size_t ios_words_per_unit, size_t num_io, size_t count, sbn_proof** proofs_out);
void sbn_batch_prover_destroy(sbn_batch_prover* b);

/* One oversized trace split over the GPUs of a node (BASELINE config "Single Fq12 exponentiation proof, trace height
 * 2^18, 8xMI355X with RCCL FRI fold"; reference workload src/fields/fq12/exp.rs:638-696).  One rank per GPU (one process
 * each, or the threads of one process with sbn_local_comm_create); every rank calls the same functions with the same
 * arguments and receives the same proof.  Work and the derived matrices (coefficients, LDE, Z, Merkle trees) are sharded:
 * COLUMNS for iNTT / LDE / Z / openings / the FRI batch combination, dealt round-robin in blocks of 64 (rank r owns the
 * column blocks r, r + world, ...); LDE ROWS for leaf hashing, Merkle subtrees, constraint evaluation and query answers
 * (rank s owns the rows whose Merkle leaf index has top log2(world) bits = s: complete cap subtrees).  The trace VALUES
 * are resident on every rank (generated there by sbn_split_prover_generate_trace, or loaded).  Exchange steps: one
 * all-to-all per column block and plane (columns -> rows), pipelined -- while block k travels, block k + 1 is transformed
 * and the blocks k - 1 of all ranks are absorbed by the leaf sponge -- and small all-gathers (caps, quotient values,
 * openings, FRI partial sums, query rows).  The collectives come with the sbn_comm: the transports of this library
 * (sbn_rccl_comm_create: RCCL send / recv over xGMI; sbn_local_comm_create: the ranks are threads of one process), or the
 * caller's own (torch.distributed in starky_bn254_amd/split.py; a host-staged backend for tests).
 * world must be 1, 2, 4, 8 or 16 (<= 2^cap_height). */
typedef struct sbn_comm {
  uint32_t struct_size;      /* sizeof(sbn_comm) of the caller's header (ABI check) */
  uint32_t reserved;
  void* ctx;                 /* passed back to the callbacks */
  uint32_t rank, world;
  void* send_buf;            /* device memory, >= send_bytes of sbn_split_exchange_bytes */
  void* recv_buf;            /* device memory, >= recv_bytes; holds this rank's row-sharded LDE matrices during a proof */
  uint64_t send_bytes, recv_bytes;
  /* Block d = send_buf[send_off[d] .. +send_len[d]) goes to rank d; the block from rank s lands at
   * recv_buf[recv_off[s] .. +recv_len[s]).  Arrays of `world` entries, bytes; zero-length blocks are skipped (rank d's
   * recv_len[s] equals rank s's send_len[d]); blocks sent to different ranks may be the same region (an all-gather); a
   * rank may send a block to itself.  STREAM-ORDERED: `stream` is a hipStream_t of the device that owns the buffers; the
   * send blocks are read after everything enqueued on it before the call, and work enqueued on it after the call sees
   * the received blocks complete and may overwrite the send blocks.  The call itself may return early (RCCL) or block
   * (a host-staged transport synchronises the stream itself).  The arrays are only valid during the call.  0 = ok. */
  int (*all_to_all)(void* ctx, void* stream, const uint64_t* send_off, const uint64_t* send_len, const uint64_t* recv_off, const uint64_t* recv_len);
  /* Host memory, blocking: every rank contributes `bytes` at send; recv = [world][bytes] in rank order. */
  int (*all_gather_host)(void* ctx, const void* send, void* recv, uint64_t bytes);
} sbn_comm;
typedef struct sbn_split_prover sbn_split_prover;
/* Staging sizes a rank needs for (air, degree_bits) in a world of `world` ranks. */
int sbn_split_exchange_bytes(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, uint32_t world, uint64_t* send_bytes, uint64_t* recv_bytes);
int sbn_split_prover_create(const sbn_air_desc* air, const sbn_config* cfg, uint32_t degree_bits, const sbn_comm* comm, sbn_split_prover** out);
void sbn_split_prover_destroy(sbn_split_prover* p);
/* As sbn_prover_generate_trace / sbn_prover_load_trace: the full trace values on this rank. */
int sbn_split_prover_generate_trace(sbn_split_prover* p, const uint32_t* ios, size_t num_io, uint64_t* pi_out);
int sbn_split_prover_load_trace(sbn_split_prover* p, const uint64_t* trace_col_major, const uint64_t* public_inputs, size_t n_pi);
/* The proof of the whole trace, on every rank, word for word the proof sbn_prover_prove gives on one GPU. */
int sbn_split_prover_prove(sbn_split_prover* p, sbn_proof** out);
int sbn_split_prover_stage_times(const sbn_split_prover* p, float* ms_out, int cap);

/* Transports that fill an sbn_comm: starky_bn254_amd/csrc/transport.hip ---------------------- */
/* RCCL over xGMI, one process per GPU, no Python: librccl is loaded at run time (SBN_RCCL_LIB overrides the name), so the
 * library itself has no link-time dependency on it.  Rank 0 calls sbn_rccl_unique_id and hands the 128 bytes to the other
 * ranks by whatever means the caller has (a file, a socket, MPI); then every rank calls sbn_rccl_comm_create with the
 * staging sizes of sbn_split_exchange_bytes (the transport allocates send_buf / recv_buf on the current device).
 * all_to_all = ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on the given stream; all_gather_host = ncclAllGather on
 * a small device buffer + copies. */
int sbn_rccl_unique_id(uint8_t id_out[128]);
int sbn_rccl_comm_create(const uint8_t id[128], uint32_t rank, uint32_t world, uint64_t send_bytes, uint64_t recv_bytes, sbn_comm* out);
void sbn_rccl_comm_destroy(sbn_comm* comm);
/* The ranks are threads of ONE process: rank r on devices[r] (null: all ranks on the current device -- how the parity tests
 * run 8 and 16 ranks on a one-GPU box).  Blocks are pulled with stream-ordered device-to-device copies (peer copies over
 * xGMI between different devices); the ranks meet at host barriers, so every rank must be inside the same library call
 * on its own thread.  A failing rank calls sbn_local_comm_abort, which releases the others with an error. */
typedef struct sbn_local_group sbn_local_group;
int sbn_local_comm_create(uint32_t world, const int* devices, uint64_t send_bytes, uint64_t recv_bytes, sbn_comm* comms_out, sbn_local_group** out);
void sbn_local_comm_abort(sbn_local_group* g);
void sbn_local_comm_destroy(sbn_local_group* g);
/* A pattern exchange through `comm` (uneven blocks, self blocks, an all-gather, the host all-gather), checked on the
 * device: every rank calls it (with the device of its staging buffers current); 0 = the transport moved every byte where it
 * belongs. */
int sbn_comm_selftest(const sbn_comm* comm);

/* Proof object ---------------------------------------------------------------------------------- */
size_t sbn_proof_num_words(const sbn_proof* proof);
const uint64_t* sbn_proof_words(const sbn_proof* proof);
/* Writes the canonical LE byte stream; returns bytes needed (call with cap=0 to size). */
size_t sbn_proof_serialize(const sbn_proof* proof, uint8_t* buf, size_t cap);
uint32_t sbn_proof_degree_bits(const sbn_proof* proof); /* StarkProof::recover_degree_bits, exp.rs:829 */
void sbn_proof_free(sbn_proof* proof);

/* Verifier: verify_stark_proof(stark, proof, &config) (src/curves/g1/exp.rs:826). */
int sbn_verify(const sbn_air_desc* air, const sbn_config* cfg, const uint8_t* proof_bytes, size_t len);

/* Building blocks exposed for parity tests and benchmarks (device in/out unless noted) --------- */
/* PolynomialBatch::from_values on a host column-major matrix: Merkle cap (2^cap_height x 4 words),
 * optionally coefficients [ncols][n] and LDE [ncols][n<<rate_bits] (natural order) back to host. */
int sbn_commit_values(const uint64_t* cols, size_t ncols, size_t n, uint32_t rate_bits, uint32_t cap_height,
                      uint64_t* cap_out, uint64_t* coeffs_out, uint64_t* lde_out);
/* Poseidon permutation of `count` independent width-12 states on the device (host in/out). */
int sbn_poseidon_permute_batch(uint64_t* states, size_t count);
/* out[i] = a[i] * b[i] in the Goldilocks field with the DEVICE multiply of every kernel (csrc/gl.cuh: 13-instruction weak product +
 * canonicalisation; mode 0), or the weak product of the transform passes canonicalised afterwards (mode 1).  a, b: any 64-bit
 * values (non-canonical representatives included: the kernels' intermediate values are); out canonical.  Host in/out. */
int sbn_field_mul_batch(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t count, int mode);
/* The host permutation behind the Fiat-Shamir transcript of prove()/verify() (plonky2 Challenger's
 * PoseidonPermutation): sparse partial rounds, or the plain definition when use_definition != 0.  Host only. */
int sbn_poseidon_permute_host(uint64_t* states, size_t count, int use_definition);
/* The table's AIR constraints (eval_packed_generic of the table, e.g. src/curves/g1/exp.rs:331-495; no permutation checks)
 * folded into the num_challenges = 2 accumulators acc_j = sum_t c_t alpha_j^(n-1-t) on ONE row pair over the base field:
 * the regrouped evaluator the quotient kernel runs, on the host.  local_row / next_row: [num_columns]; z_last = x - g^-1,
 * l_first / l_last = the Lagrange selectors at the point (starky ConstraintConsumer).  Host only. */
int sbn_eval_constraints_host(const sbn_air_desc* air, const uint64_t* local_row, const uint64_t* next_row, const uint64_t* public_inputs,
                              size_t n_pi, const uint64_t* alphas, uint64_t z_last, uint64_t l_first, uint64_t l_last, uint64_t* acc_out);

#ifdef __cplusplus
}
#endif
#endif /* SBN_H */
