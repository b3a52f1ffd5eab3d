//! Raw declarations of include/sbn.h (one line per entry point the shim uses).
#![allow(non_camel_case_types)]
use std::os::raw::c_char;

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct sbn_air_desc {
    pub kind: i32,
    pub num_io: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct sbn_config {
    pub security_bits: u32,
    pub num_challenges: u32,
    pub rate_bits: u32,
    pub cap_height: u32,
    pub proof_of_work_bits: u32,
    pub fri_arity_bits: u32,
    pub fri_final_poly_bits: u32,
    pub num_query_rounds: u32,
    /// sbn_fri_variant: 0 = default (= 1), 1 = plonky2 0.1.x FRI (final polynomial multiplied by X, PR #436), 2 = later upstream
    pub fri_variant: u32,
}

#[repr(C)]
pub struct sbn_prover {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct sbn_proof {
    _opaque: [u8; 0],
}
#[repr(C)]
pub struct sbn_batch_prover {
    _opaque: [u8; 0],
}

pub const SBN_AIR_G1_OP: i32 = 1;
pub const SBN_AIR_G1_EXP: i32 = 2;
pub const SBN_AIR_G2_EXP: i32 = 3;
pub const SBN_AIR_FQ12_EXP: i32 = 4;
pub const SBN_AIR_FQ_EXP: i32 = 5;
pub const SBN_AIR_FQ12_EXP_U64: i32 = 6;
/// the reference's single-operation test tables (`ModularStark`, `Fq12Stark`)
pub const SBN_AIR_MODULAR: i32 = 7;
pub const SBN_AIR_FQ12_MUL: i32 = 8;
/// the reference's unit-test tables `MyStark` (lookup.rs) and `FlagStark` (flags.rs)
pub const SBN_AIR_LOOKUP: i32 = 9;
pub const SBN_AIR_FLAGS: i32 = 10;
pub const SBN_AIR_FLAGS_U64: i32 = 11;

extern "C" {
    pub fn sbn_abi_version() -> i32;
    pub fn sbn_last_error() -> *const c_char;
    pub fn sbn_set_device(device: i32) -> i32;
    pub fn sbn_set_thread_device(device: i32) -> i32;
    pub fn sbn_device_count() -> i32;
    pub fn sbn_air_num_columns(air: *const sbn_air_desc) -> usize;
    pub fn sbn_air_num_public_inputs(air: *const sbn_air_desc) -> usize;

    pub fn sbn_prover_create(air: *const sbn_air_desc, cfg: *const sbn_config, degree_bits: u32, out: *mut *mut sbn_prover) -> i32;
    pub fn sbn_prover_destroy(p: *mut sbn_prover);
    pub fn sbn_prover_load_trace(p: *mut sbn_prover, trace_col_major: *const u64, public_inputs: *const u64, n_pi: usize) -> i32;
    pub fn sbn_prover_generate_trace(p: *mut sbn_prover, ios: *const u32, num_io: usize, pi_out: *mut u64) -> i32;
    pub fn sbn_prover_prove(p: *mut sbn_prover, out: *mut *mut sbn_proof) -> i32;
    pub fn sbn_prover_stage_times(p: *const sbn_prover, ms_out: *mut f32, cap: i32) -> i32;
    pub fn sbn_prover_stage_name(i: i32) -> *const c_char;
    pub fn sbn_prover_describe(p: *const sbn_prover, out: *mut c_char, cap: usize) -> i32;
    pub fn sbn_settings_check(out: *mut c_char, cap: usize) -> i32;

    pub fn sbn_batch_prover_create(air: *const sbn_air_desc, cfg: *const sbn_config, degree_bits: u32, inflight: u32, out: *mut *mut sbn_batch_prover) -> i32;
    pub fn sbn_batch_prover_prove_ios(b: *mut sbn_batch_prover, ios: *const u32, ios_words_per_unit: usize, num_io: usize, count: usize, proofs_out: *mut *mut sbn_proof) -> i32;
    pub fn sbn_batch_prover_destroy(b: *mut sbn_batch_prover);

    pub fn sbn_proof_num_words(p: *const sbn_proof) -> usize;
    pub fn sbn_proof_words(p: *const sbn_proof) -> *const u64;
    pub fn sbn_proof_free(p: *mut sbn_proof);
    pub fn sbn_verify(air: *const sbn_air_desc, cfg: *const sbn_config, bytes: *const u8, len: usize) -> i32;
}
