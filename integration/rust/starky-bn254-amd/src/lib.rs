//! Rust face of libsbn254.so for qope/starky-bn254: the `prove` / `verify_stark_proof` call shape of starky 0.1.1 over
//! the C ABI of include/sbn.h, so that `*StarkyProofGenerator::run_once` (src/curves/g1/circuit.rs:161-202 and its
//! G2 / Fq12 siblings) switches to the MI355X path by changing its imports.
//!
//! SOURCE ONLY — written without a Rust toolchain in the build image and never compiled there.  What is tested is
//! the C ABI below it (tests/test_gpu_parity.py through ctypes); the struct field names follow starky/plonky2 at
//! rev 541e127 as recalled, so expect to touch `convert.rs` if that fork renamed a field.
pub mod convert;
pub mod ffi;

use anyhow::{anyhow, ensure, Result};
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::PrimeField64;
use plonky2::plonk::config::PoseidonGoldilocksConfig;
use starky::config::StarkConfig;
use starky::proof::StarkProofWithPublicInputs;
use std::ffi::CStr;
use std::ptr;

type F = GoldilocksField;
type C = PoseidonGoldilocksConfig;
const D: usize = 2;
pub type Proof = StarkProofWithPublicInputs<F, C, D>;

/// A table of the reference named the way the C ABI names it: (kind, num_io).  Implemented in the reference crate for
/// its stark types, e.g. `impl SbnTable for G1ExpStark<F, D> { fn desc(&self) -> (i32, usize) { (ffi::SBN_AIR_G1_EXP, self.num_io) } }`
/// (G1ExpStark keeps `num_io`, src/curves/g1/exp.rs:232-236).
pub trait SbnTable {
    fn desc(&self) -> (i32, usize);
}

fn air(t: &impl SbnTable) -> ffi::sbn_air_desc {
    let (kind, num_io) = t.desc();
    ffi::sbn_air_desc { kind, num_io: num_io as u32 }
}

fn last_error() -> String {
    unsafe { CStr::from_ptr(ffi::sbn_last_error()).to_string_lossy().into_owned() }
}

fn check(rc: i32, what: &str) -> Result<()> {
    if rc == 0 {
        Ok(())
    } else {
        Err(anyhow!("{what} failed ({rc}): {}", last_error()))
    }
}

/// Which FRI the linked plonky2 speaks (include/sbn.h `sbn_config.fri_final_poly_times_x`).
pub const FRI_FINAL_POLY_TIMES_X: u32 = 1;

/// `StarkConfig` -> `sbn_config`.  Only `FriReductionStrategy::ConstantArityBits` is supported, which is what
/// `standard_fast_config` uses (exp.rs:250-253).
pub fn to_sbn_config(c: &StarkConfig) -> Result<ffi::sbn_config> {
    use plonky2::fri::reduction_strategies::FriReductionStrategy::ConstantArityBits;
    let (arity_bits, final_poly_bits) = match c.fri_config.reduction_strategy {
        ConstantArityBits(a, f) => (a as u32, f as u32),
        _ => return Err(anyhow!("only ConstantArityBits FRI reduction is supported")),
    };
    Ok(ffi::sbn_config {
        security_bits: c.security_bits as u32,
        num_challenges: c.num_challenges as u32,
        rate_bits: c.fri_config.rate_bits as u32,
        cap_height: c.fri_config.cap_height as u32,
        proof_of_work_bits: c.fri_config.proof_of_work_bits,
        fri_arity_bits: arity_bits,
        fri_final_poly_bits: final_poly_bits,
        num_query_rounds: c.fri_config.num_query_rounds as u32,
        // plonky2 0.1.3 @ 541e127 (the fork this crate links) still multiplies the final polynomial by X in
        // fri/oracle.rs::prove_openings; set to 0 when building against a plonky2 that dropped the step.
        fri_final_poly_times_x: FRI_FINAL_POLY_TIMES_X,
    })
}

/// Device context for one (table, degree_bits): HBM buffers, streams, twiddles.  Create once, prove many times.
pub struct Prover {
    raw: *mut ffi::sbn_prover,
    num_io: usize,
    n_pi: usize,
    n_cols: usize,
    degree_bits: usize,
    io_words: usize,
}
// one prover belongs to one thread at a time (include/sbn.h, "Threading")
unsafe impl Send for Prover {}

impl Prover {
    pub fn new(stark: &impl SbnTable, config: &StarkConfig, degree_bits: usize) -> Result<Self> {
        let a = air(stark);
        let cfg = to_sbn_config(config)?;
        let mut raw = ptr::null_mut();
        check(unsafe { ffi::sbn_prover_create(&a, &cfg, degree_bits as u32, &mut raw) }, "sbn_prover_create")?;
        // u32 words of one instance in `ios` (include/sbn.h, per table)
        let io_words = match a.kind {
            ffi::SBN_AIR_G1_EXP => 40,
            ffi::SBN_AIR_G2_EXP => 72,
            ffi::SBN_AIR_FQ12_EXP => 200,
            ffi::SBN_AIR_FQ_EXP => 24,
            ffi::SBN_AIR_FQ12_EXP_U64 => 194,
            _ => 0,
        };
        Ok(Self {
            raw,
            num_io: a.num_io as usize,
            n_pi: unsafe { ffi::sbn_air_num_public_inputs(&a) },
            n_cols: unsafe { ffi::sbn_air_num_columns(&a) },
            degree_bits,
            io_words,
        })
    }

    /// `prove(stark, &config, trace, pi, &mut timing)` with a host-built trace: column-major already, one copy to
    /// flatten it and one PCIe transfer (0.88 GB for G1ExpStark(128)).
    pub fn prove(&mut self, trace: Vec<PolynomialValues<F>>, public_inputs: &[F]) -> Result<Proof> {
        let n = trace.first().map(|c| c.len()).unwrap_or(0);
        // sbn_prover_load_trace reads num_columns * 2^degree_bits words and n_pi public inputs: check before the FFI call
        ensure!(trace.len() == self.n_cols, "the table has {} columns, the trace {}", self.n_cols, trace.len());
        ensure!(n == 1usize << self.degree_bits, "the prover was created for 2^{} rows, the trace has {}", self.degree_bits, n);
        ensure!(public_inputs.len() == self.n_pi, "expected {} public inputs, got {}", self.n_pi, public_inputs.len());
        let mut flat = Vec::with_capacity(trace.len() * n);
        for col in &trace {
            ensure!(col.len() == n, "ragged trace");
            flat.extend(col.values.iter().map(|x| x.to_canonical_u64()));
        }
        drop(trace);
        let pis: Vec<u64> = public_inputs.iter().map(|x| x.to_canonical_u64()).collect();
        check(unsafe { ffi::sbn_prover_load_trace(self.raw, flat.as_ptr(), pis.as_ptr(), pis.len()) }, "sbn_prover_load_trace")?;
        self.finish()
    }

    /// Witness generated on the device from the instance list (`G1ExpIONative` etc. flattened to u32 limbs as
    /// include/sbn.h documents per table); no trace on the host at all.  Returns the proof; its `public_inputs` are the
    /// ones `generate_public_inputs` would have produced.
    pub fn prove_ios(&mut self, ios: &[u32]) -> Result<Proof> {
        ensure!(self.io_words > 0, "device witness generation covers the Exp tables");
        ensure!(ios.len() == self.io_words * self.num_io, "ios must hold {} u32 words per instance x {} instances", self.io_words, self.num_io);
        let mut pi = vec![0u64; self.n_pi];
        check(unsafe { ffi::sbn_prover_generate_trace(self.raw, ios.as_ptr(), self.num_io, pi.as_mut_ptr()) }, "sbn_prover_generate_trace")?;
        self.finish()
    }

    fn finish(&mut self) -> Result<Proof> {
        let mut p = ptr::null_mut();
        check(unsafe { ffi::sbn_prover_prove(self.raw, &mut p) }, "sbn_prover_prove")?;
        let words = unsafe { std::slice::from_raw_parts(ffi::sbn_proof_words(p), ffi::sbn_proof_num_words(p)) };
        let proof = convert::proof_from_words(words);
        unsafe { ffi::sbn_proof_free(p) };
        proof
    }
}

impl Drop for Prover {
    fn drop(&mut self) {
        unsafe { ffi::sbn_prover_destroy(self.raw) }
    }
}

/// One-shot drop-in for starky's `prove::<F, C, S, D>(stark, &config, trace, public_inputs, &mut timing)`.
pub fn prove<S: SbnTable>(stark: S, config: &StarkConfig, trace: Vec<PolynomialValues<F>>, public_inputs: Vec<F>) -> Result<Proof> {
    let n = trace.first().map(|c| c.len()).unwrap_or(0);
    ensure!(n.is_power_of_two(), "trace height must be a power of two");
    Prover::new(&stark, config, n.trailing_zeros() as usize)?.prove(trace, &public_inputs)
}

/// Several proofs in flight on one GPU (BASELINE config 2): `units` instance lists of `num_io` instances each.
pub fn prove_batch<S: SbnTable>(stark: &S, config: &StarkConfig, degree_bits: usize, inflight: usize, ios: &[u32], units: usize) -> Result<Vec<Proof>> {
    let a = air(stark);
    let cfg = to_sbn_config(config)?;
    ensure!(units > 0 && ios.len() % units == 0, "ios length is not a multiple of the unit count");
    let mut b = ptr::null_mut();
    check(unsafe { ffi::sbn_batch_prover_create(&a, &cfg, degree_bits as u32, inflight as u32, &mut b) }, "sbn_batch_prover_create")?;
    let mut raw = vec![ptr::null_mut(); units];
    let rc = unsafe { ffi::sbn_batch_prover_prove_ios(b, ios.as_ptr(), ios.len() / units, a.num_io as usize, units, raw.as_mut_ptr()) };
    unsafe { ffi::sbn_batch_prover_destroy(b) };
    check(rc, "sbn_batch_prover_prove_ios")?;
    raw.into_iter()
        .map(|p| {
            let words = unsafe { std::slice::from_raw_parts(ffi::sbn_proof_words(p), ffi::sbn_proof_num_words(p)) };
            let proof = convert::proof_from_words(words);
            unsafe { ffi::sbn_proof_free(p) };
            proof
        })
        .collect()
}

/// `verify_stark_proof(stark, proof, &config)` on the library's host verifier (no GPU needed).  Whether the reference's
/// own starky verifier accepts the converted proof is UNTESTED (this crate has never been compiled: no Rust toolchain in
/// the build image); it hinges on the recalled protocol details of DESIGN.md section 4, first of all on
/// `FRI_FINAL_POLY_TIMES_X` matching the linked plonky2.  This function exists so that the two verifiers can be compared.
pub fn verify_stark_proof_words<S: SbnTable>(stark: &S, words: &[u64], config: &StarkConfig) -> Result<()> {
    let a = air(stark);
    let cfg = to_sbn_config(config)?;
    let bytes = unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, words.len() * 8) };
    check(unsafe { ffi::sbn_verify(&a, &cfg, bytes.as_ptr(), bytes.len()) }, "sbn_verify")
}

pub fn set_device(device: usize) -> Result<()> {
    check(unsafe { ffi::sbn_set_device(device as i32) }, "sbn_set_device")
}
