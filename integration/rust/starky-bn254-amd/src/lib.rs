//! Rust face of libsbn254.so for qope/starky-bn254: starky 0.1.1's `prove` and `verify_stark_proof` WITH THEIR OWN
//! SIGNATURES over the C ABI of include/sbn.h, so that the reference's call sites -- the tests
//! (src/curves/g1/exp.rs:818-826) and `*StarkyProofGenerator::run_once` (src/curves/g1/circuit.rs:192-200 and its G2 /
//! Fq12 siblings) -- switch to the MI355X path by changing `use starky::prover::prove` / `use
//! starky::verifier::verify_stark_proof` to `use starky_bn254_amd::{prove, verify_stark_proof}` and adding one
//! `impl SbnTable` per table; the call expressions themselves stay as they are:
//!
//! ```ignore
//! let inner_proof = prove::<F, C, _, D>(stark, &inner_config, trace, pi.try_into().unwrap(), &mut TimingTree::default()).unwrap();
//! verify_stark_proof(stark, inner_proof.clone(), &inner_config).unwrap();
//! ```
//!
//! SOURCE ONLY -- written without a Rust toolchain in the build image and never compiled there.  What is tested is
//! the C ABI below it (tests/test_gpu_parity.py through ctypes) and, textually, that these signatures have the argument
//! lists of the reference's call sites (tests/test_product_host.py); the struct field names follow starky/plonky2 at
//! rev 541e127 as recalled, so expect to touch `convert.rs` if that fork renamed a field.
pub mod convert;
pub mod ffi;

use anyhow::{anyhow, ensure, Result};
use plonky2::field::extension::Extendable;
use plonky2::field::polynomial::PolynomialValues;
use plonky2::field::types::PrimeField64;
use plonky2::hash::hash_types::RichField;
use plonky2::hash::poseidon::PoseidonHash;
use plonky2::plonk::config::GenericConfig;
use plonky2::util::timing::TimingTree;
use starky::config::StarkConfig;
use starky::proof::StarkProofWithPublicInputs;
use starky::stark::Stark;
use std::ffi::CStr;
use std::ptr;

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

/// Which FRI the linked plonky2 speaks (include/sbn.h `sbn_config.fri_variant`): 1 = SBN_FRI_TIMES_X, 2 = SBN_FRI_PLAIN.
pub const FRI_VARIANT: u32 = 1;

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
        // fri/oracle.rs::prove_openings; set to 2 (SBN_FRI_PLAIN) when building against a plonky2 that dropped the step.
        fri_variant: FRI_VARIANT,
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
    pub fn prove<F, C, const D: usize>(&mut self, trace: Vec<PolynomialValues<F>>, public_inputs: &[F]) -> Result<StarkProofWithPublicInputs<F, C, D>>
    where
        F: RichField + Extendable<D>,
        C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    {
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
        self.finish::<F, C, D>()
    }

    /// Witness generated on the device from the instance list (`G1ExpIONative` etc. flattened to u32 limbs as
    /// include/sbn.h documents per table); no trace on the host at all.  Returns the proof; its `public_inputs` are the
    /// ones `generate_public_inputs` would have produced.
    pub fn prove_ios<F, C, const D: usize>(&mut self, ios: &[u32]) -> Result<StarkProofWithPublicInputs<F, C, D>>
    where
        F: RichField + Extendable<D>,
        C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    {
        ensure!(self.io_words > 0, "device witness generation covers the Exp tables");
        ensure!(ios.len() == self.io_words * self.num_io, "ios must hold {} u32 words per instance x {} instances", self.io_words, self.num_io);
        let mut pi = vec![0u64; self.n_pi];
        check(unsafe { ffi::sbn_prover_generate_trace(self.raw, ios.as_ptr(), self.num_io, pi.as_mut_ptr()) }, "sbn_prover_generate_trace")?;
        self.finish::<F, C, D>()
    }

    fn finish<F, C, const D: usize>(&mut self) -> Result<StarkProofWithPublicInputs<F, C, D>>
    where
        F: RichField + Extendable<D>,
        C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    {
        let mut p = ptr::null_mut();
        check(unsafe { ffi::sbn_prover_prove(self.raw, &mut p) }, "sbn_prover_prove")?;
        let words = unsafe { std::slice::from_raw_parts(ffi::sbn_proof_words(p), ffi::sbn_proof_num_words(p)) };
        let proof = convert::proof_from_words::<F, C, D>(words);
        unsafe { ffi::sbn_proof_free(p) };
        proof
    }

    /// Per-stage device times of the last prove() in milliseconds (sbn_prover_stage_times / sbn_prover_stage_name).
    pub fn stage_times(&self) -> Vec<(String, f32)> {
        let mut ms = [0f32; 32];
        let k = unsafe { ffi::sbn_prover_stage_times(self.raw, ms.as_mut_ptr(), 32) } as usize;
        (0..k)
            .map(|i| (unsafe { CStr::from_ptr(ffi::sbn_prover_stage_name(i as i32)) }.to_string_lossy().into_owned(), ms[i]))
            .collect()
    }
}

impl Drop for Prover {
    fn drop(&mut self) {
        unsafe { ffi::sbn_prover_destroy(self.raw) }
    }
}

/// starky 0.1.1 `prover::prove`, signature for signature (the fork the reference pins takes the public inputs as a `Vec`,
/// which is why its call sites write `pi.try_into().unwrap()`: src/curves/g1/exp.rs:822, src/curves/g1/circuit.rs:196):
///
/// `prove::<F, C, S, D>(stark, &config, trace_poly_values, public_inputs, &mut timing) -> Result<StarkProofWithPublicInputs<F, C, D>>`
///
/// `S: Stark<F, D>` is kept so that the call sites type-check unchanged; `SbnTable` names the table for the C ABI (the
/// constraint code of `S::eval_packed_generic` lives natively on the far side, include/sbn.h).  `C::Hasher` must be
/// Poseidon -- the only hasher the reference uses (`PoseidonGoldilocksConfig`).  `timing` gets one scope for the call; the
/// per-stage device times are logged at debug level (`Prover::stage_times`).
pub fn prove<F, C, S, const D: usize>(
    stark: S,
    config: &StarkConfig,
    trace_poly_values: Vec<PolynomialValues<F>>,
    public_inputs: Vec<F>,
    timing: &mut TimingTree,
) -> Result<StarkProofWithPublicInputs<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    S: Stark<F, D> + SbnTable,
{
    let n = trace_poly_values.first().map(|c| c.len()).unwrap_or(0);
    ensure!(n.is_power_of_two(), "trace height must be a power of two");
    timing.push("sbn prove (MI355X)", log::Level::Debug);
    let mut prover = Prover::new(&stark, config, n.trailing_zeros() as usize)?;
    let proof = prover.prove::<F, C, D>(trace_poly_values, &public_inputs);
    for (name, ms) in prover.stage_times() {
        log::debug!("sbn stage {name}: {ms:.3} ms");
    }
    timing.pop();
    proof
}

/// starky 0.1.1 `verifier::verify_stark_proof`, signature for signature (src/curves/g1/exp.rs:826, circuit.rs:200):
/// the proof is serialised back to the library's canonical words (`convert::words_from_proof`) and checked by the
/// library's host verifier (no GPU needed).  The reference's own starky verifier can be kept instead -- then the two
/// verifiers are compared on every proof, which is the parity campaign of DESIGN.md section 6; whether it accepts hinges on
/// the recalled protocol details of DESIGN.md section 4, first of all on `FRI_VARIANT` matching the linked plonky2.
pub fn verify_stark_proof<F, C, S, const D: usize>(stark: S, proof_with_pis: StarkProofWithPublicInputs<F, C, D>, config: &StarkConfig) -> Result<()>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    S: Stark<F, D> + SbnTable,
{
    let degree_bits = proof_with_pis.proof.recover_degree_bits(config);
    let words = convert::words_from_proof::<F, C, D>(&proof_with_pis, degree_bits, config)?;
    verify_stark_proof_words(&stark, &words, config)
}

/// Several proofs in flight on one GPU (BASELINE config 2): `units` instance lists of `num_io` instances each.
pub fn prove_batch<F, C, S, const D: usize>(stark: &S, config: &StarkConfig, degree_bits: usize, inflight: usize, ios: &[u32], units: usize) -> Result<Vec<StarkProofWithPublicInputs<F, C, D>>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
    S: SbnTable,
{
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
            let proof = convert::proof_from_words::<F, C, D>(words);
            unsafe { ffi::sbn_proof_free(p) };
            proof
        })
        .collect()
}

/// The library's host verifier on canonical proof words (what `verify_stark_proof` above ends in).
pub fn verify_stark_proof_words<S: SbnTable>(stark: &S, words: &[u64], config: &StarkConfig) -> Result<()> {
    let a = air(stark);
    let cfg = to_sbn_config(config)?;
    let bytes = unsafe { std::slice::from_raw_parts(words.as_ptr() as *const u8, words.len() * 8) };
    check(unsafe { ffi::sbn_verify(&a, &cfg, bytes.as_ptr(), bytes.len()) }, "sbn_verify")
}

pub fn set_device(device: usize) -> Result<()> {
    check(unsafe { ffi::sbn_set_device(device as i32) }, "sbn_set_device")
}
