//! Reference-parity hand-over kit, Rust side (tests/parity_kit.py).  SOURCE ONLY: never compiled
//! here (no Rust toolchain in the build image).
//!
//! Drop this file into the `tests/` directory of qope/starky-bn254 (it uses that crate's own types), add
//! `serde_json = "1"`, `sha2 = "0.10"`, `hex = "0.4"` and `starky-bn254-amd = { path = ".../integration/rust/starky-bn254-amd" }`
//! to its `[dev-dependencies]`, and run
//!
//!     SBN_PARITY_KIT=<repo>/tests/golden/parity_kit RAYON_NUM_THREADS=1 cargo test --release --test parity -- --nocapture
//!
//! It feeds the committed inputs of a kit file to the REFERENCE's `generate_trace` + `generate_public_inputs` + starky's
//! `prove` (src/curves/g1/exp.rs:811-826), converts the proof to the library's canonical words
//! (`convert::words_from_proof`) and compares every stage with the committed digests of the `times_x` variant, in prove()
//! order, naming the FIRST stage that differs -- DESIGN.md section 4 maps each recalled protocol choice to that stage.
//!
//! The cheapest first contact is the kit's `mystark_lookup_fixed.json`: `MyStark` on `[6, 3, 1, 1, 0, 0, 0, 0]` / `0..7` is the one
//! workload whose inputs the reference itself fixes (src/utils/lookup.rs:154-161) -- no input plumbing at all.  `MyStark` is
//! private to lookup.rs's test module, so add, at the end of `test_mystark` (lookup.rs:228), after `verify_stark_proof(...)`:
//!
//!     let words = starky_bn254_amd::convert::words_from_proof::<F, C, D>(&proof, 3, &config).unwrap();
//!     println!("{:x?}", &words[12..76]);      // the trace cap: kit stages.times_x.trace_cap
//!     println!("{:x?}", &words[words.len() - 17..words.len() - 1]);   // final polynomial (8 ext coefficients) and, last, pow_witness
//!
//! and compare with the kit (8 rows: no FRI layer, a 1,093-word proof; `stages.times_x` vs `stages.plain` decides the x X question).
//! `RAYON_NUM_THREADS=1` makes plonky2's proof-of-work search (`find_any`) return the smallest witness, which is what the
//! kit holds; with more threads `pow_witness`, `pow_response`, `query_indices` and `query_rounds` may differ legitimately.
use ark_bn254::{Fq, G1Affine};
use plonky2::field::types::{Field, PrimeField64};
use plonky2::iop::challenger::Challenger;
use plonky2::plonk::config::{GenericConfig, PoseidonGoldilocksConfig};
use plonky2::util::timing::TimingTree;
use sha2::{Digest, Sha256};
use starky::prover::prove;
use starky_bn254::curves::g1::exp::{G1ExpIONative, G1ExpStark};
use starky_bn254_amd::convert::words_from_proof;

const D: usize = 2;
type C = PoseidonGoldilocksConfig;
type F = <C as GenericConfig<D>>::F;

fn sha_words(w: &[u64]) -> String {
    let mut h = Sha256::new();
    for x in w {
        h.update(x.to_le_bytes());
    }
    hex::encode(h.finalize())
}

fn fq_from_u32_limbs(l: &[u64]) -> Fq {
    // eight little-endian u32 limbs -> Fq (the inverse of the reference's `fq_to_u32_columns`-style flattening)
    let mut bytes = Vec::with_capacity(32);
    for x in l {
        bytes.extend((*x as u32).to_le_bytes());
    }
    use ark_ff::PrimeField;
    Fq::from_le_bytes_mod_order(&bytes)
}

fn hexes(v: &serde_json::Value) -> Vec<u64> {
    v.as_array().unwrap().iter().map(|x| u64::from_str_radix(x.as_str().unwrap().trim_start_matches("0x"), 16).unwrap()).collect()
}

#[test]
fn g1exp_io128_seed1_matches_the_committed_stage_digests() {
    let dir = std::env::var("SBN_PARITY_KIT").expect("SBN_PARITY_KIT = <repo>/tests/golden/parity_kit");
    let kit: serde_json::Value = serde_json::from_str(&std::fs::read_to_string(format!("{dir}/g1exp_io128_seed1.json")).unwrap()).unwrap();
    let num_io = kit["num_io"].as_u64().unwrap() as usize;
    let flat: Vec<u64> = kit["inputs_u32"].as_array().unwrap().iter().map(|x| x.as_u64().unwrap()).collect();
    assert_eq!(flat.len(), num_io * 40);
    // inputs_layout: x.x[8] x.y[8] offset.x[8] offset.y[8] exp_val[8]; output = x * exp_val + offset as the reference's test does
    let inputs: Vec<G1ExpIONative> = flat
        .chunks_exact(40)
        .map(|io| {
            let x = G1Affine::new(fq_from_u32_limbs(&io[0..8]), fq_from_u32_limbs(&io[8..16]));
            let offset = G1Affine::new(fq_from_u32_limbs(&io[16..24]), fq_from_u32_limbs(&io[24..32]));
            let exp_val: [u32; 8] = core::array::from_fn(|i| io[32 + i] as u32);
            let e: ark_bn254::Fr = num_bigint::BigUint::new(exp_val.to_vec()).into();
            let output: G1Affine = (x * e + offset).into();
            G1ExpIONative { x, offset, exp_val, output }
        })
        .collect();

    let stark = G1ExpStark::<F, D>::new(num_io);
    let config = stark.config();
    let trace = stark.generate_trace(&inputs);
    let pi = stark.generate_public_inputs(&inputs);

    // stage 0: the witness itself (column-major canonical u64, little endian)
    let mut h = Sha256::new();
    for col in &trace {
        for v in &col.values {
            h.update(v.to_canonical_u64().to_le_bytes());
        }
    }
    let trace_sha = hex::encode(h.finalize());
    println!("trace_sha256         {trace_sha}  (kit {})", kit["trace_sha256"].as_str().unwrap());
    let pi_words: Vec<u64> = pi.iter().map(|x| x.to_canonical_u64()).collect();
    println!("public_inputs_sha256 {}  (kit {})", sha_words(&pi_words), kit["public_inputs_sha256"].as_str().unwrap());
    assert_eq!(trace_sha, kit["trace_sha256"].as_str().unwrap(), "generate_trace differs: the witness restatement (A1-A12) is wrong, not the prover");

    let proof = prove::<F, C, _, D>(stark, &config, trace, pi.try_into().unwrap(), &mut TimingTree::default()).unwrap();
    let degree_bits = proof.proof.recover_degree_bits(&config);
    let words = words_from_proof::<F, C, D>(&proof, degree_bits, &config).unwrap();

    // cut the words exactly as tests/parity_kit.py parse_proof does
    let h = &words[..12];
    let (n_trace, n_zs, n_quot, n_pi) = (h[2] as usize, h[3] as usize, h[4] as usize, h[5] as usize);
    let (cap_height, n_layers, final_len) = (h[6] as usize, h[8] as usize, h[10] as usize);
    let capw = 4usize << cap_height;
    let mut pos = 12;
    let mut take = |n: usize| {
        let s = &words[pos..pos + n];
        pos += n;
        s
    };
    let trace_cap = take(capw);
    let zs_cap = take(if n_zs > 0 { capw } else { 0 });
    let quot_cap = take(capw);
    let local = take(2 * n_trace);
    let next = take(2 * n_trace);
    let zs = take(2 * n_zs);
    let zs_next = take(2 * n_zs);
    let quot = take(2 * n_quot);
    let fri_caps: Vec<&[u64]> = (0..n_layers).map(|_| take(capw)).collect();
    let tail = words.len() - (2 * final_len + 1 + n_pi);
    let query_rounds = &words[pos..tail];
    let final_poly = &words[tail..tail + 2 * final_len];
    let pow_witness = words[tail + 2 * final_len];

    // replay the transcript with plonky2's OWN challenger: the challenges the reference prover drew
    let mut ch = Challenger::<F, <C as GenericConfig<D>>::Hasher>::new();
    let obs = |ch: &mut Challenger<F, _>, w: &[u64]| {
        for x in w {
            ch.observe_element(F::from_canonical_u64(*x));
        }
    };
    let draw = |ch: &mut Challenger<F, _>, n: usize| -> Vec<u64> { (0..n).map(|_| ch.get_challenge().to_canonical_u64()).collect() };
    let st = &kit["stages"]["times_x"];
    let mut report = |name: &str, got: &[u64], want: &serde_json::Value| {
        let ok = got == hexes(want).as_slice();
        println!("{name:<24} {}", if ok { "ok" } else { "DIFFERS  <-- first place to look if nothing above differs" });
        ok
    };
    let mut all = true;
    all &= report("trace_cap", trace_cap, &st["trace_cap"]);
    obs(&mut ch, trace_cap);
    all &= report("permutation_challenges", &draw(&mut ch, 8), &st["permutation_challenges"]);
    all &= report("permutation_zs_cap", zs_cap, &st["permutation_zs_cap"]);
    obs(&mut ch, zs_cap);
    all &= report("alphas", &draw(&mut ch, 2), &st["alphas"]);
    all &= report("quotient_polys_cap", quot_cap, &st["quotient_polys_cap"]);
    obs(&mut ch, quot_cap);
    all &= report("zeta", &draw(&mut ch, 2), &st["zeta"]);
    for (name, sec) in [("local_values", local), ("next_values", next), ("permutation_zs", zs), ("permutation_zs_next", zs_next), ("quotient_polys", quot)] {
        let ok = sha_words(sec) == st["openings"][name]["sha256"].as_str().unwrap();
        println!("openings.{name:<15} {}", if ok { "ok" } else { "DIFFERS" });
        all &= ok;
    }
    for sec in [local, zs, quot, next, zs_next] {
        obs(&mut ch, sec);
    }
    all &= report("fri_alpha", &draw(&mut ch, 2), &st["fri_alpha"]);
    for (i, cap) in fri_caps.iter().enumerate() {
        let ok = sha_words(cap) == st["fri_commit_caps"][i]["sha256"].as_str().unwrap();
        println!("fri_commit_caps[{i}]       {}", if ok { "ok" } else { "DIFFERS" });
        all &= ok;
        obs(&mut ch, cap);
        all &= report("fri_beta", &draw(&mut ch, 2), &st["fri_betas"][i]);
    }
    all &= report("final_poly", final_poly, &st["final_poly"]);
    obs(&mut ch, final_poly);
    println!("pow_witness              {} (kit {}; equal only with RAYON_NUM_THREADS=1)", pow_witness, st["pow_witness"]);
    let same_pow = pow_witness == st["pow_witness"].as_u64().unwrap();
    if same_pow {
        let ok = sha_words(query_rounds) == st["query_rounds"]["sha256"].as_str().unwrap();
        println!("query_rounds             {}", if ok { "ok" } else { "DIFFERS" });
        all &= ok;
        let ok = sha_words(&words) == st["proof"]["sha256"].as_str().unwrap();
        println!("proof                    {}", if ok { "ok: the library's proof bytes ARE the reference's" } else { "DIFFERS" });
        all &= ok;
    }
    assert!(all, "a stage differs: see the first DIFFERS line above and DESIGN.md section 4");
}
