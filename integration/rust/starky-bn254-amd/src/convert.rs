//! Canonical proof words (layout: include/sbn.h:20-31) -> `StarkProofWithPublicInputs`.
//!
//! Field order follows the struct declarations of starky `proof.rs` (`StarkProof`, `StarkOpeningSet`) and plonky2
//! `fri/proof.rs` (`FriProof`, `FriQueryRound`, `FriInitialTreeProof`, `FriQueryStep`), which is also the order the
//! reference's recursive verifier reads them in `set_stark_proof_with_pis_target` (src/curves/g1/circuit.rs:201).
use anyhow::{ensure, Result};
use plonky2::field::extension::quadratic::QuadraticExtension;
use plonky2::field::extension::FieldExtension;
use plonky2::field::goldilocks_field::GoldilocksField;
use plonky2::field::polynomial::PolynomialCoeffs;
use plonky2::field::types::Field;
use plonky2::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use plonky2::hash::hash_types::HashOut;
use plonky2::hash::merkle_proofs::MerkleProof;
use plonky2::hash::merkle_tree::MerkleCap;
use plonky2::hash::poseidon::PoseidonHash;
use plonky2::plonk::config::PoseidonGoldilocksConfig;
use starky::proof::{StarkOpeningSet, StarkProof, StarkProofWithPublicInputs};

type F = GoldilocksField;
type FE = QuadraticExtension<F>;
type C = PoseidonGoldilocksConfig;
type H = PoseidonHash;
const D: usize = 2;

const MAGIC: u64 = u64::from_le_bytes(*b"SNBPROV1");

struct Reader<'a> {
    w: &'a [u64],
    pos: usize,
}

impl<'a> Reader<'a> {
    fn take(&mut self, n: usize) -> Result<&'a [u64]> {
        ensure!(self.pos + n <= self.w.len(), "proof words truncated");
        let s = &self.w[self.pos..self.pos + n];
        self.pos += n;
        Ok(s)
    }
    fn f(&mut self) -> Result<F> {
        // the library only emits canonical values; from_canonical_u64 debug-asserts that
        Ok(F::from_canonical_u64(self.take(1)?[0]))
    }
    fn fs(&mut self, n: usize) -> Result<Vec<F>> {
        Ok(self.take(n)?.iter().map(|&x| F::from_canonical_u64(x)).collect())
    }
    fn exts(&mut self, n: usize) -> Result<Vec<FE>> {
        Ok(self
            .take(2 * n)?
            .chunks_exact(2)
            .map(|c| FE::from_basefield_array([F::from_canonical_u64(c[0]), F::from_canonical_u64(c[1])]))
            .collect())
    }
    fn hash(&mut self) -> Result<HashOut<F>> {
        let e = self.fs(4)?;
        Ok(HashOut { elements: [e[0], e[1], e[2], e[3]] })
    }
    fn cap(&mut self, cap_height: usize) -> Result<MerkleCap<F, H>> {
        Ok(MerkleCap((0..1usize << cap_height).map(|_| self.hash()).collect::<Result<Vec<_>>>()?))
    }
    fn merkle_proof(&mut self, len: usize) -> Result<MerkleProof<F, H>> {
        Ok(MerkleProof { siblings: (0..len).map(|_| self.hash()).collect::<Result<Vec<_>>>()? })
    }
}

pub fn proof_from_words(words: &[u64]) -> Result<StarkProofWithPublicInputs<F, C, D>> {
    let mut r = Reader { w: words, pos: 0 };
    let h = r.take(12)?;
    ensure!(h[0] == MAGIC, "not a proof of this library");
    let (degree_bits, n_trace, n_zs, n_quot, n_pi) = (h[1] as usize, h[2] as usize, h[3] as usize, h[4] as usize, h[5] as usize);
    let (cap_height, rate_bits, n_layers, arity_bits, final_len, n_queries) =
        (h[6] as usize, h[7] as usize, h[8] as usize, h[9] as usize, h[10] as usize, h[11] as usize);
    let lde_bits = degree_bits + rate_bits;

    let trace_cap = r.cap(cap_height)?;
    let permutation_zs_cap = if n_zs > 0 { Some(r.cap(cap_height)?) } else { None };
    let quotient_polys_cap = r.cap(cap_height)?;

    let local_values = r.exts(n_trace)?;
    let next_values = r.exts(n_trace)?;
    let (permutation_zs, permutation_zs_next) = if n_zs > 0 { (Some(r.exts(n_zs)?), Some(r.exts(n_zs)?)) } else { (None, None) };
    let quotient_polys = r.exts(n_quot)?;
    let openings = StarkOpeningSet { local_values, next_values, permutation_zs, permutation_zs_next, quotient_polys };

    let commit_phase_merkle_caps = (0..n_layers).map(|_| r.cap(cap_height)).collect::<Result<Vec<_>>>()?;

    // leaf widths of the initial oracles, in commitment order: trace, permutation Zs (if any), quotient chunks
    let mut widths = vec![n_trace];
    if n_zs > 0 {
        widths.push(n_zs);
    }
    widths.push(n_quot);
    let mut query_round_proofs = Vec::with_capacity(n_queries);
    for _ in 0..n_queries {
        let mut evals_proofs = Vec::with_capacity(widths.len());
        for &w in &widths {
            let leaf = r.fs(w)?;
            evals_proofs.push((leaf, r.merkle_proof(lde_bits - cap_height)?));
        }
        let mut steps = Vec::with_capacity(n_layers);
        let mut bits = lde_bits;
        for _ in 0..n_layers {
            let evals = r.exts(1 << arity_bits)?;
            bits -= arity_bits;
            // a layer's tree has 2^bits leaves of 2^arity_bits evaluations; paths stop at the cap
            steps.push(FriQueryStep { evals, merkle_proof: r.merkle_proof(bits.saturating_sub(cap_height))? });
        }
        query_round_proofs.push(FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps });
    }
    let final_poly = PolynomialCoeffs::new(r.exts(final_len)?);
    let pow_witness = r.f()?;
    let public_inputs = r.fs(n_pi)?;
    ensure!(r.pos == words.len(), "trailing words after the proof");

    let opening_proof = FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness };
    Ok(StarkProofWithPublicInputs {
        proof: StarkProof { trace_cap, permutation_zs_cap, quotient_polys_cap, openings, opening_proof },
        public_inputs,
    })
}
