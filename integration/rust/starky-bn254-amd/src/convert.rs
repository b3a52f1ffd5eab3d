//! Canonical proof words (layout: include/sbn.h:20-31) -> `StarkProofWithPublicInputs`.
//!
//! Field order follows the struct declarations of starky `proof.rs` (`StarkProof`, `StarkOpeningSet`) and plonky2
//! `fri/proof.rs` (`FriProof`, `FriQueryRound`, `FriInitialTreeProof`, `FriQueryStep`), which is also the order the
//! reference's recursive verifier reads them in `set_stark_proof_with_pis_target` (src/curves/g1/circuit.rs:201).
use anyhow::{ensure, Result};
use plonky2::field::extension::{Extendable, FieldExtension};
use plonky2::field::polynomial::PolynomialCoeffs;
use plonky2::field::types::{Field, PrimeField64};
use plonky2::fri::proof::{FriInitialTreeProof, FriProof, FriQueryRound, FriQueryStep};
use plonky2::hash::hash_types::{HashOut, RichField};
use plonky2::hash::merkle_proofs::MerkleProof;
use plonky2::hash::merkle_tree::MerkleCap;
use plonky2::hash::poseidon::PoseidonHash;
use plonky2::plonk::config::GenericConfig;
use starky::config::StarkConfig;
use starky::proof::{StarkOpeningSet, StarkProof, StarkProofWithPublicInputs};

type H = PoseidonHash;

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
    fn f<F: RichField>(&mut self) -> Result<F> {
        // the library only emits canonical values; from_canonical_u64 debug-asserts that
        Ok(F::from_canonical_u64(self.take(1)?[0]))
    }
    fn fs<F: RichField>(&mut self, n: usize) -> Result<Vec<F>> {
        Ok(self.take(n)?.iter().map(|&x| F::from_canonical_u64(x)).collect())
    }
    /// extension elements are two words (c0, c1): the library speaks the quadratic extension only (D = 2)
    fn exts<F: RichField + Extendable<D>, const D: usize>(&mut self, n: usize) -> Result<Vec<F::Extension>> {
        Ok(self
            .take(2 * n)?
            .chunks_exact(2)
            .map(|c| {
                let mut a = [F::ZERO; D];
                a[0] = F::from_canonical_u64(c[0]);
                a[1] = F::from_canonical_u64(c[1]);
                F::Extension::from_basefield_array(a)
            })
            .collect())
    }
    fn hash<F: RichField>(&mut self) -> Result<HashOut<F>> {
        let e = self.fs::<F>(4)?;
        Ok(HashOut { elements: [e[0], e[1], e[2], e[3]] })
    }
    fn cap<F: RichField>(&mut self, cap_height: usize) -> Result<MerkleCap<F, H>> {
        Ok(MerkleCap((0..1usize << cap_height).map(|_| self.hash()).collect::<Result<Vec<_>>>()?))
    }
    fn merkle_proof<F: RichField>(&mut self, len: usize) -> Result<MerkleProof<F, H>> {
        Ok(MerkleProof { siblings: (0..len).map(|_| self.hash()).collect::<Result<Vec<_>>>()? })
    }
}

pub fn proof_from_words<F, C, const D: usize>(words: &[u64]) -> Result<StarkProofWithPublicInputs<F, C, D>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
{
    ensure!(D == 2, "the library proves over the quadratic extension (D = 2)");
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

    let local_values = r.exts::<F, D>(n_trace)?;
    let next_values = r.exts::<F, D>(n_trace)?;
    let (permutation_zs, permutation_zs_next) = if n_zs > 0 { (Some(r.exts::<F, D>(n_zs)?), Some(r.exts::<F, D>(n_zs)?)) } else { (None, None) };
    let quotient_polys = r.exts::<F, D>(n_quot)?;
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
            let leaf = r.fs::<F>(w)?;
            evals_proofs.push((leaf, r.merkle_proof(lde_bits - cap_height)?));
        }
        let mut steps = Vec::with_capacity(n_layers);
        let mut bits = lde_bits;
        for _ in 0..n_layers {
            let evals = r.exts::<F, D>(1 << arity_bits)?;
            bits -= arity_bits;
            // a layer's tree has 2^bits leaves of 2^arity_bits evaluations; paths stop at the cap
            steps.push(FriQueryStep { evals, merkle_proof: r.merkle_proof(bits.saturating_sub(cap_height))? });
        }
        query_round_proofs.push(FriQueryRound { initial_trees_proof: FriInitialTreeProof { evals_proofs }, steps });
    }
    let final_poly = PolynomialCoeffs::new(r.exts::<F, D>(final_len)?);
    let pow_witness = r.f::<F>()?;
    let public_inputs = r.fs::<F>(n_pi)?;
    ensure!(r.pos == words.len(), "trailing words after the proof");

    let opening_proof = FriProof { commit_phase_merkle_caps, query_round_proofs, final_poly, pow_witness };
    Ok(StarkProofWithPublicInputs {
        proof: StarkProof { trace_cap, permutation_zs_cap, quotient_polys_cap, openings, opening_proof },
        public_inputs,
    })
}

/// `StarkProofWithPublicInputs` -> canonical proof words: the inverse of `proof_from_words`, field by field in the order of
/// include/sbn.h:20-31, so that `verify_stark_proof(stark, proof, &config)` can hand a starky proof object to the library's
/// verifier (and so that a proof made by the REFERENCE prover can be diffed word for word against the library's: the parity
/// campaign, tests/parity.rs).  `degree_bits` = `proof.recover_degree_bits(config)`.
pub fn words_from_proof<F, C, const D: usize>(p: &StarkProofWithPublicInputs<F, C, D>, degree_bits: usize, config: &StarkConfig) -> Result<Vec<u64>>
where
    F: RichField + Extendable<D>,
    C: GenericConfig<D, F = F, Hasher = PoseidonHash>,
{
    ensure!(D == 2, "the library proves over the quadratic extension (D = 2)");
    let pr = &p.proof;
    let fri = &pr.opening_proof;
    let cap_height = config.fri_config.cap_height;
    let n_zs = pr.openings.permutation_zs.as_ref().map(|v| v.len()).unwrap_or(0);
    let arity_bits = fri.query_round_proofs.first().and_then(|q| q.steps.first()).map(|s| s.evals.len().trailing_zeros() as usize).unwrap_or(0);
    let mut w: Vec<u64> = vec![
        MAGIC,
        degree_bits as u64,
        pr.openings.local_values.len() as u64,
        n_zs as u64,
        pr.openings.quotient_polys.len() as u64,
        p.public_inputs.len() as u64,
        cap_height as u64,
        config.fri_config.rate_bits as u64,
        fri.commit_phase_merkle_caps.len() as u64,
        arity_bits as u64,
        fri.final_poly.coeffs.len() as u64,
        fri.query_round_proofs.len() as u64,
    ];
    fn put_hash<F: RichField>(w: &mut Vec<u64>, h: &HashOut<F>) {
        w.extend(h.elements.iter().map(|x| x.to_canonical_u64()));
    }
    fn put_exts<F: RichField + Extendable<D>, const D: usize>(w: &mut Vec<u64>, v: &[F::Extension]) {
        for e in v {
            let a: [F; D] = e.to_basefield_array();
            w.push(a[0].to_canonical_u64());
            w.push(a[1].to_canonical_u64());
        }
    }
    for h in &pr.trace_cap.0 {
        put_hash(&mut w, h);
    }
    if let Some(cap) = &pr.permutation_zs_cap {
        for h in &cap.0 {
            put_hash(&mut w, h);
        }
    }
    for h in &pr.quotient_polys_cap.0 {
        put_hash(&mut w, h);
    }
    put_exts::<F, D>(&mut w, &pr.openings.local_values);
    put_exts::<F, D>(&mut w, &pr.openings.next_values);
    if let (Some(z), Some(zn)) = (&pr.openings.permutation_zs, &pr.openings.permutation_zs_next) {
        put_exts::<F, D>(&mut w, z);
        put_exts::<F, D>(&mut w, zn);
    }
    put_exts::<F, D>(&mut w, &pr.openings.quotient_polys);
    for cap in &fri.commit_phase_merkle_caps {
        for h in &cap.0 {
            put_hash(&mut w, h);
        }
    }
    for q in &fri.query_round_proofs {
        for (leaf, path) in &q.initial_trees_proof.evals_proofs {
            w.extend(leaf.iter().map(|x| x.to_canonical_u64()));
            for h in &path.siblings {
                put_hash(&mut w, h);
            }
        }
        for s in &q.steps {
            put_exts::<F, D>(&mut w, &s.evals);
            for h in &s.merkle_proof.siblings {
                put_hash(&mut w, h);
            }
        }
    }
    put_exts::<F, D>(&mut w, &fri.final_poly.coeffs);
    w.push(fri.pow_witness.to_canonical_u64());
    w.extend(p.public_inputs.iter().map(|x| x.to_canonical_u64()));
    Ok(w)
}
