"""The reference-parity hand-over kit: per-STAGE digests of a proof.

The reference (qope/starky-bn254) holds no golden vector and cannot be built in this image, so proof bytes are pinned
GPU == oracle only.  What a maintainer with a Rust toolchain needs in order to pin them against the real starky prover is
(i) the exact inputs of every golden case, (ii) the value every stage of prove() must produce for them, in an order that
names the FIRST stage that differs, and (iii) the program that prints the same values from the reference's own
`generate_trace` + `prove` (integration/rust/starky-bn254-amd/tests/parity.rs).  This module is (ii): it cuts canonical
proof words (include/sbn.h:20-31) into their sections and replays the Fiat-Shamir transcript of starky 0.1.1
`prover::prove` (trace cap -> permutation challenges -> Z cap -> alphas -> quotient cap -> zeta -> openings -> FRI alpha ->
per layer cap / beta -> final polynomial -> proof of work -> query indices) with the plain-definition Poseidon permutation,
so every challenge is listed next to the commitment that determines it.

Test infrastructure: used by tests/ (CPU: oracle proofs against the committed kit; GPU: device proofs against it) and by
tests/golden/make_parity_kit.py.  Nothing of the product imports it.
"""
import hashlib

import numpy as np

GL_P = 0xFFFFFFFF00000001
STAGES = ["trace_cap", "permutation_challenges", "permutation_zs_cap", "alphas", "quotient_polys_cap", "zeta", "openings", "fri_alpha",
          "fri_commit_caps", "fri_betas", "final_poly", "pow_witness", "pow_response", "query_indices", "query_rounds", "public_inputs", "proof"]


def sha(words):
    return hashlib.sha256(np.asarray(words, dtype="<u8").tobytes()).hexdigest()


def parse_proof(words):
    """Canonical proof words -> dict of sections (numpy views), following include/sbn.h:20-31."""
    w = np.asarray(words, dtype=np.uint64)
    h = [int(x) for x in w[:12]]
    assert h[0] == int.from_bytes(b"SNBPROV1", "little"), "not a proof of this library"
    keys = ["magic", "degree_bits", "n_trace", "n_zs", "n_quot", "n_pi", "cap_height", "rate_bits", "n_layers", "arity_bits", "final_len", "n_queries"]
    hd = dict(zip(keys, h))
    pos = 12
    out = {"header": hd}

    def take(n):
        nonlocal pos
        s = w[pos:pos + n]
        assert len(s) == n, "proof words truncated"
        pos += n
        return s
    capw = 4 << hd["cap_height"]
    out["trace_cap"] = take(capw)
    out["permutation_zs_cap"] = take(capw) if hd["n_zs"] else w[:0]
    out["quotient_polys_cap"] = take(capw)
    o0 = pos
    out["local_values"] = take(2 * hd["n_trace"])
    out["next_values"] = take(2 * hd["n_trace"])
    out["permutation_zs"] = take(2 * hd["n_zs"])
    out["permutation_zs_next"] = take(2 * hd["n_zs"])
    out["quotient_polys"] = take(2 * hd["n_quot"])
    out["openings"] = w[o0:pos]
    out["fri_commit_caps"] = [take(capw) for _ in range(hd["n_layers"])]
    q0 = pos
    lde_bits = hd["degree_bits"] + hd["rate_bits"]
    per_query = 0
    for width in [hd["n_trace"]] + ([hd["n_zs"]] if hd["n_zs"] else []) + [hd["n_quot"]]:
        per_query += width + 4 * (lde_bits - hd["cap_height"])
    bits = lde_bits
    for _ in range(hd["n_layers"]):
        bits -= hd["arity_bits"]
        per_query += 2 * (1 << hd["arity_bits"]) + 4 * max(bits - hd["cap_height"], 0)
    out["query_rounds"] = take(per_query * hd["n_queries"])
    out["query_stride"] = per_query
    assert pos - q0 == per_query * hd["n_queries"]
    out["final_poly"] = take(2 * hd["final_len"])
    out["pow_witness"] = int(take(1)[0])
    out["public_inputs"] = take(hd["n_pi"])
    assert pos == len(w), "trailing words after the proof"
    return out


class Challenger:
    """plonky2 iop/challenger.rs `Challenger` (duplex sponge in overwrite mode, rate 8, width 12) -- the same restatement as
    starky_bn254_amd/csrc/host_common.hpp and oracle/stark.hpp, in Python over a permutation callable."""

    def __init__(self, permute):
        self.permute = permute
        self.st = [0] * 12
        self.inp, self.out = [], []

    def _duplex(self):
        for i, v in enumerate(self.inp):
            self.st[i] = v
        self.inp = []
        self.st = self.permute(self.st)
        self.out = list(self.st[:8])

    def observe(self, v):
        self.out = []
        self.inp.append(int(v))
        if len(self.inp) == 8:
            self._duplex()

    def observe_words(self, ws):
        for v in ws:
            self.observe(v)

    def challenge(self):
        if self.inp or not self.out:
            self._duplex()
        return self.out.pop()

    def ext_challenge(self):
        a = self.challenge()
        b = self.challenge()
        return [a, b]


def stage_digests(words, permute, num_challenges=2, pow_bits=16):
    """Every stage output of prove() for this proof, in stage order (STAGES).  Small values are listed in full (hex), large
    sections as sha256 of their little-endian words plus their first four words."""
    p = parse_proof(words)
    hd = p["header"]
    ch = Challenger(permute)
    hexl = lambda ws: [hex(int(x)) for x in ws]   # noqa: E731
    big = lambda ws: {"sha256": sha(ws), "words": int(len(ws)), "first": hexl(ws[:4])}   # noqa: E731
    d = {}
    d["trace_cap"] = hexl(p["trace_cap"])
    ch.observe_words(p["trace_cap"])
    # get_n_permutation_challenge_sets(num_challenges, batch size 2): 2 sets x num_challenges x (beta, gamma)
    d["permutation_challenges"] = hexl([ch.challenge() for _ in range(2 * num_challenges * 2)]) if hd["n_zs"] else []
    d["permutation_zs_cap"] = hexl(p["permutation_zs_cap"])
    ch.observe_words(p["permutation_zs_cap"])
    d["alphas"] = hexl([ch.challenge() for _ in range(num_challenges)])
    d["quotient_polys_cap"] = hexl(p["quotient_polys_cap"])
    ch.observe_words(p["quotient_polys_cap"])
    d["zeta"] = hexl(ch.ext_challenge())
    d["openings"] = {k: big(p[k]) for k in ("local_values", "next_values", "permutation_zs", "permutation_zs_next", "quotient_polys")}
    # observe_openings: batch at zeta = local ++ perm_zs ++ quotient; batch at g*zeta = next ++ perm_zs_next
    for k in ("local_values", "permutation_zs", "quotient_polys", "next_values", "permutation_zs_next"):
        ch.observe_words(p[k])
    d["fri_alpha"] = hexl(ch.ext_challenge())
    d["fri_commit_caps"], d["fri_betas"] = [], []
    for cap in p["fri_commit_caps"]:
        d["fri_commit_caps"].append({"sha256": sha(cap), "first": hexl(cap[:4])})
        ch.observe_words(cap)
        d["fri_betas"].append(hexl(ch.ext_challenge()))
    d["final_poly"] = hexl(p["final_poly"])
    ch.observe_words(p["final_poly"])
    d["pow_witness"] = p["pow_witness"]
    ch.observe(p["pow_witness"])
    resp = ch.challenge()
    d["pow_response"] = hex(resp)
    d["pow_ok"] = (64 - resp.bit_length()) >= pow_bits
    m = 1 << (hd["degree_bits"] + hd["rate_bits"])
    d["query_indices"] = [ch.challenge() % m for _ in range(hd["n_queries"])]
    d["query_rounds"] = big(p["query_rounds"])
    d["public_inputs"] = big(p["public_inputs"])
    d["proof"] = {"sha256": sha(words), "words": int(len(words))}
    return d


def first_difference(got, want):
    """Name of the first stage (in prove() order) whose value differs, or None."""
    for k in STAGES:
        if got.get(k) != want.get(k):
            return k
    return None
