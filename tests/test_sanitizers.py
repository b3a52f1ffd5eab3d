"""AddressSanitizer + UBSan runs of the CPU side: the oracle's whole prove / verify path and the product's
host code (host witness generators, verifier, host constraint evaluator) on the reference's 512-row test tables
(ModularStark, modular.rs:539-569; G1Stark, g1/muladd.rs:626-680).  Sanitizers on the GPU are not available on this pool, so
device code is covered by the parity tests instead.  The two drivers (tests/sanitize/) are built by `make san` in oracle/
and starky_bn254_amd/csrc/ (__graft_entry__.build() does it; built here on first use otherwise)."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SAN_ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0:halt_on_error=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1",
               OMP_NUM_THREADS="4", SBN_HOST_THREADS="4")


def _build(directory):
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, directory), "san"])


@pytest.fixture(scope="module")
def san_dir(O, tmp_path_factory):
    d = tmp_path_factory.mktemp("san")
    ops, _ = O.modular_inputs(512, 6)
    pts, _ = O.g1op_inputs(512, 0)
    np.ascontiguousarray(ops, dtype=np.uint32).tofile(os.path.join(d, "modular_ops.bin"))
    np.ascontiguousarray(pts, dtype=np.uint32).tofile(os.path.join(d, "g1op_pts.bin"))
    return str(d)


def _run(exe, san_dir):
    r = subprocess.run([exe, san_dir], env=SAN_ENV, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    out, err = r.stdout.decode(errors="replace"), r.stderr.decode(errors="replace")
    assert r.returncode == 0, (out[-2000:], err[-4000:])
    assert "Sanitizer" not in err and "runtime error" not in err, err[-4000:]
    return out


def test_oracle_under_asan_ubsan(O, golden, san_dir):
    _build("oracle")
    out = _run(os.path.join(ROOT, "oracle", "_build", "san_oracle"), san_dir)
    assert "modular:" in out and "g1op:" in out
    # the instrumented build produces the committed proofs
    import hashlib
    for name, key in (("modular", "modular_rows512_seed6"), ("g1op", "g1op_rows512_seed0")):
        w = np.fromfile(os.path.join(san_dir, name + "_proof.bin"), dtype=np.uint64)
        assert hashlib.sha256(w.astype("<u8").tobytes()).hexdigest() == golden["proof_digests"][key]["proof_sha256"]


def test_product_host_code_under_asan_ubsan(O, san_dir):
    if not os.path.exists(os.path.join(san_dir, "g1op_proof.bin")):
        _build("oracle")
        _run(os.path.join(ROOT, "oracle", "_build", "san_oracle"), san_dir)
    _build(os.path.join("starky_bn254_amd", "csrc"))
    out = _run(os.path.join(ROOT, "starky_bn254_amd", "csrc", "_san", "san_host"), san_dir)
    assert out.count("host witness == oracle") == 2
