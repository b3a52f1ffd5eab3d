"""The C ABI from compiled C code (examples/c): no Python, no torch in the process that proves.  prove_g1exp runs the
reference's test_g1_exp_raw workload (src/curves/g1/exp.rs:784-845) on one GPU; split_rccl is one rank of a split proof over
the library's RCCL transport, the unique id handed around through a file (world 1 on a one-GPU box).  Both must print the
checksum of the proof the Python binding gets for the same inputs."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples", "c")


def _checksum(words):
    s = 0
    for w in np.asarray(words, dtype=np.uint64).tolist():
        s = ((s * 0x100000001B3) & 0xFFFFFFFFFFFFFFFF) ^ w
    return "%016x" % s


def test_c_examples_build_against_the_header():
    """gcc (a C compiler, not hipcc) accepts include/sbn.h and links the examples against libsbn254.so."""
    subprocess.check_call(["make", "-s", "-C", EX, "clean"])
    subprocess.check_call(["make", "-s", "-C", EX])
    assert os.path.exists(os.path.join(EX, "prove_g1exp")) and os.path.exists(os.path.join(EX, "split_rccl"))


@pytest.mark.gpu
def test_c_examples_prove_the_same_proofs(S, O, tmp_path):
    if S.lib().sbn_device_count() < 1:
        pytest.fail("no HIP device")
    subprocess.check_call(["make", "-s", "-C", EX])
    ios, _ = O.g1exp_inputs(128, 1)
    np.ascontiguousarray(ios, dtype="<u4").tofile(tmp_path / "g1.bin")
    out = subprocess.run([os.path.join(EX, "prove_g1exp"), str(tmp_path / "g1.bin")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert "verified" in out, out
    stark = S.G1ExpStark(128)
    p = S.Prover(stark, stark.config(), 16)
    p.generate_trace(ios)
    want = _checksum(p.prove().words)
    p.close()
    assert re.search(r"checksum ([0-9a-f]{16})", out).group(1) == want, out
    ios12, _ = O.fq12exp_inputs(16, 3)
    np.ascontiguousarray(ios12, dtype="<u4").tofile(tmp_path / "fq12.bin")
    out = subprocess.run([os.path.join(EX, "split_rccl"), "0", "1", str(tmp_path / "id"), str(tmp_path / "fq12.bin"), "16"], stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, timeout=300).stdout.decode()
    assert "verified" in out, out
    stark = S.Fq12ExpStark(16)
    p = S.Prover(stark, stark.config(), 13)
    p.generate_trace(ios12)
    want = _checksum(p.prove().words)
    p.close()
    assert re.search(r"checksum ([0-9a-f]{16})", out).group(1) == want, out
