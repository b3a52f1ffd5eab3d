import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def O():
    """The CPU oracle (test infrastructure)."""
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def S():
    """The product package; builds libsbn254.so if it is missing (hipcc cross-compiles without a GPU)."""
    import starky_bn254_amd as pkg
    if not os.path.exists(pkg.lib_path()):
        import subprocess
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "starky_bn254_amd", "csrc")])
    pkg.lib()
    return pkg


@pytest.fixture(scope="session")
def golden():
    d = os.path.join(ROOT, "tests", "golden")
    return {f[:-5]: json.load(open(os.path.join(d, f))) for f in os.listdir(d) if f.endswith(".json")}


@pytest.fixture(scope="session")
def g1op_case(O):
    """Seeded G1Stark trace (512 rows) + the oracle's proof of it."""
    pts, native = O.g1op_inputs(512, 0)
    trace = O.g1op_trace(pts)
    words, secs = O.prove(O.AIR_G1_OP, 0, trace, np.zeros(0, dtype=np.uint64))
    return {"pts": pts, "native": native, "trace": trace, "proof": words, "pi": np.zeros(0, dtype=np.uint64)}


@pytest.fixture(scope="session")
def g1exp_case(O):
    """Seeded G1ExpStark(128) trace (2^16 rows); ~1 GB, a few seconds."""
    ios, native = O.g1exp_inputs(128, 1)
    trace, pi = O.g1exp_trace(ios)
    return {"ios": ios, "native": native, "trace": trace, "pi": pi}


@pytest.fixture(scope="session")
def g2exp_case(O):
    """Seeded G2ExpStark(128) trace (2^16 rows x 2822 columns); ~1.5 GB."""
    ios, native = O.g2exp_inputs(128, 2)
    trace, pi = O.g2exp_trace(ios)
    return {"ios": ios, "native": native, "trace": trace, "pi": pi}


@pytest.fixture(scope="session")
def fq12exp_case(O):
    """Seeded Fq12ExpStark(16) trace (2^13 rows x 9802 columns), the size of the reference's test_fq12_exp_raw."""
    ios, native = O.fq12exp_inputs(16, 3)
    trace, pi = O.fq12exp_trace(ios)
    return {"ios": ios, "native": native, "trace": trace, "pi": pi}


@pytest.fixture(scope="session")
def fqexp_case(O):
    """Seeded FqExpStark(128) trace (2^16 rows x 960 columns), the size of the reference's test_fq_exp_raw."""
    ios, native = O.fqexp_inputs(128, 4)
    trace, pi = O.fqexp_trace(ios)
    return {"ios": ios, "native": native, "trace": trace, "pi": pi}


@pytest.fixture(scope="session")
def fq12expu64_case(O):
    """Seeded Fq12ExpU64Stark(16) trace (2^11 rows x 9792 columns), the size of the reference's test_fq12_exp_u64_raw."""
    ios, native = O.fq12expu64_inputs(16, 5)
    trace, pi = O.fq12expu64_trace(ios)
    return {"ios": ios, "native": native, "trace": trace, "pi": pi}

