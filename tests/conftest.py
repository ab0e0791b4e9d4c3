import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")
FM_PATH = os.path.join(GOLDEN, "fm_indices", "GL000193.1_p1.fm")
REF_PATH = os.path.join(GOLDEN, "fm_indices", "GL000193.1_p1.ref")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def fm_dict():
    from oracle import etf
    return etf.read_fm(FM_PATH)


@pytest.fixture(scope="session")
def ref_bytes():
    with open(REF_PATH, "rb") as f:
        return f.read()


@pytest.fixture(scope="session")
def orc_fm(fm_dict):
    from oracle import orc
    orc.build()
    return orc.OrcFM.from_etf(fm_dict)


@pytest.fixture(scope="session")
def lit_fm(fm_dict):
    from oracle import erl_literal
    return erl_literal.FM(fm_dict)


@pytest.fixture(scope="session")
def gpu_index():
    from bwt_amd import lib
    ix = lib.Index.open(FM_PATH, REF_PATH, 0)
    yield ix
    ix.close()


def synth_reads(ref, n, length, seed, max_subs=2, n_frac=0.0, iupac_frac=0.0, indel_frac=0.0):
    """Seeded synthetic reads drawn from `ref`: uniform start, 0..max_subs substitutions,
    optional N / IUPAC bases and one indel of 1-5 bp."""
    rng = np.random.default_rng(seed)
    out = []
    L = len(ref)
    for _ in range(n):
        st = int(rng.integers(0, L - length - 8))
        q = bytearray(ref[st:st + length + 8])
        if indel_frac and rng.random() < indel_frac:
            k = int(rng.integers(1, 6)); p = int(rng.integers(10, length - 10))
            if rng.random() < 0.5:
                del q[p:p + k]
            else:
                q[p:p] = bytes(rng.choice(list(b"ACGT"), k).tolist())
        q = q[:length]
        for _ in range(int(rng.integers(0, max_subs + 1))):
            q[int(rng.integers(0, length))] = int(rng.choice(list(b"ACGT")))
        if n_frac and rng.random() < n_frac:
            for _ in range(int(rng.integers(1, 4))):
                q[int(rng.integers(0, length))] = ord("N")
        if iupac_frac and rng.random() < iupac_frac:
            q[int(rng.integers(0, length))] = int(rng.choice(list(b"RYKMSWBDV")))
        out.append(bytes(q))
    return out
