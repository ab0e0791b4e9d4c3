"""GPU tests of the host-side mirror of the reference's interfaces (bwt_amd.bwa / sga / sw /
worker_bwt): same names, same result shapes, same list orders as the Erlang modules, checked
against the literal Python restatement of the Erlang source."""
import pytest

from conftest import GOLDEN, synth_reads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fm():
    from bwt_amd import fm_index
    fm_index.set_env("fm_indices", GOLDEN + "/fm_indices")
    meta, fm = fm_index.get_index("GL000193.1", 1)
    assert meta == [("pointers", (53511, 94127, 134735, 189790)), ("shift", 1)]
    yield fm
    fm.close()
    fm_index._env.clear()


def test_find_seeds_mirror(fm, lit_fm, ref_bytes):
    from bwt_amd import bwa
    from oracle import erl_literal as E
    for q in synth_reads(ref_bytes, 12, 100, 51):
        for cut in (100, 60, 23, 9):
            sub = q[:cut]
            exp = E.find_seeds(lit_fm, sub)
            assert bwa.find_seeds(fm, fm.Pc, fm.Pg, fm.Pt, fm.Last, sub) == exp
    with pytest.raises(bwa.CaseClause):
        bwa.find_seeds(fm, fm.Pc, fm.Pg, fm.Pt, fm.Last, b"ACGTACGTX")


def test_sga_and_sw_mirror(fm, lit_fm, ref_bytes):
    from bwt_amd import sga, sw
    from oracle import erl_literal as E
    for q in synth_reads(ref_bytes, 25, 100, 52, n_frac=0.3):
        assert sga.sga(fm, fm.Pc, fm.Pg, fm.Pt, fm.Last, q) == E.sga(lit_fm, q)
    assert sw.sw(b"ACGTACGT", b"ACGTTACGT") == (14, "3M1D5M")
    assert sw.sw(b"AAAA", b"CCCC") == "no_match"
    assert sw.sw(b"ACGTNACGT", b"ACGTAACGT") == (17, "9M")


def test_worker_running_clause(fm, lit_fm, ref_bytes):
    """One QseqList through worker_bwt.running == the literal fold + pick, in the reference's cast order."""
    from bwt_amd import worker_bwt
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 60, 100, 53)
    qlist = [("synth.%d" % i, q) for i, q in enumerate(reads)]
    casts = worker_bwt.running(fm, qlist)
    exp = []
    for name, q in reversed(qlist):                       # Seeds is built by prepending (worker_bwt.erl:74-80)
        cand = E.worker_candidates(lit_fm, ref1, q)
        if cand is None:
            continue
        top = E.worker_pick(cand)
        exp.append("no_cigar" if top == "no_cigar" else ("cigar", name, top[0], top[1], top[2]))
    assert casts == exp
    assert sum(1 for c in casts if c != "no_cigar") > 10
