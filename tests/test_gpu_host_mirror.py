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


def test_master_run_end_to_end(tmp_path, lit_fm, ref_bytes):
    """FASTQ file -> workloads -> GPU worker -> result sink, against the literal restatement of
    fastq.erl + worker_bwt.erl + master.erl (150 bp reads: scores above the 265 filter exist)."""
    import io
    from bwt_amd import fm_index, master
    from oracle import erl_literal as E
    reads = synth_reads(ref_bytes, 450, 150, 91, max_subs=1)
    text = "".join("@synth.%d\n%s\n+\n%s\n" % (i, q.decode(), "I" * len(q)) for i, q in enumerate(reads)).encode()
    p = tmp_path / "reads.fastq"
    p.write_bytes(text)
    fm_index.set_env("fm_indices", GOLDEN + "/fm_indices")
    try:
        out = io.StringIO()
        n, lines, _ = master.run(str(p), "GL000193.1", out=out)
    finally:
        fm_index._env.clear()
    assert n == 450
    # literal pipeline
    ref1 = E.make_ref1(ref_bytes)
    lit, st = E.LineDev(text), dict(fastq_eof=False, stopping=False)
    exp = []
    while True:
        wl = E.master_produce_workload(10, lit, st)
        if wl == "stop":
            break
        for qlist in wl:
            for name, q in reversed(qlist):                      # worker answers in reverse batch order
                cand = E.worker_candidates(lit_fm, ref1, q)
                if cand is None:
                    continue
                top = E.worker_pick(cand)
                if top == "no_cigar":
                    continue
                line = E.master_cigar_line(name.decode(), "GL000193.1", top[1], top[0][1], top[0][0], top[2].decode())
                if line is not None:
                    exp.append(line)
    got = out.getvalue().split("\n\nReads:")[0]
    assert got == "".join(exp)
    assert lines == len(exp) > 20


def test_reverse_complement(ref_bytes):
    """fastq:reverse_complement/1 (src/fastq.erl:43-50): the GPU batch kernel against the literal oracle, ragged lengths,
    empty reads, N, and the inputs the reference has no clause for."""
    import numpy as np
    import pytest
    from bwt_amd import lib, fastq
    from oracle import erl_literal as E
    rng = np.random.default_rng(5)
    reads = [b"", b"A", b"ACGTN", ref_bytes[:1], ref_bytes[100:1100]] + \
            [bytes(rng.choice(list(b"ACGTN"), int(rng.integers(0, 300))).tolist()) for _ in range(200)] + [b"ACGR", b"acgt"]
    out, st = lib.reverse_complement_batch(*lib.pack_reads(reads))
    o = 0
    for i, r in enumerate(reads):
        try:
            exp = E.fastq_reverse_complement(r)
            assert st[i] == 0 and out[o:o + len(r)].tobytes() == exp, r
        except E.ErlCrash:
            assert st[i] == -1
        o += len(r)
    assert fastq.reverse_complement(b"AACGTN") == b"NACGTT"
    with pytest.raises(ValueError):
        fastq.reverse_complement(b"ACGX")


def test_native_runner_prints_what_the_mirror_prints(tmp_path, ref_bytes):
    """bwt_run_fastq (threaded reader -> pipe -> pick -> sink, all native) against master.run (the per-QseqList Python mirror):
    the same result lines in the same order on a FASTQ that spans several workloads and batches, with repeats (several
    alignments per read, equal scores), indels and N; and the same counts."""
    import io
    import numpy as np
    from bwt_amd import fm_index, master
    from conftest import GOLDEN, synth_reads
    rng = np.random.default_rng(3)
    reads = synth_reads(ref_bytes, 5300, 100, 901, indel_frac=0.15, n_frac=0.05) + [ref_bytes[i:i + 100] for i in range(0, 3000, 37)]
    fq = tmp_path / "r.fastq"
    with open(fq, "wb") as f:
        for i, r in enumerate(reads):
            f.write(b"@read.%d some words\n%s\n+\n%s\n" % (i, r, b"I" * len(r)))
    fm_index.set_env("fm_indices", GOLDEN + "/fm_indices")
    buf = io.StringIO()
    n_reads, n_lines, _ = master.run(str(fq), "GL000193.1", out=buf, threshold=100)
    text = buf.getvalue()
    want = text[: text.index("\n\nReads: ")]
    out = tmp_path / "native.txt"
    st = master.run_native(str(fq), "GL000193.1", str(out), threshold=100, batch_reads=4000)
    got = open(out, "r", encoding="latin1").read()
    assert st["reads"] == n_reads == len(reads) and st["lines"] == n_lines > 2000
    assert got == want
    st2 = master.run_native(str(fq), "GL000193.1", None, threshold=265)
    assert st2["lines"] == 0 and st2["cigars"] == st["cigars"] and st2["no_cigar"] == st["no_cigar"]
