"""FASTQ ingest and result sink (8f-3): the native parser / host mirror against the literal
restatement of fastq.erl and master.erl -- record boundaries, droplast quirk, reverse batch order,
workload production, the 265 filter and the printed line.  CPU only (no compute calls)."""
import io

import pytest

from oracle import erl_literal as E


def make_fastq(n, with_junk=True, final_newline=True, seed=0):
    import random
    rng = random.Random(seed)
    out = []
    if with_junk:
        out.append("junk line before the first record\n")
    for i in range(n):
        seq = "".join(rng.choice("ACGTN") for _ in range(rng.randrange(20, 120)))
        qual = "".join(rng.choice("@IJ#+") for _ in seq)      # quality lines may start with '@'
        out.append("@read.%d extra words\n%s\n+\n%s\n" % (i, seq, qual))
    text = "".join(out)
    if not final_newline:
        text = text[:-1]
    return text.encode()


@pytest.mark.parametrize("n,final_newline", [(0, True), (1, True), (7, True), (450, True), (5, False)])
def test_read_seqs_matches_literal(tmp_path, n, final_newline):
    from bwt_amd import fastq
    data = make_fastq(n, final_newline=final_newline, seed=n)
    p = tmp_path / "x.fastq"
    p.write_bytes(data)
    dev = fastq.Dev(str(p))
    lit = E.LineDev(data)
    for want in (200, 200, 3, 200, 1):
        assert fastq.read_seqs(dev, want) == E.fastq_read_seqs(lit, want)
    dev.close()


def test_header_without_sequence_is_reported(tmp_path):
    from bwt_amd import fastq, lib
    p = tmp_path / "bad.fastq"
    p.write_bytes(b"@only a header\n")
    dev = fastq.Dev(str(p))
    with pytest.raises(lib.BwtError):          # {ok,SeqData} = eof: the reference badmatches here
        fastq.read_seqs(dev, 10)
    dev.close()


def test_produce_workload_matches_literal(tmp_path):
    from bwt_amd import fastq, master
    data = make_fastq(1234, seed=5)
    p = tmp_path / "y.fastq"
    p.write_bytes(data)
    s = master.State(fastq.Dev(str(p)), "GL000193.1")
    lit, lst = E.LineDev(data), dict(fastq_eof=False, stopping=False)
    total = 0
    for _ in range(5):
        a = master.get_workload(3, s)
        b = E.master_produce_workload(3, lit, lst)
        assert a == b
        if a == "stop":
            break
        total += sum(len(x) for x in a)
    assert total == 1234 == s.stat_workload_amount and s.stopping and lst["stopping"]
    s.fastq.close()


def test_cigar_sink_filter_and_format():
    from bwt_amd import master
    cast = ("cigar", "r1", (266, "150M"), 117832, b"ACGT")
    assert master.cigar_line(cast, "GL000193.1") == E.master_cigar_line("r1", "GL000193.1", 117832, "150M", 266, "ACGT")
    assert master.cigar_line(("cigar", "r1", (200, "100M"), 5, b"AC"), "21") is None      # master.erl:149: 100 bp reads never pass
    assert master.cigar_line(("cigar", "r1", (200, "100M"), 5, b"AC"), "21", threshold=100) == \
        "\nr1      21      5      100M      200      AC\n"


@pytest.mark.parametrize("kind", ["clean", "no_final_newline", "junk_first", "junk_inside", "partial_tail", "tiny"])
@pytest.mark.parametrize("threads", [2, 5, 16])
def test_threaded_reader_equals_sequential_and_literal(tmp_path, kind, threads):
    """The threaded reader (four-line records, '@' headers; quality lines that start with '@' included) against the sequential
    reader and the literal restatement, batch by batch, on clean files and on every kind of file it must hand back to the
    sequential rules: a junk line first or in the middle (the reference skips lines until an '@'), a last record cut short,
    a file without final newline."""
    from bwt_amd import fastq
    n = 30 if kind == "tiny" else 4000
    data = make_fastq(n, with_junk=(kind == "junk_first"), final_newline=(kind != "no_final_newline"), seed=17 + threads)
    if kind == "junk_inside":
        cut = data.index(b"@read.1500 ")
        data = data[:cut] + b"not a header\n" + data[cut:]
    if kind == "partial_tail":
        data = data + b"@tail record\nACGTACGT\n"                      # header + sequence, no delimiter / quality: still a record
    p = tmp_path / "t.fastq"
    p.write_bytes(data)
    par, seq = fastq.Dev(str(p)), fastq.Dev(str(p))
    par.set_threads(threads, min_records=1)
    seq.set_threads(1)
    lit = E.LineDev(data)
    got_any = False
    for want in (700, 1, 1299, 64, 5000, 10):
        a = fastq.read_seqs(par, want)
        b = fastq.read_seqs(seq, want)
        assert a == b == E.fastq_read_seqs(lit, want), (kind, want)
        got_any = got_any or (a != "eof" and len(a[1]) > 0)
    assert got_any
    par.close(); seq.close()


def test_native_pick_equals_otp_sort_head():
    """bwt_otp_pick (C++ restatement of lists:sort/2 in the library) against the Python twin: the head of the sorted candidate
    list for the comparator of worker_bwt.erl:104, ties included."""
    import random
    from bwt_amd import lib, otp_lists
    rng = random.Random(11)
    for _ in range(5000):
        n = rng.randrange(1, 16)
        sc = [rng.randrange(0, rng.choice([1, 2, 3, 6, 100])) for _ in range(n)]
        exp = otp_lists.sort(lambda a, b: sc[a] > sc[b], list(range(n)))[0]
        assert lib.otp_pick(sc) == exp, sc
