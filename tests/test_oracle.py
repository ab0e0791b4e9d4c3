"""CPU tests of the oracle (no GPU): the literal Python restatement and the plain-C restatement of
the reference's hot path against (1) the bundled fixture's structural invariants, (2) the
hand-derived known answers of SURVEY 8c, (3) each other on seeded random inputs, and (4) the
closed forms the HIP kernels rely on (Occ form of se_down/se_up; last-row-only abort test)."""
import hashlib
import random

import numpy as np
import pytest

from conftest import FM_PATH, REF_PATH, synth_reads
from oracle import erl_literal as E
from oracle import etf, orc


def test_fixture_sha256_and_invariants(fm_dict, ref_bytes):
    assert hashlib.sha256(open(FM_PATH, "rb").read()).hexdigest() == \
        "9a403b6340c1229ad33a2b632c541958566c8849a997210c9fc89268207425b9"
    assert hashlib.sha256(ref_bytes).hexdigest() == \
        "8c2f429d6229640b62c04f80cb50eb997b1e8c77f0dcd4867d918af61adc38b0"
    d = fm_dict
    assert d["pointers"] == (53511, 94127, 134735, 189790) and d["shift"] == 1
    assert (chr(d["F"][0]), chr(d["L"][0]), d["I"][0], d["SA"][0]) == ("$", "C", 53510, 189789)
    assert (chr(d["F"][1]), chr(d["L"][1]), d["I"][1], d["SA"][1]) == ("A", "C", 53511, 184841)
    assert (chr(d["F"][2]), chr(d["L"][2]), d["I"][2], d["SA"][2]) == ("A", "A", 1, 184842)
    assert (chr(d["F"][-1]), chr(d["L"][-1]), d["I"][-1], d["SA"][-1]) == ("T", "C", 94125, 52727)
    T = np.frombuffer(ref_bytes + b"$", np.uint8)
    assert (np.sort(d["SA"]) == np.arange(len(T))).all()
    assert (d["F"] == T[d["SA"]]).all() and (d["L"] == T[d["SA"] - 1]).all()
    assert (np.diff(d["F"].astype(int)) >= 0).all()
    # SA is the true suffix array of ref ++ "$" ($ smallest): adjacent suffixes are ordered
    rng = np.random.default_rng(0)
    t = T.tobytes()
    for r in rng.integers(0, len(T) - 1, 600):
        a, b = int(d["SA"][r]), int(d["SA"][r + 1])
        assert t[a:] < t[b:]                        # '$' (0x24) < 'A': plain byte order; repeats run >400 bp
    Pc, Pg, Pt, Last = d["pointers"]
    c0 = {ord("$"): 0, ord("A"): 1, ord("C"): Pc - 1, ord("G"): Pg - 1, ord("T"): Pt - 1}
    for c, base in c0.items():
        m = d["L"] == c
        assert (d["I"][m] == base + np.cumsum(m)[m] - 1).all()


def test_etf_roundtrip(tmp_path, fm_dict):
    p = tmp_path / "x.fm"
    etf.write_fm(str(p), fm_dict["pointers"], fm_dict["shift"], fm_dict["F"], fm_dict["L"], fm_dict["I"], fm_dict["SA"])
    assert open(p, "rb").read() == open(FM_PATH, "rb").read()


SW_KATS = [  # derived by hand from src/sw.erl
    ("ACGT", "ACGT", (8, "4M")), ("ACGT", "TTACGTTT", (8, "4M")),
    ("ACGTACGT", "ACGTTACGT", (14, "3M1D5M")), ("ACGTTACGT", "ACGTACGT", (14, "3M1I5M")),
    ("ACGTACGTAC", "ACGAACGTAC", (17, "10M")), ("AAAA", "CCCC", "no_match"),
    ("ACGTNACGT", "ACGTAACGT", (17, "9M")), ("ACGTACGTACGT", "ACGTACCCGTACGT", (21, "5M2D7M")),
    ("ACGTACCCGTACGT", "ACGTACGTACGT", (21, "5M2I7M")),
]


@pytest.mark.parametrize("q,r,exp", SW_KATS)
def test_sw_known_answers(q, r, exp):
    assert E.sw(q.encode(), r.encode()) == exp
    assert orc.sw(q.encode(), r.encode()) == exp


def test_sigma_table():
    s = lambda a, b: E.sigma(ord(a), ord(b))
    assert s("N", "N") == 1.6 and s("A", "N") == 1.6 and s("N", "A") == 1.6 and s("A", "A") == 2
    assert s("A", "R") == 2 and s("G", "R") == 2 and s("C", "R") == -1
    assert s("R", "A") == -1 and s("R", "R") == 2          # IUPAC only on the reference side
    assert s("A", "B") == -1 and s("C", "B") == 2 and s("C", "D") == -1 and s("T", "V") == -1 and s("A", "V") == 2
    assert s("A", "H") == -1                                # no H clause (src/sw.erl:202)


def _rand_pair(rng, n_ok=True):
    alpha = b"ACGT"
    lq = rng.randrange(1, 60)
    base = bytes(rng.choice(alpha) for _ in range(lq + 30))
    q = bytearray(base[8:8 + lq])
    r = bytearray(base[:lq + rng.randrange(0, 25)])
    for _ in range(rng.randrange(0, 4)):
        q[rng.randrange(lq)] = rng.choice(alpha)
    if lq > 12 and rng.random() < 0.5:
        p = rng.randrange(3, lq - 3); k = rng.randrange(1, 4)
        if rng.random() < 0.5:
            del q[p:p + k]
        else:
            q[p:p] = bytes(rng.choice(alpha) for _ in range(k))
    if n_ok and rng.random() < 0.4:
        q[rng.randrange(len(q))] = ord("N")
    if n_ok and rng.random() < 0.4:
        r[rng.randrange(len(r))] = rng.choice(b"NRYKMSWBDVH")
    return bytes(q), bytes(r)


def test_sw_c_equals_literal():
    rng = random.Random(7)
    nm = 0
    for _ in range(600):
        q, r = _rand_pair(rng)
        a, b = E.sw(q, r), orc.sw(q, r)
        assert a == b, (q, r)
        nm += a == "no_match"
    assert 0 < nm < 600


def test_sw_abort_only_depends_on_last_row():
    """The kernels test sw.erl:69 only after the last row: RowMax_{i+1} <= RowMax_i + 2 makes the
    per-row test monotone.  Check on the literal table builder with the per-row abort disabled."""
    rng = random.Random(8)
    for _ in range(300):
        q, r = _rand_pair(rng)
        lit = E.sw(q, r)
        # full table without early abort
        lq = len(q)
        rows = [[(0, "undef")] * (len(r) + 1)]
        for i in range(1, lq + 1):
            prev = rows[-1]
            cur = [(prev[0][0] + (E.GAP_PENALTY if i == 1 else E.GAP_EXT_PENALTY), "up")]
            for j in range(1, len(r) + 1):
                cur.append(max([(prev[j - 1][0] + E.sigma(q[i - 1], r[j - 1]), "up_left"),
                                (E._get_Lvalue(*cur[j - 1]), "left"), (E._get_Uvalue(*prev[j]), "up")]))
            rows.append(cur)
        rowmax = [max(v for v, _ in row) for row in rows]
        for i in range(1, lq):
            assert rowmax[i + 1] <= rowmax[i] + 2
        thr = -2 * (lq // 2) + 2 * lq
        assert (lit == "no_match") == (rowmax[lq] < thr)


def test_skip_Ns_cases():
    f = lambda s: E.skip_Ns(s.encode())
    assert f("ACGTACG") is None                      # shorter than MIN_LEN
    assert f("ACGTACGT") == (0, b"ACGTACGT")
    assert f("ACGTACGTNACG") == (3, b"ACGTACGTN")    # the N counts as one of the 8: sub-read ends in N
    assert f("NNNNNNNNNNNN") is None
    assert f("ACGTACGTACNACGTACGT") == (0, b"ACGTACGTACNACGTACGT")
    assert f("AACCGGTTNACGTAC") == (6, b"AACCGGTTN")  # N + the 7 bases left of it make the 8
    assert f("ACCGGTTNACGTAC") == (6, b"ACCGGTTN")
    assert f("CCGGTTNACGTAC") is None                # only 6 bases left of the N: list exhausted at J=7


def test_find_seeds_and_sga_c_equals_literal(lit_fm, orc_fm, ref_bytes):
    reads = (synth_reads(ref_bytes, 60, 100, 3) + synth_reads(ref_bytes, 30, 100, 4, n_frac=0.7) +
             synth_reads(ref_bytes, 20, 150, 5, indel_frac=0.5) + [b"ACGT" * 5, b"N" * 30, b"ACGTACGTACGTNNACGTACGTAC"])
    nseed = 0
    for q in reads:
        # every sub-read of the chain
        qq = q
        while True:
            r = E.skip_Ns(qq)
            if r is None:
                break
            _, q1 = r
            a = E.find_seeds(lit_fm, q1)
            st, seeds, sp, ep, n = orc.find_seeds(orc_fm, q1)
            if a == E.NO_SEEDS:
                assert st == 1
            elif a == E.TOO_MANY:
                assert st == 2
            else:
                assert st == 0 and seeds == a
                nseed += len(a)
            qq = q1[:len(q1) - E.MIN_LEN]
        assert E.sga(lit_fm, q) == orc.sga(orc_fm, q)
    assert nseed > 100


def test_occ_closed_form_of_lf_step(fm_dict):
    """found <=> occ(c,Ep-1) > occ(c,Sp-1); Sp' = C0[c]+occ(c,Sp-1); Ep' = C0[c]+occ(c,Ep)-1."""
    d = fm_dict
    fm = E.FM(d)
    Pc, Pg, Pt, Last = d["pointers"]
    c0 = {ord("A"): 1, ord("C"): Pc - 1, ord("G"): Pg - 1, ord("T"): Pt - 1}
    occ = {c: np.concatenate([[0], np.cumsum(d["L"] == c)]) for c in c0}
    rng = random.Random(9)
    for _ in range(3000):
        sp = rng.randrange(1, Last + 1)
        ep = min(Last, sp + rng.choice([0, 0, 1, 2, 5, 40, 300, 5000]))
        c = rng.choice(list(c0))
        s1 = E.se_down(sp, ep, c, fm)
        found = occ[c][ep - 1] > occ[c][sp - 1]
        assert (s1 is not None) == bool(found)
        if s1 is not None:
            assert s1 == c0[c] + occ[c][sp - 1]
            assert E.se_up(ep, c, fm) == c0[c] + occ[c][ep] - 1


def test_first_step_kat(lit_fm, fm_dict):
    """sub-read ...AC: range of C is [Pc, Pg-1]; extending by A gives
    Sp' = 1 + occ(A, Pc-1), Ep' = 1 + occ(A, Pg-1) - 1."""
    d = fm_dict
    Pc, Pg, Pt, Last = d["pointers"]
    occA = np.concatenate([[0], np.cumsum(d["L"] == ord("A"))])
    assert E.se_down(Pc, Pg - 1, ord("A"), lit_fm) == 1 + occA[Pc - 1]
    assert E.se_up(Pg - 1, ord("A"), lit_fm) == 1 + occA[Pg - 1] - 1


def test_worker_fold_c_equals_literal(lit_fm, orc_fm, ref_bytes):
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 40, 100, 12) + [ref_bytes[:100], ref_bytes[-100:]]
    nh = 0
    for q in reads:
        a = E.worker_candidates(lit_fm, ref1, q)
        b = orc.align_read(orc_fm, ref1, q)
        if a is None:
            assert b is None
            continue
        assert [(s, c, p, w) for ((s, c), p, w) in a] == [(s, c, p, ref1[wo:wo + wl]) for (s, c, p, wo, wl) in b]
        nh += len(a)
        top = E.worker_pick(a)
        if a:
            assert top[0][0] == max(x[0][0] for x in a)
    assert nh > 10


def test_bwt_erl_test_queries_run(lit_fm, orc_fm):
    """The 33 query strings of bwt:test/0 (src/bwt.erl:24-57) are real inputs (one holds N runs);
    a sample of them through both restatements, against the bundled GL000193.1 index."""
    qs = [b"ACCCCACGTTTTTGGAGTTATATGTTGGCACTGATACTGGCCATAGAATTCCCTATGGTA",
          b"ACCNNNNCCACGTTTTTGGAGTTATATGTTGATACTGGCCATAGAATTCCCTATNGGTA",
          b"TAATCAGAACAGGTTTACAACATAAATAAATAGATTGAACTTACTTTGTATAAAAATTGT",
          b"CAGCGGTGCGATCTCGGCTCACTGCAAGCTCTGTCTCCTGGGTTCACGCCATTCTCCTGC"]
    for q in qs:
        assert E.sga(lit_fm, q) == orc.sga(orc_fm, q)


def test_digest_of_hit_tables_matches_the_c_oracle(orc_fm, ref_bytes):
    """orc.hits_digest (numpy, over a hit table in the C ABI's layout) and orc_batch_digest (C, straight from align_read) are
    the same function of the answer: built here from the oracle's own per-read results."""
    import numpy as np
    from conftest import synth_reads
    from oracle import orc, erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 300, 100, 5, indel_frac=0.3, n_frac=0.1) + [b"ACGTACGX" * 13]
    status, hoff, score, pos, woff, wlen, coff, text = [], [0], [], [], [], [], [0], b""
    for q in reads:
        try:
            r = orc.align_read(orc_fm, ref1, q)
        except RuntimeError:
            status.append(-1); hoff.append(hoff[-1]); continue
        status.append(1 if r is None else 0)
        for (s, c, p, wo, wl) in (r or []):
            score.append(s); pos.append(p); woff.append(wo); wlen.append(wl); text += c.encode(); coff.append(len(text))
        hoff.append(len(score))
    out = dict(read_status=np.array(status, np.int32), hit_off=np.array(hoff, np.uint64), score=np.array(score, np.int32),
               pos=np.array(pos, np.int64), win_off=np.array(woff, np.int64), win_len=np.array(wlen, np.int32),
               cig_off=np.array(coff, np.uint64), cigar=text)
    bases = np.frombuffer(b"".join(reads), np.uint8)
    off = np.zeros(len(reads) + 1, np.uint32); off[1:] = np.cumsum([len(r) for r in reads])
    a = orc.hits_digest(out)
    b = orc.batch_digest(orc_fm, ref1, bases, off, 4)
    assert np.array_equal(a, b) and (a > 1).sum() > 100 and (a == 1).sum() == 1
