"""GPU tests of the reverse-strand and pairing entry points.  The reference has neither, so the checks are:
a both-strand batch gives, for read i and for read n + i, exactly what the oracle gives for the read and for its reverse
complement (every read still goes through the reference's path); the pairing rule is restated here in plain Python over the
fetched hit tables and must agree record by record, and simulated fragments must come back as proper pairs."""
import numpy as np
import pytest

from conftest import synth_reads
from test_gpu_modes import check_against_oracle

pytestmark = pytest.mark.gpu

COMP = bytes.maketrans(b"ACGTRYKMBVDHSWN", b"TGCAYRMKVBHDSWN")


def rc(q):
    return q.translate(COMP)[::-1]


def test_both_strands_batch_is_the_path_on_read_and_reverse_complement(gpu_index, orc_fm, ref_bytes):
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 300, 100, 1201, indel_frac=0.2, iupac_frac=0.1, n_frac=0.05)
    reads = [rc(q) if i % 2 else q for i, q in enumerate(reads)] + synth_reads(ref_bytes, 100, 72, 1202)
    b = lib.Batch(gpu_index)
    try:
        b.set_reads_both_strands(*lib.pack_reads(reads))
        b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        assert len(out["read_status"]) == 2 * len(reads)
        both = reads + [rc(q) for q in reads]
        assert check_against_oracle(out, both, orc_fm, ref1) > 300
        # the reads that were drawn reverse-complemented align on their reverse strand
        n = len(reads)
        # (on this small contig about four reads in ten find their cluster at all)
        rev_hits = sum(int(out["hit_off"][n + i + 1] - out["hit_off"][n + i]) > 0 for i in range(1, 300, 2))
        fwd_hits = sum(int(out["hit_off"][i + 1] - out["hit_off"][i]) > 0 for i in range(0, 300, 2))
        assert rev_hits > 40 and fwd_hits > 40
        wrong = sum(int(out["hit_off"][i + 1] - out["hit_off"][i]) > 0 for i in range(1, 300, 2))
        assert wrong < rev_hits // 4
    finally:
        b.destroy()


def ref_span(cig):
    span = num = 0
    for ch in cig:
        if ch.isdigit():
            num = num * 10 + int(ch)
        else:
            if ch in "MD":
                span += num
            num = 0
    return span


def pair_rule(out, n_fwd, p, lo, hi):
    """include/bwt_gpu.h, bwt_batch_pair_hits, restated."""
    best = None
    n_proper = 0
    for o in (0, 1):
        rf, rr = 2 * p + o, n_fwd + 2 * p + (1 - o)
        for hf in range(int(out["hit_off"][rf]), int(out["hit_off"][rf + 1])):
            for hr in range(int(out["hit_off"][rr]), int(out["hit_off"][rr + 1])):
                pf, pr = int(out["pos"][hf]), int(out["pos"][hr])
                if pr < pf:
                    continue
                cig = out["cigar"][int(out["cig_off"][hr]):int(out["cig_off"][hr + 1])].decode()
                ins = pr + ref_span(cig) - pf
                if ins < lo or ins > hi:
                    continue
                n_proper += 1
                sc = int(out["score"][hf]) + int(out["score"][hr])
                if best is None or sc > best[3]:
                    best = ((hf, hr) if o == 0 else (hr, hf)) + (ins, sc, o)
    return best, n_proper


def test_pairing_of_simulated_fragments(gpu_index, ref_bytes):
    from bwt_amd import lib
    rng = np.random.default_rng(77)
    L = len(ref_bytes)
    mates, truth = [], []
    for p in range(400):
        frag = int(rng.integers(300, 501))
        st = int(rng.integers(0, L - frag))
        f = bytearray(ref_bytes[st:st + frag])
        for _ in range(int(rng.integers(0, 4))):
            f[int(rng.integers(0, frag))] = int(rng.choice(list(b"ACGT")))
        f = bytes(f)
        m1, m2 = f[:100], rc(f[-100:])
        if rng.random() < 0.5:                      # the fragment came from the other strand: mate 1 is the reverse one
            m1, m2 = m2, m1
            truth.append((st, frag, 1))
        else:
            truth.append((st, frag, 0))
        mates += [m1, m2]
    b = lib.Batch(gpu_index)
    try:
        b.set_reads_both_strands(*lib.pack_reads(mates))
        b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        pairs = b.pair_hits(250, 550)
        assert len(pairs) == 400
        found = 0
        for p in range(400):
            best, n_proper = pair_rule(out, len(mates), p, 250, 550)
            rec = pairs[p]
            assert int(rec["n_proper"]) == n_proper, p
            if best is None:
                assert int(rec["hit1"]) == lib.NO_HIT and int(rec["hit2"]) == lib.NO_HIT
                continue
            assert (int(rec["hit1"]), int(rec["hit2"]), int(rec["insert"]), int(rec["score"]), int(rec["flags"]) & 1) == best, p
            st, frag, flipped = truth[p]
            if int(rec["insert"]) == frag and int(rec["flags"]) & 1 == flipped:
                found += 1
        assert found > 40           # fragments whose two mates both align come back as the proper pair they were drawn as
        with pytest.raises(RuntimeError):
            b.pair_hits(500, 100)
    finally:
        b.destroy()


def test_pairing_ties_and_many_hits_against_array_form():
    """Pairs drawn from a genome with a 400-bp segment in 12 exact copies: every mate has ~12 equally good hits, so score sums tie
    and the rule's tie-break (first in enumeration order: mate 1 forward before mate 2 forward, forward hit outer, reverse hit
    inner) decides.  Checked against an ARRAY formulation -- all combinations of a pair at once, np.argmax takes the first
    maximum in C order -- instead of the loop restatement above; the reference span comes from the CIGAR text (k_hit_spans
    computes it once per hit now)."""
    from bwt_amd import lib
    rng = np.random.default_rng(5)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 600_000, dtype=np.uint8)].tobytes())
    seg = bytes(g[10_000:10_400])
    for k in range(1, 12):
        g[40_000 * k + 10_000:40_000 * k + 10_400] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    mates = []
    for p in range(60):
        a = int(rng.integers(0, 60)); frag = int(rng.integers(280, 400 - a))
        f = bytearray(seg[a:a + frag])
        if p % 3 == 0:                                           # one indel in the reverse mate: its CIGAR consumes Lq +- k reference bases
            del f[frag - 50:frag - 50 + 1 + p % 3]
        m1, m2 = bytes(f[:100]), rc(bytes(f[-100:]))
        mates += [m2, m1] if p % 2 else [m1, m2]
    b = lib.Batch(ix)
    try:
        b.set_reads_both_strands(*lib.pack_reads(mates))
        b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        lo, hi = 200, 450
        pairs = b.pair_hits(lo, hi)
        n_fwd = len(mates)
        span = np.array([ref_span(out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode()) for h in range(len(out["score"]))], np.int64)
        pos, score, hoff = out["pos"].astype(np.int64), out["score"].astype(np.int64), out["hit_off"].astype(np.int64)
        tied = many = 0
        for p in range(60):
            cands = []
            for o in (0, 1):
                rf, rr = 2 * p + o, n_fwd + 2 * p + (1 - o)
                hf = np.arange(hoff[rf], hoff[rf + 1]); hr = np.arange(hoff[rr], hoff[rr + 1])
                F, R = np.meshgrid(hf, hr, indexing="ij")
                ins = pos[R] + span[R] - pos[F]
                ok = (pos[R] >= pos[F]) & (ins >= lo) & (ins <= hi)
                cands.append((F[ok], R[ok], ins[ok], (score[F] + score[R])[ok], np.full(int(ok.sum()), o)))
            F, R, ins, sc, oo = (np.concatenate([c[i] for c in cands]) for i in range(5))
            rec = pairs[p]
            assert int(rec["n_proper"]) == len(sc), p
            if not len(sc):
                assert int(rec["hit1"]) == lib.NO_HIT
                continue
            k = int(np.argmax(sc))                                # the first maximum in enumeration order
            tied += int((sc == sc[k]).sum() > 1); many += len(sc) >= 3
            h1, h2 = (F[k], R[k]) if oo[k] == 0 else (R[k], F[k])
            assert (int(rec["hit1"]), int(rec["hit2"]), int(rec["insert"]), int(rec["score"]), int(rec["flags"]) & 1) == \
                (int(h1), int(h2), int(ins[k]), int(sc[k]), int(oo[k])), p
        assert tied > 20 and many > 20
    finally:
        b.destroy(); ix.close()
