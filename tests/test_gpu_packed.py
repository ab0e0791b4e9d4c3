"""GPU tests of the host boundary's wire formats (include/bwt_gpu.h): reads uploaded PACKED (2 bits per base + exception list,
expanded to the bit-planes on the device) and hits fetched COMPACT (16-byte records + extras) give, bit for bit, what the ASCII
upload and the per-hit tables give -- and what the oracle says (the worker loop of src/worker_bwt.erl:73-106)."""
import numpy as np
import pytest

from conftest import synth_reads

pytestmark = pytest.mark.gpu

KEYS = ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off")


def oracle_check(out, reads, ofm, ref1):
    from oracle import orc
    nh = 0
    for i, q in enumerate(reads):
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        try:
            exp = orc.align_read(ofm, ref1, q)
        except RuntimeError:               # the reference crashes on this read (a sub-read ending outside ACGTN; a window outside Ref1)
            assert out["read_status"][i] < 0 and a == b
            continue
        if exp is None:
            assert out["read_status"][i] == 1 and a == b
            continue
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(), int(out["pos"][h]),
                int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
        assert got == exp, (i, q)
        nh += len(exp)
    return nh


def mixed(ref, length, seed, n=1200):
    return (synth_reads(ref, n, length, seed, max_subs=3) + synth_reads(ref, n // 3, length, seed + 1, indel_frac=1.0) +
            synth_reads(ref, n // 6, length, seed + 2, n_frac=1.0) + synth_reads(ref, n // 6, length, seed + 3, iupac_frac=1.0) +
            [b"N" * length, b"ACGT" * (length // 4) + b"A" * (length % 4), ref[:length], ref[-length:]])


@pytest.mark.parametrize("opts", [{}, {"sw_prefilter": 0}, {"sw_wave": 0, "sw_band": 0}, {"seed_fast": 0}], ids=str)
@pytest.mark.parametrize("length", [100, 150, 67])
def test_packed_upload_equals_ascii_upload_and_oracle(gpu_index, orc_fm, ref_bytes, length, opts):
    """One batch of one read length, uploaded both ways: identical tables, the oracle's answers.  Reads with N and IUPAC codes
    (exceptions), indels (DP kernels that read the read's BYTES: made from the planes for queued candidates only, or for the
    whole batch when no prefilter runs), reads that are all N."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = mixed(ref_bytes, length, 1300 + length)
    bases, off = lib.pack_reads(reads)
    codes, exc = lib.pack_2bit(bases)
    assert len(exc) > 100
    a = lib.Batch(gpu_index); a.set_options(**opts)
    a.set_reads(bases, off)
    ca = a.run(lib.STAGE_EXTEND)
    want = a.fetch_hits()
    p = lib.Batch(gpu_index); p.set_options(**opts)
    p.set_reads_packed(codes, exc, len(reads), read_len=length)
    cp = p.run(lib.STAGE_EXTEND)
    got = p.fetch_hits()
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    assert got["cigar"] == want["cigar"]
    for k in ("n_tasks", "n_clusters", "n_hits", "cigar_bytes", "sw_cells", "n_float", "n_dp"):
        assert ca[k] == cp[k], k
    assert oracle_check(got, reads, orc_fm, ref1) > 500
    # the compact table of the same run, expanded on the host, is the same again
    c = p.fetch_hits_compact()
    ex = lib.hits_expand(len(reads), length, None, gpu_index.meta()[1], c["hits"], c["extra"], c["cigar"])
    for k in KEYS[1:]:
        assert np.array_equal(ex[k], want[k]), k
    assert ex["cigar"] == want["cigar"] and np.array_equal(c["read_status"].astype(np.int32), want["read_status"])
    # substitutions only: plain; indels / N: extras (only the prefilter settles hits as plain)
    assert 0 < len(c["extra"]) <= len(c["hits"]) and (len(c["extra"]) < len(c["hits"]) or opts.get("sw_prefilter") == 0)
    a.destroy(); p.destroy()


def test_packed_upload_ragged_batch_and_reuse(gpu_index, orc_fm, ref_bytes):
    """Reads of different lengths (offsets go up with the codes), an empty batch, and a batch object reused for ASCII after packed."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = (synth_reads(ref_bytes, 300, 100, 41, indel_frac=0.3, n_frac=0.1) + synth_reads(ref_bytes, 200, 48, 42) +
             synth_reads(ref_bytes, 100, 333, 43, max_subs=6, iupac_frac=0.2) + [b"ACGTACG", b"", b"N"])
    rng = np.random.default_rng(5)
    reads = [reads[i] for i in rng.permutation(len(reads))]
    bases, off = lib.pack_reads(reads)
    codes, exc = lib.pack_2bit(bases)
    b = lib.Batch(gpu_index)
    b.set_reads_packed(codes, exc, len(reads), off=off)
    b.run(lib.STAGE_EXTEND)
    got = b.fetch_hits()
    assert oracle_check(got, reads, orc_fm, ref1) > 300
    c = b.fetch_hits_compact()
    ex = lib.hits_expand(len(reads), 0, off, gpu_index.meta()[1], c["hits"], c["extra"], c["cigar"])
    for k in KEYS[1:]:
        assert np.array_equal(ex[k], got[k]), k
    assert ex["cigar"] == got["cigar"]
    # empty packed batch
    b.set_reads_packed(np.zeros(2, np.uint32), np.zeros(0, np.uint64), 0, read_len=100)
    assert b.run(lib.STAGE_EXTEND)["n_hits"] == 0
    e = b.fetch_hits_compact()
    assert len(e["hits"]) == 0 and len(e["extra"]) == 0 and len(e["read_status"]) == 0
    # ... and the same object with an ASCII batch again
    b.set_reads(bases, off)
    b.run(lib.STAGE_EXTEND)
    again = b.fetch_hits()
    for k in KEYS:
        assert np.array_equal(again[k], got[k]), k
    b.destroy()


def test_packed_upload_rejects_bad_input(gpu_index):
    from bwt_amd import lib
    bases, off = lib.pack_reads([b"ACGTACGTAC" * 10] * 4)
    codes, exc = lib.pack_2bit(bases)
    b = lib.Batch(gpu_index)
    with pytest.raises(lib.BwtError) as e:                            # an exception that is no exception
        b.set_reads_packed(codes, np.array([5 | (ord("A") << 56)], np.uint64), 4, read_len=100)
    assert e.value.code == -22
    with pytest.raises(lib.BwtError) as e:                            # ... behind the last base
        b.set_reads_packed(codes, np.array([400 | (ord("N") << 56)], np.uint64), 4, read_len=100)
    assert e.value.code == -22
    for dup in ([7 | (ord("N") << 56), 7 | (ord("R") << 56)], [9 | (ord("N") << 56), 7 | (ord("N") << 56)]):
        with pytest.raises(lib.BwtError) as e:                        # ... two records for one base / descending positions
            b.set_reads_packed(codes, np.array(dup, np.uint64), 4, read_len=100)
        assert e.value.code == -22 and "ascending" in str(e.value)
    with pytest.raises(lib.BwtError):                                 # n * read_len != total
        b.set_reads_packed(codes, exc, 4, read_len=99, total=400)
    with pytest.raises(lib.BwtError):                                 # ragged without offsets
        b.set_reads_packed(codes, exc, 4, read_len=0, total=400)
    bad = off.copy(); bad[2] = 50                                     # decreasing offsets
    with pytest.raises(lib.BwtError):
        b.set_reads_packed(codes, exc, 4, off=bad)
    b.set_reads_packed(codes, exc, 4, read_len=100)                   # the batch is still usable
    assert b.run(lib.STAGE_EXTEND)["n_reads"] == 4
    b.destroy()


def test_compact_table_on_a_repeat_genome():
    """Reads with many hits each (a 300-bp segment in 40 copies): the compact table keeps the reference's list order inside a read,
    extras stay ascending by hit, and expansion equals the per-hit fetch."""
    from bwt_amd import lib
    rng = np.random.default_rng(77)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_500_000, dtype=np.uint8)].tobytes())
    seg = bytes(g[5000:5300])
    for k in range(1, 40):
        g[30_000 * k:30_000 * k + 300] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    try:
        reads = [bytes(seg[i:i + 100]) for i in range(0, 200, 3)] + synth_reads(ref, 400, 100, 9, indel_frac=0.5)
        bases, off = lib.pack_reads(reads)
        codes, exc = lib.pack_2bit(bases)
        b = lib.Batch(ix)
        b.set_reads_packed(codes, exc, len(reads), read_len=100)
        c = b.run(lib.STAGE_EXTEND)
        full = b.fetch_hits()
        comp = b.fetch_hits_compact()
        assert c["n_hits"] == len(comp["hits"]) > 40 * 60
        assert np.all(np.diff(comp["hits"]["read"].astype(np.int64)) >= 0) and np.all(np.diff(comp["extra"]["hit"].astype(np.int64)) > 0)
        ex = lib.hits_expand(len(reads), 100, None, 1, comp["hits"], comp["extra"], comp["cigar"])
        for k in KEYS[1:]:
            assert np.array_equal(ex[k], full[k]), k
        assert ex["cigar"] == full["cigar"]
        b.destroy()
    finally:
        ix.close()


def test_reads_with_hundreds_of_hits():
    """A 100-base unit in 300 exact copies and 60 copies with a 2-base deletion.  A 15-mer of it has 360 occurrences -- too many
    seeds (bwa.erl:33) -- but a sub-read whose search dies on a substituted base after 12 steps returns its WHOLE interval
    (fmi:get_seed_ends has no range check on that path): two such sub-reads and the read has a cluster at every copy.
    360 candidates per read: the hit kernels take such a read with the whole wave (HITS_FAT); plain hits (two mismatches on a
    square window), DP hits (six mismatches) and windows with the deletion in them, all in one read's list -- against the oracle,
    and the compact table against the per-hit one."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(4711)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    g = bytearray(acgt[rng.integers(0, 4, 2_000_000, dtype=np.uint8)].tobytes())
    unit = bytes(acgt[rng.integers(0, 4, 100, dtype=np.uint8)].tobytes())
    for k in range(360):
        at = 3000 + 5000 * k
        if k % 6 == 5:
            g[at:at + 98] = unit[:40] + unit[42:]
        else:
            g[at:at + 100] = unit
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    ix.set_ktable(15, 1)
    try:
        L, SA = ix.export()
        ptr, shift = ix.meta()
        ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
        ref1 = E.make_ref1(ref)

        def sub(q, ps):
            q = bytearray(q)
            for p_ in ps:
                q[p_] = b"ACGT"[(b"ACGT".index(q[p_]) + 1 + p_ % 3) & 3]
            return bytes(q)
        # a substitution 13 bases left of a sub-read's end (ends at 100, 92, 84, ...): that sub-read dies after 12 matched bases
        reads = [sub(unit, (87, 71)), sub(unit, (87, 71, 55, 39, 23, 7)), sub(unit, (87, 23)), sub(unit, (87, 55, 23))] * 2
        reads += synth_reads(ref, 200, 100, 12, max_subs=2)
        exp = [orc.align_read(ofm, ref1, q) for q in reads[:4]]
        assert min(len(e) for e in exp) > 300                   # (more than HITS_FAT candidates each)
        bases, off = lib.pack_reads(reads)
        codes, exc = lib.pack_2bit(bases)
        for opts in ({}, {"sw_band_min": 0}):
            b = lib.Batch(ix)
            b.set_options(**opts)
            b.set_reads_packed(codes, exc, len(reads), read_len=100)
            c = b.run(lib.STAGE_EXTEND)
            full = b.fetch_hits()
            comp = b.fetch_hits_compact()
            assert oracle_check(full, reads, ofm, ref1) > 8 * 300
            assert 0 < len(comp["extra"]) < len(comp["hits"]) == c["n_hits"]        # plain and non-plain hits
            assert np.all(np.diff(comp["hits"]["read"].astype(np.int64)) >= 0) and np.all(np.diff(comp["extra"]["hit"].astype(np.int64)) > 0)
            ex = lib.hits_expand(len(reads), 100, None, shift, comp["hits"], comp["extra"], comp["cigar"])
            for k in KEYS[1:]:
                assert np.array_equal(ex[k], full[k]), k
            assert ex["cigar"] == full["cigar"]
            b.destroy()
    finally:
        ix.close()


def test_reads_with_thousands_of_seeds():
    """The same construction with 5200 copies: a sub-read that dies after 12 matched bases returns 5200 seeds.  Two such sub-reads
    (10 400 seeds) and the read is past what k_cluster_big sorts in its own LDS (4096 seeds; the reads of the test above, 720 and 2160
    seeds, are that kernel's): it lists the read for k_cluster_huge (1024 threads, 30 720 seeds in LDS); six of them (31 200 seeds) and
    that kernel takes its HBM path.  About 5200 hits per read, against the oracle."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(5200)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    g = bytearray(acgt[rng.integers(0, 4, 2_700_000, dtype=np.uint8)].tobytes())
    unit = bytes(acgt[rng.integers(0, 4, 100, dtype=np.uint8)].tobytes())
    for k in range(5200):
        g[1000 + 500 * k:1100 + 500 * k] = unit
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    ix.set_ktable(15, 1)
    try:
        L, SA = ix.export()
        ptr, shift = ix.meta()
        ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
        ref1 = E.make_ref1(ref)

        def sub(q, ps):
            q = bytearray(q)
            for p_ in ps:
                q[p_] = b"ACGT"[(b"ACGT".index(q[p_]) + 1 + p_ % 3) & 3]
            return bytes(q)
        reads = [sub(unit, (87, 71)), sub(unit, (87, 71, 55, 39, 23, 7)), sub(unit, (87, 55, 23))]
        reads += synth_reads(ref, 100, 100, 13, max_subs=2)
        b = lib.Batch(ix)
        b.set_reads(*lib.pack_reads(reads))
        b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        b.destroy()
        nh = np.diff(out["hit_off"].astype(np.int64))
        assert oracle_check(out, reads, ofm, ref1) > 2 * 5000
        assert int((nh[:3] > 5000).sum()) >= 2, nh[:3]       # (whether two dumping sub-reads make a cluster per copy depends on the other sub-reads' seeds)
    finally:
        ix.close()


def test_pipe_packs_a_batch_that_is_mostly_exceptions(gpu_index, orc_fm, ref_bytes):
    """bwt_pipe_submit_ascii on a batch whose exception list outgrows the slot's buffer (reads that are mostly N): the slot's thread
    grows the pinned buffer and packs again; the answers are still the oracle's."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 200, 100, 77) + [b"N" * 100] * 300 + synth_reads(ref_bytes, 100, 100, 78, n_frac=1.0)
    bases, off = lib.pack_reads(reads)
    pipe = lib.Pipe(gpu_index, depth=1)
    try:
        s, v = pipe.acquire_packed(len(bases), len(reads), 16)            # room for 16 exceptions; the batch has > 30 000
        v["off"][:] = off
        v["ascii"][:len(bases)] = bases
        pipe.submit_ascii(s, len(reads), 100)
        r = pipe.wait()
        out = pipe.expand(r)
        pipe.release(r["slot"])
    finally:
        pipe.close()
    assert oracle_check(out, reads, orc_fm, ref1) > 100


def test_reference_with_n_runs_and_iupac_codes():
    """A reference whose .ref bytes hold N runs and IUPAC codes (fm_index.erl:130-137 keeps them in the .ref and substitutes bases
    only in the indexed sequence): windows over them score with sigma's 1.6 / IUPAC clauses (src/sw.erl:164-202).  Reads
    inside, across and next to those stretches, against the oracle."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(321)
    n = 400_000
    g = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)].copy()
    spots = []
    for st in range(1500, n - 1500, 1900):
        k = int(rng.integers(1, 60))
        g[st:st + k] = ord("N")                                          # an N run
        j = st + 700 + int(rng.integers(0, 300))
        g[j] = int(rng.choice(list(b"RYKMSWBDV")))                       # an IUPAC code (no H: fm_index.erl:136 has no clause for it)
        spots += [st, j]
    ref = g.tobytes()
    ix = lib.Index.build(ref, shift=1)
    try:
        L, SA = ix.export()
        ptr, shift = ix.meta()
        ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
        ref1 = E.make_ref1(ref)
        reads = []
        for s_ in spots:                                                 # reads that start before, across and after every spot
            for d in (-130, -99, -60, -20, -1, 0, 3, 31, 33, 64):
                a = s_ + d
                q = bytearray(ref[a:a + 100])
                for i in range(len(q)):                                  # reads carry bases, not the reference's codes (one in four keeps an N)
                    if q[i] not in b"ACGT" and (q[i] != ord("N") or rng.random() < 0.75):
                        q[i] = int(rng.choice(list(b"ACGT")))
                if rng.random() < 0.3:
                    q[int(rng.integers(0, 100))] = int(rng.choice(list(b"ACGT")))
                reads.append(bytes(q))
        reads += synth_reads(ref.replace(b"N", b"A"), 500, 100, 5)
        bases, off = lib.pack_reads(reads)
        b = lib.Batch(ix)
        codes, exc = lib.pack_2bit(bases)
        b.set_reads_packed(codes, exc, len(reads), read_len=100)
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        b.destroy()
        assert oracle_check(out, reads, ofm, ref1) > 1000
        assert c["n_float"] > 200                                        # windows with N / IUPAC went to the fp64 kernel
    finally:
        ix.close()


def test_pipe_api_errors(gpu_index):
    """The pipe's error behaviour: every misuse is a BWT_EINVAL / BWT_ENOSPC with a text, never a crash; a failed batch is reported
    by wait and its slot can be released and used again."""
    from bwt_amd import lib
    pipe = lib.Pipe(gpu_index, depth=2)
    try:
        with pytest.raises(lib.BwtError) as e:                      # nothing submitted
            pipe.wait()
        assert e.value.code == -22
        with pytest.raises(lib.BwtError):                           # no such slot
            pipe.submit_packed(7, 1, 100, 100, 0)
        with pytest.raises(lib.BwtError):                           # slot not acquired
            pipe.submit_packed(0, 1, 100, 100, 0)
        s, v = pipe.acquire_packed(400, 4, 4)
        with pytest.raises(lib.BwtError):                           # more reads than acquired for
            pipe.submit_packed(s, 5, 100, 500, 0)
        with pytest.raises(lib.BwtError):                           # total_bases disagrees with n * read_len
            pipe.submit_packed(s, 4, 100, 399, 0)
        with pytest.raises(lib.BwtError):                           # an ASCII submit on a slot acquired for packed input is fine, the other way round is not
            pipe.submit(s, 4)
        s2, v2 = pipe.acquire_packed(400, 4, 4)
        assert pipe.acquire_packed(16, 1, 0) is None                 # both slots taken: BWT_ENOSPC -> None
        # a batch that fails on the device side of the boundary: an exception record that names one of ACGT
        v["codes"][:] = 0
        v["exc"][0] = 5 | (ord("C") << 56)
        pipe.submit_packed(s, 4, 100, 400, 1)
        with pytest.raises(lib.BwtError) as e:
            pipe.wait()
        assert e.value.code == -22 and "ACGT" in str(e.value)
        pipe.release(s)
        pipe.release(s2)                                              # (acquired, never submitted: may be given back)
        with pytest.raises(lib.BwtError):
            pipe.release(s2)                                          # ... once
        # and the pipe still works
        s, v = pipe.acquire_packed(400, 4, 4)
        bases, off = lib.pack_reads([b"ACGTACGTAC" * 10] * 4)
        codes, exc = lib.pack_2bit(bases)
        v["codes"][:len(codes)] = codes
        pipe.submit_packed(s, 4, 100, 400, 0)
        r = pipe.wait()
        assert r["n_reads"] == 4 and r["read_len"] == 100
        pipe.release(r["slot"])
    finally:
        pipe.close()


@pytest.mark.parametrize("opts", [{}, {"sw_prefilter": 0}, {"seed_fast": 0}], ids=str)
def test_narrow_table_equals_compact_table_and_oracle(gpu_index, orc_fm, ref_bytes, opts):
    """bwt_options.hits_narrow: a read whose answer is one plain hit leaves the 16-byte record stream (status 2, window offset u32 +
    score i16 in the narrow arrays).  The table decoded on the host (bwt_hits_expand_narrow), the same table turned back into
    records on the device (what every per-hit / compact entry point does first), and a run without the option are one and the same,
    and the oracle's."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = mixed(ref_bytes, 100, 77) + synth_reads(ref_bytes, 600, 100, 78, max_subs=1)
    bases, off = lib.pack_reads(reads)
    plain = lib.Batch(gpu_index); plain.set_options(**opts)
    plain.set_reads(bases, off); plain.run(lib.STAGE_EXTEND)
    want = plain.fetch_hits()
    b = lib.Batch(gpu_index); b.set_options(hits_narrow=1, **opts)
    b.set_reads(bases, off); b.run(lib.STAGE_EXTEND)
    t = b.fetch_hits_narrow()
    nn = len(t["narrow_off"])
    assert int((t["read_status"] == 2).sum()) == nn and (nn > 300 or opts.get("sw_prefilter") == 0)       # (only the prefilter settles hits as plain)
    assert len(t["hits"]) + nn == len(want["score"]) and not np.isin(np.nonzero(t["read_status"] == 2)[0], t["hits"]["read"]).any()
    ex = lib.hits_expand(len(reads), 0, off, gpu_index.meta()[1], t["hits"], t["extra"], t["cigar"], t["read_status"], t["narrow_off"], t["narrow_score"])
    for k in KEYS[1:]:
        assert np.array_equal(ex[k], want[k]), k
    assert ex["cigar"] == want["cigar"]
    st = t["read_status"].astype(np.int32); st[st == 2] = 0
    assert np.array_equal(st, want["read_status"])
    t2 = b.fetch_hits_narrow()                                  # (fetching twice: the same table)
    assert all(np.array_equal(t[k], t2[k]) for k in ("read_status", "narrow_off", "narrow_score", "hits", "extra")) and t["cigar"] == t2["cigar"]
    # ... and through the entry points that hand out records: widened on the device first
    c = b.fetch_hits_compact()
    assert len(c["hits"]) == len(want["score"]) and not (c["read_status"] == 2).any()
    ex2 = lib.hits_expand(len(reads), 0, off, gpu_index.meta()[1], c["hits"], c["extra"], c["cigar"])
    got = b.fetch_hits()
    for k in KEYS:
        assert np.array_equal(got[k], want[k]), k
    for k in KEYS[1:]:
        assert np.array_equal(ex2[k], want[k]), k
    assert got["cigar"] == want["cigar"] and ex2["cigar"] == want["cigar"]
    assert len(b.fetch_hits_narrow()["narrow_off"]) == 0       # (the table is in its 16-byte form now)
    assert oracle_check(got, reads, orc_fm, ref1) > 500
    # a batch with a read too long for 16-bit scores: no narrow reads at all, same answers
    long_reads = reads[:50] + [ref_bytes[2000:2000 + 17000]]
    lb, lo = lib.pack_reads(long_reads)
    b.set_reads(lb, lo); b.run(lib.STAGE_EXTEND)
    tl = b.fetch_hits_narrow()
    assert len(tl["narrow_off"]) == 0 and not (tl["read_status"] == 2).any()
    plain.set_reads(lb, lo); plain.run(lib.STAGE_EXTEND)
    wl, gl = plain.fetch_hits(), b.fetch_hits()
    for k in KEYS:
        assert np.array_equal(gl[k], wl[k]), k
    plain.destroy(); b.destroy()


def test_pipe_results_in_the_narrow_form(gpu_index, ref_bytes):
    """a pipe whose batches run with hits_narrow: the result carries the narrow arrays, expand() decodes them, and the table equals the
    synchronous call's"""
    from bwt_amd import lib
    reads = synth_reads(ref_bytes, 3000, 100, 5, max_subs=2) + mixed(ref_bytes, 100, 6)
    bases, off = lib.pack_reads(reads)
    want = gpu_index.align_batch(bases, off)
    codes, exc = lib.pack_2bit(bases)
    pipe = lib.Pipe(gpu_index, depth=2)
    pipe.set_options(hits_narrow=1)
    try:
        for _ in range(3):
            s, v = pipe.acquire_packed(len(bases), len(reads), len(exc))
            v["codes"][:len(codes)] = codes; v["exc"][:len(exc)] = exc
            pipe.submit_packed(s, len(reads), 100, len(bases), len(exc))
            r = pipe.wait()
            assert len(r["narrow_off"]) > 1000 and len(r["narrow_off"]) + len(r["hits"]) == len(want["score"])
            out = pipe.expand(r)
            for k in KEYS:
                assert np.array_equal(out[k], want[k]), k
            assert out["cigar"] == want["cigar"]
            pipe.release(r["slot"])
    finally:
        pipe.close()


def test_zero_copy_upload_only_reads_pinned_buffers(gpu_index, ref_bytes):
    """up_zero_copy = 1: a pipe slot's pinned `codes` buffer is read in place by the unpack kernel (no copy into device memory); a
    caller's pageable buffer under the same option is copied as before (a kernel reading it would fault).  Both give the default
    upload's table, with and without the whole-batch ASCII pass (sw_prefilter = 0 reads the codes a second time)."""
    from bwt_amd import lib
    reads = synth_reads(ref_bytes, 3000, 100, 15, max_subs=2) + mixed(ref_bytes, 100, 16)
    bases, off = lib.pack_reads(reads)
    want = gpu_index.align_batch(bases, off)
    codes, exc = lib.pack_2bit(bases)
    for extra in ({}, {"sw_prefilter": 0}):
        pipe = lib.Pipe(gpu_index, depth=2)
        pipe.set_options(up_zero_copy=1, **extra)
        try:
            for _ in range(3):
                s, v = pipe.acquire_packed(len(bases), len(reads), len(exc))
                v["codes"][:len(codes)] = codes; v["exc"][:len(exc)] = exc
                pipe.submit_packed(s, len(reads), 100, len(bases), len(exc))
                r = pipe.wait()
                out = pipe.expand(r)
                for k in KEYS:
                    assert np.array_equal(out[k], want[k]), k
                assert out["cigar"] == want["cigar"]
                pipe.release(r["slot"])
        finally:
            pipe.close()
        b = lib.Batch(gpu_index)
        b.set_options(up_zero_copy=1, **extra)
        b.set_reads_packed(codes, exc, len(reads), read_len=100)       # numpy arrays: pageable
        b.run(lib.STAGE_EXTEND)
        got = b.fetch_hits()
        for k in KEYS:
            assert np.array_equal(got[k], want[k]), k
        b.destroy()
