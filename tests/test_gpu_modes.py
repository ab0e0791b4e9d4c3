"""GPU tests of the library's execution modes and of its threading contract (include/bwt_gpu.h):
every experiment switch (read per call) must give the oracle's answers, and several host threads may
run batches on ONE index at the same time (the NIF's dirty-scheduler case)."""
import threading

import numpy as np
import pytest

from conftest import synth_reads

pytestmark = pytest.mark.gpu


def hits_of(out, i):
    a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
    return [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
             int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]


def check_against_oracle(out, reads, orc_fm, ref1):
    from oracle import orc
    nh = 0
    for i, q in enumerate(reads):
        try:
            exp = orc.align_read(orc_fm, ref1, q)
        except RuntimeError:
            assert out["read_status"][i] == -1
            continue
        if exp is None:
            assert out["read_status"][i] == 1 and out["hit_off"][i] == out["hit_off"][i + 1]
            continue
        assert hits_of(out, i) == exp, (i, q)
        nh += len(exp)
    return nh


# every execution option of bwt_options (include/bwt_gpu.h) that selects between equivalent kernels, and the table's summary bitmap
MODES = [{"seed_big": 1}, {"ktab_summary": 2}, {"ktab_summary": 0}, {"sw_band": 0}, {"sw_band": 2, "sw_band_min": 0}, {"sw_band_min": 0}, {"sw_band_min": 0, "sw_wave": 0},
         {"seed_fuse": 0}, {"sw_prefilter": 0}, {"sw_pk": 0}, {"sw_pk": 2}, {"sw_lds": 0}, {"sw_trace_all": 0, "sw_wave": 0},
         {"sw_wave": 0, "sw_prefilter": 0}, {"search_ilp": 2}, {"seed_fast": 0}, {"sw_fuse_cand": 0}, {"sw_band12": 0, "sw_band_min": 0},
         {"hits_narrow": 1}, {"sw_cert": 0, "sw_band_min": 0}]


@pytest.mark.parametrize("mode", MODES, ids=lambda m: ",".join("%s=%s" % kv for kv in m.items()))
@pytest.mark.parametrize("K", ["15w", 0])
def test_experiment_switches_keep_parity(gpu_index, orc_fm, ref_bytes, mode, K):
    """The options go in through bwt_index_set_options (the batches of the host-buffer entry points take the index's) -- the
    library itself never reads the environment (tests/test_abi.py checks that)."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    mode = dict(mode)
    summary = mode.pop("ktab_summary", 1)
    wide = isinstance(K, str)
    gpu_index.set_ktable(15 if wide else K, 1 if wide else 0, summary)
    gpu_index.set_options(**mode)
    try:
        ref1 = E.make_ref1(ref_bytes)
        reads = (synth_reads(ref_bytes, 500, 100, 301, max_subs=3) + synth_reads(ref_bytes, 250, 100, 302, indel_frac=1.0) +
                 synth_reads(ref_bytes, 80, 100, 303, n_frac=1.0))
        bases, off = lib.pack_reads(reads)
        out = gpu_index.align_batch(bases, off)
        assert check_against_oracle(out, reads, orc_fm, ref1) > 300
    finally:
        gpu_index.set_options()
        gpu_index.set_ktable(-1)


def test_concurrent_batches_on_one_index(gpu_index, orc_fm, ref_bytes):
    """Four host threads, each with its own batches (own HIP stream) on the same index, several rounds each: every
    result equals the oracle's (ctypes releases the GIL inside the library calls, so the calls really overlap)."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    sets = [synth_reads(ref_bytes, 600 + 150 * t, 100 if t % 2 == 0 else 150, 400 + t, indel_frac=0.2, n_frac=0.05) for t in range(4)]
    packed = [lib.pack_reads(s) for s in sets]
    results = [[None] * 3 for _ in range(4)]
    errors = []

    def work(t):
        try:
            for rnd in range(3):
                if rnd == 1:
                    results[t][rnd] = gpu_index.align_batch(*packed[t])          # host-buffer entry point (own batch per call)
                else:
                    b = lib.Batch(gpu_index)
                    b.set_reads(*packed[t])
                    b.run(lib.STAGE_EXTEND)
                    results[t][rnd] = b.fetch_hits()
                    b.destroy()
        except Exception as e:                                                    # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    for t in range(4):
        for rnd in range(3):
            assert check_against_oracle(results[t][rnd], sets[t], orc_fm, ref1) > 200


def test_set_ktable_waits_for_running_batches(gpu_index, orc_fm, ref_bytes):
    """bwt_index_set_ktable takes the index's exclusive lock: called while another thread runs batches it neither
    corrupts their results nor crashes."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 800, 100, 501)
    packed = lib.pack_reads(reads)
    outs, errors = [], []

    def runner():
        try:
            for _ in range(6):
                outs.append(gpu_index.align_batch(*packed))
        except Exception as e:                                                    # noqa: BLE001
            errors.append(repr(e))

    t = threading.Thread(target=runner)
    t.start()
    try:
        for K, wide in ((9, 0), (15, 1), (12, 0), (0, 0)):
            gpu_index.set_ktable(K, wide)
    finally:
        t.join()
        gpu_index.set_ktable(-1)
    assert not errors, errors
    for out in outs:
        assert check_against_oracle(out, reads, orc_fm, ref1) > 300


@pytest.mark.parametrize("length", [103, 100])
def test_fast_seed_kernel_against_general_and_oracle(length):
    """k_seed_fast (uniform read length, K = 15 wide table) with reads that make it hand groups back (N, IUPAC, repeats
    with many seeds / clusters): same hit tables as the general kernel alone, and the oracle's answers.  103-bp reads:
    every sub-read has >= 15 bases (all tasks end in the table's top level); 100-bp reads on this small genome: the
    12-base last sub-read has seeds below the top level, which the kernel reads from the SA."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(91)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 3_000_000, dtype=np.uint8)].tobytes())
    seg = bytes(g[1000:1300])
    for k in range(1, 9):                                            # a 300-bp repeat in 9 copies: reads from it have ~9 clusters
        g[200_000 * k:200_000 * k + 300] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=3)
    try:
        ix.set_ktable(15, 1)
        L, SA = ix.export()
        ptr, shift = ix.meta()
        ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
        ref1 = E.make_ref1(ref)
        reads = (synth_reads(ref, 4000, length, 601, max_subs=2) + synth_reads(ref, 300, length, 602, indel_frac=1.0) +
                 synth_reads(ref, 100, length, 603, n_frac=1.0) + synth_reads(ref, 60, length, 604, iupac_frac=1.0) +
                 [ref[1000 + 7 * i:1000 + length + 7 * i] for i in range(25)] + [ref[:length], ref[5:5 + length], ref[-length:]])
        order = rng.permutation(len(reads))
        reads = [reads[i] for i in order]
        packed = lib.pack_reads(reads)
        outs = {}
        for fast in ("1", "0"):
            b = lib.Batch(ix)
            b.set_options(seed_fast=int(fast))
            b.set_reads(*packed)
            c = b.run(lib.STAGE_EXTEND)
            outs[fast] = (b.fetch_hits(), c)
            b.destroy()
        c1, c0 = outs["1"][1], outs["0"][1]
        assert c0["n_handed_back"] == 0
        assert 0 < c1["n_handed_back"] < len(reads) // 8             # reads are handed back one by one: nearly all stay with the fast kernel
        assert check_against_oracle(outs["1"][0], reads, ofm, ref1) > 1500
        for k in ("n_tasks", "n_lookups", "sa_reads", "n_clusters", "n_hits", "cigar_bytes", "sw_cells"):
            assert c1[k] == c0[k], k
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(outs["1"][0][k], outs["0"][0][k]), k
        assert outs["1"][0]["cigar"] == outs["0"][0]["cigar"]
    finally:
        ix.close()


@pytest.mark.parametrize("form", ["packed", "ascii_staging", "ascii_pinned"])
def test_pipe_overlapped_batches_equal_align_batch(gpu_index, ref_bytes, form):
    """bwt_pipe: seven batches of different sizes through a pipe of depth 3 (pinned buffers, a host thread and a HIP stream per
    slot), the reads going up in each of the three input forms -- packed by the caller (2 bits per base + exceptions), ASCII left in
    the slot's staging buffer and packed by the pipe, ASCII uploaded as it is; the compact hit tables that come back, expanded on
    the host, equal bwt_align_batch's arrays, in submission order, empty and ragged batches included."""
    from bwt_amd import lib
    sets = [synth_reads(ref_bytes, n, 100, 700 + i, indel_frac=0.2, n_frac=0.05, iupac_frac=0.03) for i, n in enumerate((900, 1, 0, 1500, 300, 700))]
    sets.append(synth_reads(ref_bytes, 200, 100, 790, indel_frac=0.3) + synth_reads(ref_bytes, 150, 73, 791, n_frac=0.2) + synth_reads(ref_bytes, 50, 151, 792))
    packed = [lib.pack_reads(s) for s in sets]
    want = [gpu_index.align_batch(*p) for p in packed]
    pipe = lib.Pipe(gpu_index, depth=3)
    got, nxt = [], 0
    try:
        while len(got) < len(sets):
            slot = None
            if nxt < len(sets):
                bases, off = packed[nxt]
                n, total = len(sets[nxt]), len(bases)
                uniform = 100 if nxt < 6 else 0
                if form == "ascii_pinned":
                    slot = pipe.acquire(total, n)
                    if slot is not None:
                        s, pb, po = slot
                        pb[:total] = bases
                        po[:] = off
                        pipe.submit(s, n)
                else:
                    codes, exc = lib.pack_2bit(bases)
                    slot = pipe.acquire_packed(total, n, len(exc))
                    if slot is not None:
                        s, v = slot
                        v["off"][:] = off
                        if form == "packed":
                            v["codes"][:len(codes)] = codes
                            v["exc"][:len(exc)] = exc
                            pipe.submit_packed(s, n, uniform, total, len(exc))
                        else:
                            v["ascii"][:total] = bases
                            pipe.submit_ascii(s, n, uniform)
            if slot is not None:
                nxt += 1
                continue
            r = pipe.wait()                                    # every slot busy (or nothing left to submit): take the oldest result
            g = pipe.expand(r, packed[len(got)][1])
            g["counts"] = r["counts"]; g["n_extra"] = len(r["extra"])
            got.append(g)
            pipe.release(r["slot"])
        assert pipe.acquire(16, 1) is not None                 # all slots free again
    finally:
        pipe.close()
    for g, w in zip(got, want):
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(g[k], w[k]), k
        assert g["cigar"] == w["cigar"]
        assert g["counts"]["n_hits"] == len(w["score"])
    # most hits of the big batches are plain (no extra record), the reads with indels are not
    assert 0 < got[3]["n_extra"] < got[3]["counts"]["n_hits"]


@pytest.fixture(scope="module")
def repeat_index():
    """A 2 M-row genome with a 400-bp segment in 7 copies (reads from it carry 60..130 seeds and several clusters), K = 15 wide."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(123)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 2_000_000, dtype=np.uint8)].tobytes())
    seg = bytes(g[5000:5400])
    for k in range(1, 7):
        g[250_000 * k:250_000 * k + 400] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=2)
    ix.set_ktable(15, 1)
    L, SA = ix.export()
    ptr, shift = ix.meta()
    yield ix, ref, orc.OrcFM.from_L_SA(L, SA, ptr, shift), E.make_ref1(ref)
    ix.close()


@pytest.mark.parametrize("length", [64, 71, 72, 89, 96, 111, 112, 120, 128, 150, 168, 175])
def test_fast_seed_kernel_read_lengths(repeat_index, length):
    """Every number of sub-reads per read the fast seed kernel takes (8..21: 8..3 reads per wave pass), in both variants
    (one / two registers per read), against the oracle -- with reads from a repeat (many seeds, several clusters), indels, N."""
    from bwt_amd import lib
    ix, ref, ofm, ref1 = repeat_index
    reads = (synth_reads(ref, 300, length, 800 + length, max_subs=2) + synth_reads(ref, 60, length, 801 + length, indel_frac=1.0) +
             synth_reads(ref, 20, length, 802 + length, n_frac=1.0) +
             [ref[5000 + 3 * i:5000 + 3 * i + length] for i in range(40)] + [ref[250_000 + 11 * i:250_000 + 11 * i + length] for i in range(20)])
    packed = lib.pack_reads(reads)
    kept = []
    for big in ("0", "1"):
        b = lib.Batch(ix)
        b.set_options(seed_big=int(big))
        b.set_reads(*packed)
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        b.destroy()
        assert check_against_oracle(out, reads, ofm, ref1) > 100
        kept.append(len(reads) - c["n_handed_back"])
    # both variants of the fast kernel really took reads (on this small genome the 8..10-base last sub-reads of the longest reads
    # alone bring ~30 seeds: past 64 per read for the one-register variant; the two-register variant holds up to 128 but sends the
    # other reads of a pass along with one it hands back)
    assert kept[0] > 0 and kept[1] > 0


@pytest.mark.parametrize("length", [48, 64, 100, 150, 175, 400])
def test_banded_dp_settles_what_its_certificate_covers(gpu_index, orc_fm, ref_bytes, length):
    """k_sw_band runs the DP on 20 (then 32, then 64) diagonals around the cluster and keeps its result only when the best score proves that
    nothing outside the band can matter (kernels.hip.h); everything else goes on to the full-rectangle kernels.
    Reads with one indel and up to 4 substitutions (most are settled by the band), reads with up to 14 substitutions (few are):
    every answer must be the oracle's, and both routes must have been taken."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = (synth_reads(ref_bytes, 500, length, 900 + length, max_subs=4, indel_frac=1.0) +
             synth_reads(ref_bytes, 300, length, 901 + length, max_subs=14, indel_frac=0.5))
    # the third tier's: a clean stretch that brings the seeds and the cluster, the rest of the read at 20-45 % substitutions (what a
    # read from a diverged copy of a repeat family looks like against another copy: a loss of 40..100 on a square window)
    rng = np.random.default_rng(77 + length)
    n_heavy = 0
    if length >= 100:
        for _ in range(400):
            a = int(rng.integers(0, len(ref_bytes) - length))
            q = bytearray(ref_bytes[a:a + length])
            c0 = int(rng.integers(0, length - 40))
            rate = float(rng.uniform(0.2, 0.45))
            for i in range(length):
                if not (c0 <= i < c0 + 40) and rng.random() < rate:
                    q[i] = int(rng.choice([x for x in b"ACGT" if x != q[i]]))
            reads.append(bytes(q))
            n_heavy += 1
    b = lib.Batch(gpu_index)
    b.set_options(sw_band_min=0)                                # (queues this short are k_sw_wave's otherwise)
    try:
        b.set_reads(*lib.pack_reads(reads))
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        assert check_against_oracle(out, reads, orc_fm, ref1) > 200
        assert c["n_banded"] > 100 and c["n_dp"] > c["n_banded"], c
        # the first two tiers alone: the same tables, and (long reads) fewer candidates settled -- the 64-diagonal tier took the rest
        b.set_options(sw_band_min=0, sw_band=2)
        c2 = b.run(lib.STAGE_EXTEND)
        out2 = b.fetch_hits()
        assert c2["n_dp"] == c["n_dp"] and c2["n_banded"] <= c["n_banded"]
        if n_heavy:
            assert c["n_banded"] - c2["n_banded"] > 50, (c, c2)
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(out[k], out2[k]), k
        assert out["cigar"] == out2["cigar"]
        # the same batch without the band: the same tables
        b.set_options(sw_band_min=0, sw_band=0)
        c0 = b.run(lib.STAGE_EXTEND)
        out0 = b.fetch_hits()
        assert c0["n_banded"] == 0 and c0["n_certified"] == 0 and c0["n_dp"] == c["n_dp"] + c["n_certified"]      # (no band tiers: no certificate kernel either)
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(out[k], out0[k]), k
        assert out["cigar"] == out0["cigar"]
    finally:
        b.destroy()


def test_twelve_diagonal_tier(gpu_index, orc_fm, ref_bytes):
    """Reads with 3..8 substitutions and no indel: square windows the prefilter cannot settle (more than two mismatches).  Up to six
    mismatches score >= 2 Lq - 18 on the diagonal: the prefilter queues them for the 12-diagonal tier of the banded DP, whose
    certificate covers exactly that; with more they start at 20 diagonals as before.  The oracle's answers either way, the same
    tables with the tier switched off, and the counts say the tier was used."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    rng = np.random.default_rng(12)
    reads = []
    for length, n in ((100, 900), (150, 500), (40, 200)):
        for _ in range(n):
            a = int(rng.integers(0, len(ref_bytes) - length))
            q = bytearray(ref_bytes[a:a + length])
            for p in rng.choice(length, int(rng.integers(3, 9)), replace=False):
                q[int(p)] = int(rng.choice([x for x in b"ACGT" if x != q[int(p)]]))
            reads.append(bytes(q))
    b = lib.Batch(gpu_index)
    b.set_options(sw_band_min=0, sw_cert=0)
    try:
        b.set_reads(*lib.pack_reads(reads))
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        assert check_against_oracle(out, reads, orc_fm, ref1) > 500
        assert c["n_banded"] > 400 and c["n_dp"] >= c["n_banded"] and c["n_certified"] == 0, c
        b.set_options(sw_band_min=0, sw_band12=0)
        c0 = b.run(lib.STAGE_EXTEND)
        out0 = b.fetch_hits()
        assert c0["n_dp"] == c["n_dp"] and c0["n_banded"] == c["n_banded"]
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(out[k], out0[k]), k
        assert out["cigar"] == out0["cigar"]
        b.set_options(sw_band12=1, sw_cert=0)                       # the short-queue route: every tier passes its queue on to k_sw_wave
        b.run(lib.STAGE_EXTEND)
        out1 = b.fetch_hits()
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(out[k], out1[k]), k
        assert out["cigar"] == out1["cigar"]
        # the certificate kernel in front of the tier (the default): the same tables, most of the tier's queue settled without a DP
        b.set_options(sw_band_min=0)
        c2 = b.run(lib.STAGE_EXTEND)
        out2 = b.fetch_hits()
        assert c2["n_certified"] > 300 and c2["n_dp"] + c2["n_certified"] == c["n_dp"], (c2, c)
        for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
            assert np.array_equal(out[k], out2[k]), k
        assert out["cigar"] == out2["cigar"]
    finally:
        b.destroy()


def _sub_reads(rng, ref, n, lengths, where=None, clump=False):
    """reads with 3..10 substitutions (both instances of the certificate kernel) and no indel; where: (start, length) regions to sample from; clump: the substitutions within a
    dozen bases of each other (and of a read end, one time in three)"""
    reads = []
    for i in range(n):
        L = int(lengths[i % len(lengths)])
        if where:
            st, ln = where[int(rng.integers(0, len(where)))]
            a = int(np.clip(int(rng.integers(st - L // 2, st + ln - L // 2)), 0, len(ref) - L))
        else:
            a = int(rng.integers(0, len(ref) - L))
        q = bytearray(ref[a:a + L])
        m = int(rng.integers(3, 11))
        if clump:
            c0 = 0 if i % 3 == 0 else L - 13 if i % 3 == 1 else int(rng.integers(0, L - 12))
            ps = c0 + rng.choice(12, m, replace=False)
        else:
            ps = rng.choice(L, m, replace=False)
        for p in ps:
            q[int(p)] = int(rng.choice([x for x in b"ACGT" if x != q[int(p)]]))
        reads.append(bytes(q))
    return reads


def _same_tables(a, b, what):
    for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
        assert np.array_equal(a[k], b[k]), (what, k)
    assert a["cigar"] == b["cigar"], what


def test_certificate_kernel_every_word_count(gpu_index, orc_fm, ref_bytes):
    """k_sw_cert (bwt_amd/csrc/sw_cert.h): read lengths around its word boundaries (the four-word instance takes up to 128 bases, the
    five-word one up to 160, longer batches skip it), substitutions spread and in clumps at the read's ends -- where its start and
    end rules decide.  Against the oracle, and against the same batch with the kernel switched off."""
    from bwt_amd import lib
    from oracle import erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    rng = np.random.default_rng(77)
    for lengths, expect in (((100,), True), ((24, 31, 32, 33, 64, 96, 127, 128), True), ((129, 150, 159, 160), True), ((100, 161), False)):
        reads = _sub_reads(rng, ref_bytes, 700, lengths) + _sub_reads(rng, ref_bytes, 500, lengths, clump=True)
        b = lib.Batch(gpu_index)
        b.set_options(sw_band_min=0)
        try:
            b.set_reads(*lib.pack_reads(reads))
            c = b.run(lib.STAGE_EXTEND)
            out = b.fetch_hits()
            assert check_against_oracle(out, reads, orc_fm, ref1) > 300
            assert (c["n_certified"] > 200) == expect, (lengths, c)
            b.set_options(sw_band_min=0, sw_cert=0)
            c0 = b.run(lib.STAGE_EXTEND)
            assert c0["n_certified"] == 0 and c0["n_dp"] == c["n_dp"] + c["n_certified"], (c0, c)
            _same_tables(out, b.fetch_hits(), lengths)
            b.set_options(sw_band_min=0, sw_cert=2)                  # the 11-diagonal instance alone: fewer settled, the same tables
            c2 = b.run(lib.STAGE_EXTEND)
            assert c2["n_certified"] <= c["n_certified"] and c2["n_dp"] + c2["n_certified"] == c0["n_dp"], (c2, c)
            if expect:
                assert c2["n_certified"] < c["n_certified"], (c2, c)
            _same_tables(out, b.fetch_hits(), lengths)
        finally:
            b.destroy()


@pytest.fixture(scope="module")
def tandem_index():
    """A 1 M-row genome with 240 tandem repeats (periods 1..31, 120..300 bp each): alignments there have many equally good
    placements of a gap and near-perfect matches a few diagonals away -- the case a banded DP must not get wrong."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(4242)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_000_000, dtype=np.uint8)].tobytes())
    regions = []
    for k in range(240):
        period = int(rng.choice([1, 2, 3, 4, 5, 7, 11, 17, 23, 31]))
        ln = int(rng.integers(120, 301))
        st = 2000 + 4000 * k
        unit = bytes(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, period, dtype=np.uint8)].tobytes())
        body = bytearray((unit * (ln // period + 2))[:ln])
        for _ in range(int(rng.integers(0, 4))):                       # a few impurities
            body[int(rng.integers(0, ln))] = int(rng.choice(list(b"ACGT")))
        g[st:st + ln] = body
        regions.append((st, ln, period))
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=0)
    ix.set_ktable(15, 1)
    L, SA = ix.export()
    ptr, shift = ix.meta()
    yield ix, ref, regions, orc.OrcFM.from_L_SA(L, SA, ptr, shift), E.make_ref1(ref)
    ix.close()


def test_certificate_kernel_in_tandem_repeats(tandem_index):
    """Substitution-only reads inside tandem tracts: the neighbouring diagonals of their square windows match almost everywhere (period
    1..5 tracts put a perfect copy within the certificate's eleven diagonals), so a detour is often as good as the diagonal -- the case
    the certificate must refuse, not settle.  Whatever it settles has to be the oracle's answer."""
    from bwt_amd import lib
    ix, ref, regions, ofm, ref1 = tandem_index
    rng = np.random.default_rng(4711)
    where = [(st, ln) for (st, ln, period) in regions]
    short = [(st, ln) for (st, ln, period) in regions if period <= 5]
    reads = (_sub_reads(rng, ref, 700, (100, 150, 64), where) + _sub_reads(rng, ref, 500, (100, 128), where, clump=True) +
             _sub_reads(rng, ref, 400, (100, 48), short, clump=True))
    b = lib.Batch(ix)
    b.set_options(sw_band_min=0)
    try:
        b.set_reads(*lib.pack_reads(reads))
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        assert check_against_oracle(out, reads, ofm, ref1) > 300
        assert c["n_certified"] > 50, c
        b.set_options(sw_band_min=0, sw_cert=0)
        c0 = b.run(lib.STAGE_EXTEND)
        assert c0["n_dp"] == c["n_dp"] + c["n_certified"], (c0, c)
        _same_tables(out, b.fetch_hits(), "tandem")
    finally:
        b.destroy()


@pytest.mark.parametrize("length", [100, 150])
def test_banded_dp_in_tandem_repeats(tandem_index, length):
    from bwt_amd import lib
    ix, ref, regions, ofm, ref1 = tandem_index
    rng = np.random.default_rng(99 + length)
    reads = []
    for (st, ln, period) in regions:
        for _ in range(4):
            a = int(rng.integers(st - length + 30, st + ln - 30))
            q = bytearray(ref[a:a + length + 40])
            kind = rng.random()
            if kind < 0.7:                                               # an indel inside (or next to) the repeat
                p = int(np.clip(int(rng.integers(st, st + ln)) - a, 5, length - 5))
                k = period * int(rng.integers(1, 3)) if rng.random() < 0.6 else int(rng.integers(1, 6))
                k = max(1, min(k, 12))
                if rng.random() < 0.5:
                    del q[p:p + k]
                else:
                    q[p:p] = q[max(0, p - k):p] if rng.random() < 0.6 else bytes(rng.choice(list(b"ACGT"), k).tolist())
            q = q[:length]
            for _ in range(int(rng.integers(0, 3))):
                q[int(rng.integers(0, length))] = int(rng.choice(list(b"ACGT")))
            reads.append(bytes(q))
    b = lib.Batch(ix)
    b.set_options(sw_band_min=0)
    try:
        b.set_reads(*lib.pack_reads(reads))
        c = b.run(lib.STAGE_EXTEND)
        out = b.fetch_hits()
        assert check_against_oracle(out, reads, ofm, ref1) > 200
        assert c["n_banded"] > 100, c
        assert int(np.max(out["win_len"])) > 254 + 40          # clusters that chain along a tract: windows past the packed kernels' first cap
        # ... which the packed kernel's wide pass takes (sw_pk = 2: the generic lane-per-alignment kernel, as before); and no band
        # sw_wave = 0: a DP queue this short would otherwise be taken whole by the wave-per-alignment kernel, and the packed
        # kernel's passes -- the wide one above all -- would launch on empty queues: without it they get every item
        for opts in ({"sw_band_min": 0, "sw_pk": 2}, {"sw_band_min": 0, "sw_band": 0}, {"sw_band_min": 0, "sw_band": 0, "sw_pk": 2},
                     {"sw_band_min": 0, "sw_wave": 0}, {"sw_band_min": 0, "sw_wave": 0, "sw_band": 0}, {"sw_wave": 0, "sw_band": 0, "sw_pk": 2}, {"sw_band_min": 0, "sw_band12": 0},
                     {"sw_wave": 0, "sw_band": 0, "sw_trace_all": 0}):
            b.set_options(**opts)
            b.run(lib.STAGE_EXTEND)
            out0 = b.fetch_hits()
            for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
                assert np.array_equal(out[k], out0[k]), (opts, k)
            assert out["cigar"] == out0["cigar"], opts
    finally:
        b.destroy()
