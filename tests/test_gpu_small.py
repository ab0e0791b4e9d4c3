"""The small-batch path (bwt_small_align: the reference's message sizes, src/master.erl:59 -> src/worker_bwt.erl:62-69) gives the
general path's tables and the oracle's answers -- at 1, 7, 200, 2000 and 4096 reads, ragged reads, N / IUPAC reads, with every
kind of suffix-interval table, through its overflow fall-back, and as the worker mirror's whole-workload call (cast order)."""
import numpy as np
import pytest

from conftest import FM_PATH, REF_PATH, synth_reads

pytestmark = pytest.mark.gpu

KEYS = ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off")


def same_tables(a, b):
    for k in KEYS:
        assert np.array_equal(a[k], b[k]), k
    assert bytes(a["cigar"]) == bytes(b["cigar"])


def oracle_check(out, reads, ofm, ref1):
    from oracle import orc
    nh = 0
    for i, q in enumerate(reads):
        try:
            exp = orc.align_read(ofm, ref1, q)
        except RuntimeError:
            assert int(out["read_status"][i]) < 0, i
            continue
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(), int(out["pos"][h]),
                int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
        assert got == ([] if exp is None else exp), (i, q)
        nh += len(got)
    return nh


def mixed_reads(ref, n, seed, length=100):
    return (synth_reads(ref, n - n // 2, length, seed, max_subs=2) +
            synth_reads(ref, n // 2, length, seed + 1, max_subs=2, n_frac=0.1, iupac_frac=0.05, indel_frac=0.3))


@pytest.mark.parametrize("n", [1, 7, 200, 2000, 4096])
def test_small_path_equals_general_path_and_oracle(gpu_index, orc_fm, ref_bytes, n):
    from bwt_amd import lib
    from oracle import erl_literal as E
    reads = mixed_reads(ref_bytes, n, seed=100 + n)
    bases, off = lib.pack_reads(reads)
    sb = lib.SmallBatch(gpu_index)
    try:
        for rep in range(3):                                       # the handle is reused: nothing of one call leaks into the next
            out = sb.align(bases, off)
            assert out["path"] == 1, (out["path"], out["overflow_flags"])
            if rep == 0:
                gen = gpu_index.align_batch(bases, off)
            same_tables(out, gen)
        assert out["counts"]["n_hits"] == len(gen["score"]) and out["counts"]["n_float"] > (0 if n >= 200 else -1)
        if n <= 2000:
            assert oracle_check(out, reads, orc_fm, E.make_ref1(ref_bytes)) > (0 if n >= 7 else -1)
        # a different batch on the same handle, smaller: stale state of the larger one must not show
        reads2 = mixed_reads(ref_bytes, max(1, n // 3), seed=7 + n)
        b2, o2 = lib.pack_reads(reads2)
        same_tables(sb.align(b2, o2), gpu_index.align_batch(b2, o2))
    finally:
        sb.close()


def test_ragged_empty_and_bad_reads(gpu_index, orc_fm, ref_bytes):
    from bwt_amd import lib
    from oracle import erl_literal as E
    rng = np.random.default_rng(5)
    reads = []
    for i in range(300):
        L = int(rng.integers(0, 190))
        st = int(rng.integers(0, len(ref_bytes) - 200))
        reads.append(ref_bytes[st:st + L])
    reads += [b"", b"ACGT", b"ACGTACGTX", b"N" * 50, b"A" * 120, ref_bytes[1000:1100] + b"X"]
    bases, off = lib.pack_reads(reads)
    sb = lib.SmallBatch(gpu_index)
    try:
        out = sb.align(bases, off)
        assert out["path"] == 1
        same_tables(out, gpu_index.align_batch(bases, off))
        oracle_check(out, reads, orc_fm, E.make_ref1(ref_bytes))
        # offsets that do not start at 0
        sub = sb.align(bases, off[100:201])
        same_tables(sub, gpu_index.align_batch(bases, off[100:201]))
        # no reads at all, more reads than the path takes (general path inside the same call), reads too long for it
        empty = sb.align(bases[:0], np.zeros(1, np.uint64))
        assert empty["path"] == 0 and len(empty["score"]) == 0
        many = mixed_reads(ref_bytes, 5000, seed=3)
        bm, om = lib.pack_reads(many)
        big = sb.align(bm, om)
        assert big["path"] == 0
        same_tables(big, gpu_index.align_batch(bm, om))
        longr = [ref_bytes[5000:5400], ref_bytes[9000:9300]]
        bl, ol = lib.pack_reads(longr)
        lo = sb.align(bl, ol)
        assert lo["path"] == 0
        same_tables(lo, gpu_index.align_batch(bl, ol))
        with pytest.raises(lib.BwtError):
            sb.align(bases, np.array([0, 50, 20], np.uint64))
        same_tables(sb.align(bases, off), out)                     # the handle is still good
    finally:
        sb.close()


@pytest.fixture(scope="module")
def repeat_index():
    """1.5 Mbp i.i.d. genome with a 300-bp segment in 40 copies; the full-depth wide table (k_seed_fast's precondition)"""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(77)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_500_000, dtype=np.uint8)].tobytes())
    seg = bytes(g[5000:5300])
    for k in range(1, 40):
        g[30_000 * k:30_000 * k + 300] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    L, SA = ix.export()
    ptr, shift = ix.meta()
    ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
    yield ix, ref, seg, ofm, E.make_ref1(ref)
    ix.close()


@pytest.mark.parametrize("K,wide", [(15, 2), (15, 1), (15, 0), (12, 0), (0, 0)])
def test_every_table_shape(repeat_index, K, wide):
    """K = 15 wide: k_seed_fast + in-kernel clustering; 15 narrow / 12: table + SA gathers (+ LF steps); 0: LF steps only"""
    from bwt_amd import lib
    ix, ref, seg, ofm, ref1 = repeat_index
    ix.set_ktable(K, wide)
    reads = mixed_reads(ref, 600, seed=31) + [seg[i:i + 100] for i in range(0, 200, 7)]
    bases, off = lib.pack_reads(reads)
    sb = lib.SmallBatch(ix)
    try:
        out = sb.align(bases, off)
        assert out["path"] == 1
        same_tables(out, ix.align_batch(bases, off))
        assert oracle_check(out, reads, ofm, ref1) > 600
        # 150 bp reads (18 sub-reads: the two-register seed kernel with the wide table)
        r150 = mixed_reads(ref, 300, seed=32, length=150)
        b150, o150 = lib.pack_reads(r150)
        o = sb.align(b150, o150)
        assert o["path"] == 1
        same_tables(o, ix.align_batch(b150, o150))
    finally:
        sb.close()
        ix.set_ktable(-1, -1)


def test_capacity_overflow_falls_back_to_the_general_path(repeat_index):
    """2000 reads from a segment with 40 copies: ~80 000 candidates, more than the path's 32768 -> the same call redoes the batch
    through the general path (path 2) and the tables are the general path's"""
    from bwt_amd import lib
    ix, ref, seg, ofm, ref1 = repeat_index
    rng = np.random.default_rng(9)
    reads = [seg[s:s + 100] for s in rng.integers(0, 200, 2000)]
    bases, off = lib.pack_reads(reads)
    sb = lib.SmallBatch(ix)
    try:
        out = sb.align(bases, off)
        assert out["path"] == 2 and sb.fallbacks == 1
        gen = ix.align_batch(bases, off)
        assert len(gen["score"]) > 40_000
        same_tables(out, gen)
        few = sb.align(*lib.pack_reads(reads[:100]))               # ... and the next small batch is a small batch again
        assert few["path"] == 1
        same_tables(few, ix.align_batch(*lib.pack_reads(reads[:100])))
    finally:
        sb.close()


def test_options_that_the_small_path_does_not_take(gpu_index, ref_bytes):
    from bwt_amd import lib
    reads = mixed_reads(ref_bytes, 300, seed=2)
    bases, off = lib.pack_reads(reads)
    gen = gpu_index.align_batch(bases, off)
    sb = lib.SmallBatch(gpu_index)
    try:
        for opts, path in (({"sw_prefilter": 0}, 0), ({"sw_wave": 0}, 0), ({"seed_fuse": 0}, 1), ({"seed_fast": 0}, 1), ({}, 1)):
            sb.set_options(**opts)
            out = sb.align(bases, off)
            assert out["path"] == path, (opts, out["path"])
            same_tables(out, gen)
    finally:
        sb.close()


def test_worker_mirror_takes_a_whole_workload_message(gpu_index, ref_bytes):
    """worker_bwt.workload: the ten QseqLists of one {workload, ...} message in ONE call, casts in the reference's order -- list by
    list in message order (src/worker_bwt.erl:73,112), inside a list in reverse read order (:74-80) -- i.e. exactly what one
    worker_bwt.running per list gives"""
    from bwt_amd import fm_index, worker_bwt
    fm = fm_index.FM(gpu_index)
    reads = mixed_reads(ref_bytes, 10 * 200, seed=44)
    lists = [[(("r%d" % (200 * k + i)).encode(), reads[200 * k + i]) for i in range(200)] for k in range(10)]
    lists[3] = lists[3][:57]                                       # the last list of a file is short
    lists.append([])
    one_call = worker_bwt.workload(fm, lists)
    per_list = [c for ql in lists for c in worker_bwt.running(fm, ql)]
    assert len(one_call) > 500                                      # (about half of the reads of this contig end without a cluster: DESIGN 2)

    def norm(c):
        return c if isinstance(c, str) else (c[0], c[1], str(c[2])) if c[0] == "error" else c
    assert [norm(c) for c in one_call] == [norm(c) for c in per_list]
    assert fm._small_batch.paths[1] >= 11 and fm._small_batch.paths[2] == 0
    fm._small_batch.close()
    fm._small_batch = None                                         # (the index belongs to the session fixture: not closed here)


def test_widened_lower_levels_every_read_length(repeat_index):
    """set_ktable(15, 2): the lower levels of the suffix-interval table where k-mers mostly have a seed list are kept as 64-byte
    entries too, so that a read's short last sub-read (8 + Lq mod 8 bases) gets its seeds with its table line.  Read lengths whose
    last sub-read ends at every level 9..15, against the table-less search (K = 0: the mode the oracle tests pin) and the oracle,
    through both the general and the small-batch path."""
    from bwt_amd import lib
    ix, ref, seg, ofm, ref1 = repeat_index
    ix.set_ktable(15, 2)
    try:
        levels = ix.ktable_wide_levels
        assert ix.ktable == (15, True) and len(levels) >= 2, levels          # (1.5 M positions: the 10- and 11-mers mostly occur once or twice)
        batches = {}
        for length in (81, 82, 83, 84, 85, 86, 87, 88, 100, 150):
            reads = mixed_reads(ref, 400, seed=length, length=length) + [seg[i:i + length] for i in range(0, 300 - length, 11)]
            bases, off = lib.pack_reads(reads)
            out = ix.align_batch(bases, off)
            sb = lib.SmallBatch(ix)
            got = sb.align(bases, off)
            assert got["path"] == 1
            sb.close()
            same_tables(got, out)
            batches[length] = (reads, bases, off, out)
        assert oracle_check(batches[83][3], batches[83][0], ofm, ref1) > 300 and oracle_check(batches[100][3], batches[100][0], ofm, ref1) > 300
        ix.set_ktable(0, 0)
        for length, (reads, bases, off, out) in batches.items():
            same_tables(ix.align_batch(bases, off), out)
    finally:
        ix.set_ktable(-1, -1)


def test_batches_of_one_shape_replay_a_graph(gpu_index, ref_bytes):
    """The second batch in a row of one shape (read count, lengths, options) captures the launches of the small path into a hipGraph and
    the ones after it replay it; the answers stay the general path's for every batch, a batch of another shape is launched kernel by
    kernel again, and so is everything under small_graph = 0, while profiling, and after an option changes."""
    from bwt_amd import lib
    sb = lib.SmallBatch(gpu_index)
    try:
        seen = []
        for k in range(5):
            reads = mixed_reads(ref_bytes, 200, seed=300 + k)           # the same shape, other reads each time
            bases, off = lib.pack_reads(reads)
            out = sb.align(bases, off)
            assert out["path"] == 1
            seen.append(out["graph"])
            same_tables(out, gpu_index.align_batch(bases, off))
        assert seen == [0, 2, 1, 1, 1], seen
        other = mixed_reads(ref_bytes, 150, seed=9)
        bo, oo = lib.pack_reads(other)
        o = sb.align(bo, oo)
        assert o["graph"] == 0
        same_tables(o, gpu_index.align_batch(bo, oo))
        for k in range(3):                                              # back to the first shape: captured again on the second in a row
            reads = mixed_reads(ref_bytes, 200, seed=400 + k)
            bases, off = lib.pack_reads(reads)
            out = sb.align(bases, off)
            assert out["graph"] == (0, 2, 1)[k]
            same_tables(out, gpu_index.align_batch(bases, off))
        # 2000 reads (the two-pass DP launches), 150 bp reads (18 sub-reads per read)
        for n, length in ((2000, 100), (300, 150)):
            got = []
            for k in range(3):
                reads = mixed_reads(ref_bytes, n, seed=500 + k, length=length)
                bases, off = lib.pack_reads(reads)
                out = sb.align(bases, off)
                got.append(out["graph"])
                same_tables(out, gpu_index.align_batch(bases, off))
            assert got == [0, 2, 1], (n, got)
        sb.set_options(seed_fast=0)                                     # an option changes what is launched: not the old graph
        reads = mixed_reads(ref_bytes, 300, seed=77, length=150)
        bases, off = lib.pack_reads(reads)
        out = sb.align(bases, off)
        assert out["graph"] == 0
        same_tables(out, gpu_index.align_batch(bases, off))
        sb.set_options(small_graph=0)
        for k in range(3):
            assert sb.align(bases, off)["graph"] == 0
        sb.set_options()
        sb.profile(True)
        for k in range(3):
            out = sb.align(bases, off)
            assert out["graph"] == 0
        sb.profile(False)
        assert [sb.align(bases, off)["graph"] for _ in range(2)] == [2, 1]
        same_tables(sb.align(bases, off), gpu_index.align_batch(bases, off))
    finally:
        sb.close()


def test_two_handles_on_one_index_from_two_threads(gpu_index, ref_bytes):
    """several bwt_small handles may share an index (include/bwt_gpu.h): two threads, a handle each, batches of one shape each -- both
    capture and replay their own graph while the other runs, and every table is the general path's"""
    import threading
    from bwt_amd import lib
    jobs = []
    for t, (n, length) in enumerate(((200, 100), (333, 120))):
        batches = []
        for k in range(12):
            bases, off = lib.pack_reads(mixed_reads(ref_bytes, n, seed=900 + 50 * t + k, length=length))
            batches.append((bases, off, gpu_index.align_batch(bases, off)))
        jobs.append(batches)
    errors, graphs = [], [None, None]

    def work(t):
        sb = lib.SmallBatch(gpu_index)
        try:
            for rep in range(3):
                for bases, off, want in jobs[t]:
                    out = sb.align(bases, off)
                    assert out["path"] == 1
                    same_tables(out, want)
            graphs[t] = dict(sb.graphs)
        except Exception as e:                                     # noqa: BLE001 (reported below, from the main thread)
            errors.append((t, repr(e)))
        finally:
            sb.close()

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errors, errors
    assert all(g[1] >= 30 and g[2] == 1 for g in graphs), graphs      # one capture, then replays
