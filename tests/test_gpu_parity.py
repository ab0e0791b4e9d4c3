"""GPU parity tests proper: the HIP path, called through the C ABI (bwt_amd.lib -> libbwtgpu.so),
against the CPU oracle (oracle/bwt_oracle.c, oracle/erl_literal.py) on the same seeded inputs.
Bit-exact: SA ranges, seed positions, clusters, scores, CIGAR strings, positions."""
import numpy as np
import pytest

from conftest import synth_reads

pytestmark = pytest.mark.gpu


def subreads_of(q):
    """The sub-read chain of sga:sga/9 (src/sga.erl:18-34) via the literal oracle's skip_Ns."""
    from oracle import erl_literal as E
    out = []
    while True:
        r = E.skip_Ns(q)
        if r is None:
            return out
        skip, q1 = r
        out.append(q1)
        q = q1[:len(q1) - E.MIN_LEN]


def mixed_reads(ref, n, seed):
    return (synth_reads(ref, n // 2, 100, seed) +
            synth_reads(ref, n // 4, 100, seed + 1, n_frac=0.6) +
            synth_reads(ref, n // 8, 100, seed + 2, iupac_frac=0.8, indel_frac=0.5) +
            synth_reads(ref, n - n // 2 - n // 4 - n // 8, 150, seed + 3, indel_frac=0.3))


@pytest.fixture
def ktable_depth(gpu_index):
    """Sets the suffix-interval table depth for one test and puts the default back."""
    def set_k(k):
        wide = isinstance(k, str)
        gpu_index.set_ktable(int(k[:-1]) if wide else k, 1 if wide else 0)
        assert gpu_index.ktable == ((15, True) if wide else (k, False))
    yield set_k
    gpu_index.set_ktable(-1)


# 0: every task step by step; 8: smallest table; 9/12: table + remaining steps; 15: the table answers everything;
# "15w": ... and its top level carries the SA values (the seed kernel clusters the reads itself)
@pytest.mark.parametrize("K", [0, 8, 9, 12, 15, "15w"])
def test_search_tasks_match_oracle(gpu_index, orc_fm, ref_bytes, ktable_depth, K):
    from bwt_amd import lib
    from oracle import orc
    ktable_depth(K)
    reads = mixed_reads(ref_bytes, 400, 11)
    bases, off = lib.pack_reads(reads)
    b = lib.Batch(gpu_index)
    b.set_reads(bases, off)
    counts = b.run(lib.STAGE_SEARCH)
    t = b.fetch_tasks()
    b.destroy()
    k = 0
    steps = 0
    for ri, q in enumerate(reads):
        for sub in subreads_of(q):
            st, seeds, sp, ep, n = orc.find_seeds(orc_fm, sub)
            assert t["read"][k] == ri and t["end"][k] == len(sub)
            assert t["status"][k] == st, (ri, sub)
            if st >= 0:
                assert (t["sp"][k], t["ep"][k], t["nmatch"][k]) == (sp, ep, n), (ri, sub)
                steps += n - 1
            k += 1
    assert k == len(t["status"]) == counts["n_tasks"]
    if K == 0:
        assert counts["lf_steps"] >= steps and counts["n_lookups"] == 0   # steps that ended in not_found are counted too
    else:
        depth = 15 if K == "15w" else K
        assert counts["lf_steps"] <= max(0, 15 - depth) * counts["n_tasks"] + 14 * (counts["n_tasks"] - counts["n_lookups"])
        assert 0 < counts["n_lookups"] <= counts["n_tasks"]
    if K in (15, "15w"):
        assert counts["lf_steps"] <= 14 * (counts["n_tasks"] - counts["n_lookups"])   # only tasks the table cannot take


@pytest.mark.parametrize("K", [0, 10, 15, "15w"])
def test_find_seeds_batch_lists(gpu_index, orc_fm, ref_bytes, ktable_depth, K):
    from bwt_amd import lib
    from oracle import orc
    ktable_depth(K)
    subs = []
    for q in mixed_reads(ref_bytes, 120, 21):
        subs += subreads_of(q)
    subs += [b"ACGTACGTAX", b"A", b"N", b"ACGTRACGTACGT"]   # crash / single base / IUPAC inside
    bases, off = lib.pack_reads(subs)
    out = gpu_index.find_seeds_batch(bases, off)
    for i, sub in enumerate(subs):
        st, seeds, sp, ep, n = orc.find_seeds(orc_fm, sub)
        assert out["status"][i] == st, sub
        got = out["seeds"][int(out["seed_off"][i]):int(out["seed_off"][i + 1])].tolist()
        assert got == seeds, sub


@pytest.mark.parametrize("K", [9, "15w"])
def test_sga_batch_clusters(gpu_index, orc_fm, ref_bytes, ktable_depth, K):
    from bwt_amd import lib
    from oracle import orc
    ktable_depth(K)
    reads = mixed_reads(ref_bytes, 600, 31) + [b"ACGT", b"", b"N" * 100, b"ACGTACGTACGTACGTACGT"]
    bases, off = lib.pack_reads(reads)
    out = gpu_index.sga_batch(bases, off)
    nbig = ncrash = 0
    for i, q in enumerate(reads):
        a, b = int(out["clus_off"][i]), int(out["clus_off"][i + 1])
        try:
            exp = orc.sga(orc_fm, q)
        except RuntimeError:
            # the reference crashes (case_clause, src/bwa.erl:15-24: a sub-read ends in a base outside
            # ACGTN); the C ABI reports that read instead (the documented divergence)
            assert out["read_status"][i] == -1 and a == b
            ncrash += 1
            continue
        assert out["read_status"][i] == 0
        got = list(zip(out["S"][a:b].tolist(), out["D"][a:b].tolist()))
        assert got == exp, (i, q)
        nbig += len(exp) > 4
    assert nbig > 0          # the repeat-rich contig must exercise multi-cluster reads
    assert ncrash > 0


def rand_pairs(n, seed):
    rng = np.random.default_rng(seed)
    qs, rs = [], []
    alpha = list(b"ACGT")
    for i in range(n):
        lq = int(rng.integers(1, 140))
        base = bytes(rng.choice(alpha, lq + 40).tolist())
        q = bytearray(base[10:10 + lq])
        r = bytearray(base[:lq + int(rng.integers(0, 30))])
        for _ in range(int(rng.integers(0, 4))):
            q[int(rng.integers(0, lq))] = int(rng.choice(alpha))
        if lq > 20 and rng.random() < 0.4:
            p = int(rng.integers(5, lq - 5)); k = int(rng.integers(1, 4))
            if rng.random() < 0.5:
                del q[p:p + k]
            else:
                q[p:p] = bytes(rng.choice(alpha, k).tolist())
        if i % 4 == 1:
            q[int(rng.integers(0, len(q)))] = ord("N")
        if i % 4 == 2:
            r[int(rng.integers(0, len(r)))] = int(rng.choice(list(b"NRYKMSWBDVH")))
        if i % 16 == 3:
            r = bytearray(rng.choice(alpha, len(r)).tolist())     # unrelated -> no_match
        qs.append(bytes(q)); rs.append(bytes(r))
    return qs, rs


def test_sw_batch_matches_oracle():
    from bwt_amd import lib
    from oracle import orc
    qs, rs = rand_pairs(700, 5)
    kats = [(b"ACGT", b"ACGT"), (b"ACGT", b"TTACGTTT"), (b"ACGTACGT", b"ACGTTACGT"), (b"ACGTTACGT", b"ACGTACGT"),
            (b"ACGTACGTAC", b"ACGAACGTAC"), (b"AAAA", b"CCCC"), (b"ACGTNACGT", b"ACGTAACGT"),
            (b"ACGTACGTACGT", b"ACGTACCCGTACGT"), (b"ACGTACCCGTACGT", b"ACGTACGTACGT"), (b"A", b"A"), (b"A", b"C")]
    qs += [k[0] for k in kats]; rs += [k[1] for k in kats]
    q, qoff = lib.pack_reads(qs)
    r, roff = lib.pack_reads(rs)
    out = lib.sw_batch(q, qoff, r, roff)
    nm = 0
    for i in range(len(qs)):
        exp = orc.sw(qs[i], rs[i])
        if exp == "no_match":
            assert out["status"][i] == 1, (qs[i], rs[i])
            nm += 1
        else:
            cig = out["cigar"][int(out["cig_off"][i]):int(out["cig_off"][i + 1])].decode()
            assert (out["status"][i], int(out["score"][i]), cig) == (0, exp[0], exp[1]), (qs[i], rs[i])
    assert 0 < nm < len(qs)


def test_sw_long_cigar_overflow_path():
    """many short runs force the CIGAR slot to be widened (stride retry)"""
    from bwt_amd import lib
    from oracle import orc
    unit_q, unit_r = b"ACGTTGCAAC", b"ACGTTGCAACG"       # one deletion per unit
    q = unit_q * 14
    r = unit_r * 14
    out = lib.sw_batch(*lib.pack_reads([q]), *lib.pack_reads([r]))
    exp = orc.sw(q, r)
    cig = out["cigar"].decode()
    assert exp != "no_match" and len(exp[1]) > 32
    assert (int(out["score"][0]), cig) == exp


@pytest.mark.parametrize("K", [0, 9, 15, "15w"])
def test_align_batch_matches_oracle(gpu_index, orc_fm, ref_bytes, ktable_depth, K):
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ktable_depth(K)
    ref1 = E.make_ref1(ref_bytes)
    reads = mixed_reads(ref_bytes, 500, 41)
    # reads hanging over both contig ends: windows reach into the 200-N halo (fp64 path)
    reads += [ref_bytes[:100], ref_bytes[-100:], ref_bytes[3:103], ref_bytes[-120:-20]]
    bases, off = lib.pack_reads(reads)
    out = gpu_index.align_batch(bases, off)
    nfloat = 0
    for i, q in enumerate(reads):
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        try:
            exp = orc.align_read(orc_fm, ref1, q)
        except RuntimeError:
            assert out["read_status"][i] == -1 and a == b
            continue
        if exp is None:
            assert out["read_status"][i] == 1 and a == b
            continue
        assert out["read_status"][i] == 0
        got = []
        for h in range(a, b):
            cig = out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode()
            got.append((int(out["score"][h]), cig, int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])))
        assert got == exp, (i, q)
        nfloat += b"N" in q
    assert nfloat > 0
    # RefSeq of a hit is the Ref1 slice
    if len(out["score"]):
        w = gpu_index.ref1(int(out["win_off"][0]), int(out["win_len"][0]))
        assert w == ref1[int(out["win_off"][0]):int(out["win_off"][0]) + int(out["win_len"][0])]


def test_counters_and_profile(gpu_index, ref_bytes):
    from bwt_amd import lib
    reads = synth_reads(ref_bytes, 2000, 100, 77)
    bases, off = lib.pack_reads(reads)
    b = lib.Batch(gpu_index)
    b.set_reads(bases, off)
    b.profile(True)
    c = b.run(lib.STAGE_EXTEND)
    kt = b.kernel_times()
    b.destroy()
    assert c["n_reads"] == 2000 and c["n_tasks"] == 2000 * 12
    assert c["lf_steps"] + 14 * c["n_lookups"] > 2000 * 60 and c["n_hits"] > 500
    assert "fm_search" in kt and kt["fm_search"]["launches"] == 1 and kt["fm_search"]["ms"] > 0
    assert "sw_extend" in kt


def test_bench_gather_path_single_rank():
    """bench.py's N>1 exchange step (hits_to_device + RCCL gather of 16-byte records) rehearsed with a
    world of one on the GPU: every hit of the step arrives on rank 0."""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, BWT_BENCH_FORCE_GATHER="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "config2-align", "--reads", "20000",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], env=env, capture_output=True, text=True,
                         timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads(out.stdout.strip().splitlines()[-1])
    c = line["counts_per_step"]
    assert c["gathered_hits"] == c["n_hits"] > 1000
    assert line["n_gpus"] == 1 and line["roofline"]["frac"] > 0


@pytest.mark.parametrize("wave_dp", ["1", "0"])          # short DP queues: wave-per-alignment kernel / packed kernels
@pytest.mark.parametrize("length", [100, 150])
def test_align_uniform_length_fast_path(gpu_index, orc_fm, ref_bytes, length, wave_dp):
    """Batches of one read length take the packed-i16 extend kernels (k_sw_pk<100|150>) or, for a short DP queue,
    the wave-per-alignment kernel (k_sw_wave): substitutions are settled by the prefilter, indels run the traced DP,
    N goes to the fp64 pass."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    reads = (synth_reads(ref_bytes, 1500, length, 61 + length, max_subs=3) +
             synth_reads(ref_bytes, 700, length, 62 + length, indel_frac=1.0) +
             synth_reads(ref_bytes, 200, length, 63 + length, n_frac=1.0))
    bases, off = lib.pack_reads(reads)
    b = lib.Batch(gpu_index)
    b.set_options(sw_wave=int(wave_dp))
    b.set_reads(bases, off)
    counts = b.run(lib.STAGE_EXTEND)
    out = b.fetch_hits()
    b.destroy()
    assert counts["n_traced"] > 50 and counts["n_float"] > 20 and counts["n_hits"] > 1000
    for i, q in enumerate(reads):
        a, bb = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        exp = orc.align_read(orc_fm, ref1, q)
        if exp is None:
            assert out["read_status"][i] == 1 and a == bb
            continue
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, bb)]
        assert got == exp, (i, q)


def test_parity_on_gpu_built_synthetic_index():
    """A 6 M-row i.i.d. index built by the GPU builder (the bench's kind of index): search, clustering and
    extension against the oracle run on the exported columns."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(17)
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 6_000_000, dtype=np.uint8)].tobytes()
    ix = lib.Index.build(ref, shift=7)
    L, SA = ix.export()
    ptr, shift = ix.meta()
    assert shift == 7
    ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
    ref1 = E.make_ref1(ref)
    reads = synth_reads(ref, 1500, 100, 71, max_subs=2) + synth_reads(ref, 300, 100, 72, indel_frac=0.7, n_frac=0.2)
    bases, off = lib.pack_reads(reads)
    out = ix.align_batch(bases, off)
    nh = 0
    for i, q in enumerate(reads):
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        exp = orc.align_read(ofm, ref1, q)
        if exp is None:
            assert out["read_status"][i] == 1 and a == b
            continue
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
        assert got == exp, (i, q)
        nh += len(got)
    assert nh > 400
    ix.close()


def test_repeat_genome_all_cluster_paths():
    """A genome with a 400-bp segment copied to 3, 6 and 12 places: reads from the copies have several clusters
    and 36..150 seeds, so every cluster path runs -- the seed kernel's own clustering (wide table), its fall-back
    when a read has more clusters than its staging row, and the wave / workgroup kernels for > 64 seeds.  The
    same reads through every table mode must agree with the oracle."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    rng = np.random.default_rng(29)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 400_000, dtype=np.uint8)].tobytes())
    segs = []
    for copies, at in ((3, 10_000), (6, 60_000), (12, 150_000)):
        seg = bytes(g[at:at + 400])
        segs.append((seg, copies))
        for c in range(1, copies):
            p = at + c * 20_011
            g[p:p + 400] = seg
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    L, SA = ix.export()
    ptr, shift = ix.meta()
    ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
    ref1 = E.make_ref1(ref)
    reads = []
    for seg, copies in segs:
        for s0 in range(0, 300, 37):
            q = bytearray(seg[s0:s0 + 100])
            reads.append(bytes(q))
            q[50] = b"ACGT"[(b"ACGT".index(q[50]) + 1) & 3]
            reads.append(bytes(q))
    reads += synth_reads(ref, 300, 100, 5, max_subs=2)
    bases, off = lib.pack_reads(reads)
    exp = [orc.align_read(ofm, ref1, q) for q in reads]
    assert max(len(e) for e in exp if e) >= 6
    for mode in ((0, 0), (12, 0), (15, 0), (15, 1), (15, 2)):
        ix.set_ktable(*mode)
        out = ix.align_batch(bases, off)
        for i, q in enumerate(reads):
            a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
            if exp[i] is None:
                assert out["read_status"][i] == 1 and a == b, (mode, i)
                continue
            got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                    int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
            assert got == exp[i], (mode, i, q)
    ix.close()


@pytest.mark.parametrize("length", [16, 40, 64, 72, 96, 128, 200, 256, 511])
def test_wide_table_read_lengths(gpu_index, orc_fm, ref_bytes, ktable_depth, length):
    """The fused seed + cluster kernel packs 64 / maxT reads into a wave pass (maxT = length div 8 sub-reads per read):
    8 reads of 64 bp down to one read of 511 bp, and 65..128 seeds per read for the longer ones; reads shorter than
    64 bp (fewer than 8 sub-reads) take the table but the separate cluster kernels."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ktable_depth("15w")
    ref1 = E.make_ref1(ref_bytes)
    reads = synth_reads(ref_bytes, 60, length, 900 + length, max_subs=2) + \
        synth_reads(ref_bytes, 20, length, 950 + length, n_frac=0.5, indel_frac=0.3 if length >= 32 else 0.0)
    bases, off = lib.pack_reads(reads)
    out = gpu_index.align_batch(bases, off)
    for i, q in enumerate(reads):
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        try:
            exp = orc.align_read(orc_fm, ref1, q)
        except RuntimeError:
            assert out["read_status"][i] == -1 and a == b
            continue
        if exp is None:
            assert out["read_status"][i] == 1 and a == b, (length, i)
            continue
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
        assert got == exp, (length, i, q)


def test_table_modes_agree_at_scale():
    """The suffix-interval table is a cache of LF steps: whatever its depth, and whether or not its top level
    carries SA values (the seed kernel then clusters the reads itself), every output of a batch must be the same.
    20 M-row index, 300 000 reads of both lengths with substitutions, indels and N -- too many for the CPU oracle,
    so the modes are held against each other (the step-by-step mode K = 0 is the one the oracle tests pin)."""
    from bwt_amd import lib
    rng = np.random.default_rng(41)
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 20_000_000, dtype=np.uint8)].tobytes()
    ix = lib.Index.build(ref, shift=1)
    for length, nreads, seed in ((100, 200_000, 3), (150, 100_000, 4)):
        reads = (synth_reads(ref, nreads // 2, length, seed, max_subs=2) +
                 synth_reads(ref, nreads // 4, length, seed + 10, indel_frac=0.6, max_subs=2) +
                 synth_reads(ref, nreads // 4, length, seed + 20, n_frac=0.3, iupac_frac=0.05))
        bases, off = lib.pack_reads(reads)
        base = None
        for mode in ((0, 0), (13, 0), (12, 1), (15, 0), (15, 1), (15, 2)):
            ix.set_ktable(*mode)
            out = ix.align_batch(bases, off)
            if base is None:
                base = out
                assert int(out["hit_off"][-1]) > nreads // 3
                continue
            for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
                assert np.array_equal(np.asarray(out[k]), np.asarray(base[k])), (mode, length, k)
            assert bytes(out["cigar"]) == bytes(base["cigar"]), (mode, length)
    ix.close()


def test_table_modes_agree_on_repeats():
    """Same property on a genome full of repeats (copies of segments, tandem repeats, homopolymer runs): reads with
    hundreds of seeds and many clusters, i.e. every fall-back out of the fused seed kernel (more clusters than its
    staging row, more than 128 seeds, ranges that do not come with the table entry) under the lean single-call run."""
    from bwt_amd import lib
    rng = np.random.default_rng(77)
    g = bytearray(np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, 1_500_000, dtype=np.uint8)].tobytes())
    for _ in range(300):                                   # copies of 200..600 bp segments, 2..15 times
        L = int(rng.integers(200, 600)); src = int(rng.integers(0, len(g) - L))
        seg = bytes(g[src:src + L])
        for _ in range(int(rng.integers(2, 16))):
            d = int(rng.integers(0, len(g) - L)); g[d:d + L] = seg
    for _ in range(200):                                   # tandem repeats and homopolymers
        L = int(rng.integers(50, 300)); d = int(rng.integers(0, len(g) - L))
        unit = bytes(rng.choice(list(b"ACGT"), int(rng.integers(1, 7))).tolist())
        g[d:d + L] = (unit * (L // len(unit) + 1))[:L]
    ref = bytes(g)
    ix = lib.Index.build(ref, shift=1)
    for length in (100, 150):
        reads = synth_reads(ref, 30_000, length, 5 + length, max_subs=2) + \
            synth_reads(ref, 10_000, length, 6 + length, indel_frac=0.5, n_frac=0.1)
        bases, off = lib.pack_reads(reads)
        base = None
        for mode in ((0, 0), (12, 0), (15, 0), (15, 1), (15, 2)):
            ix.set_ktable(*mode)
            out = ix.align_batch(bases, off)
            if base is None:
                base = out
                continue
            for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
                assert np.array_equal(np.asarray(out[k]), np.asarray(base[k])), (mode, length, k)
            assert bytes(out["cigar"]) == bytes(base["cigar"]), (mode, length)
        nh = np.diff(np.asarray(base["hit_off"]).astype(np.int64))
        assert (nh > 4).sum() > 100                       # reads with more hits than the staging row holds
    ix.close()


def test_prefilter_shortcuts_on_low_complexity_pairs():
    """The extend stage skips the DP when (a) the read matches a window diagonal exactly, or (b) the window
    is square (W = Lq) and its diagonal has at most one mismatch.  Adversarial inputs for both claims:
    homopolymers, short tandem repeats, shifted repeats, and 0/1/2 mismatches at every kind of position."""
    from bwt_amd import lib
    from oracle import orc
    rng = np.random.default_rng(123)
    qs, rs = [], []
    units = [b"A", b"AC", b"ACG", b"AAC", b"ACGT", b"AACCGGTT", b"ATATATGC"]
    for lq in (8, 9, 31, 32, 33, 64, 100, 150):
        for u in units:
            base = (u * (lq // len(u) + 3))
            for shift in (0, 1, len(u)):
                for nm in (0, 1, 2):
                    q = bytearray(base[shift:shift + lq])
                    r = bytearray(base[:lq])                       # square window
                    for _ in range(nm):
                        p = int(rng.integers(0, lq))
                        q[p] = b"ACGT"[(b"ACGT".index(q[p]) + 1 + int(rng.integers(0, 3))) & 3]
                    qs.append(bytes(q)); rs.append(bytes(r))
                    r2 = bytearray(base[:lq + int(rng.integers(1, 12))])   # wider window (D > 0)
                    qs.append(bytes(q)); rs.append(bytes(r2))
        for _ in range(40):                                       # random sequences, mismatch at the ends
            r = bytes(rng.choice(list(b"ACGT"), lq).tolist())
            q = bytearray(r)
            for p in {0: [], 1: [0], 2: [lq - 1], 3: [0, lq - 1]}[int(rng.integers(0, 4))]:
                q[p] = b"ACGT"[(b"ACGT".index(q[p]) + 1) & 3]
            qs.append(bytes(q)); rs.append(r)
    out = lib.sw_batch(*lib.pack_reads(qs), *lib.pack_reads(rs))
    nshort = 0
    for i in range(len(qs)):
        exp = orc.sw(qs[i], rs[i])
        if exp == "no_match":
            assert out["status"][i] == 1, (qs[i], rs[i])
        else:
            cig = out["cigar"][int(out["cig_off"][i]):int(out["cig_off"][i + 1])].decode()
            assert (out["status"][i], int(out["score"][i]), cig) == (0, exp[0], exp[1]), (qs[i], rs[i])
            nshort += cig == "%dM" % len(qs[i])
    assert nshort > 200


def test_prefilter_two_mismatch_detours():
    """Square window, two mismatches on the diagonal: the prefilter answers {2Lq-6, "<Lq>M"} unless one of two
    single-insertion detours exists (k_sw_prefilter).  Inputs built to contain exactly those detours and their
    near misses -- run-structured references read one base off the diagonal before / after an inserted base --
    plus two-letter random pairs where off-diagonal matches are common."""
    from bwt_amd import lib
    from oracle import orc
    rng = np.random.default_rng(321)
    acgt = b"ACGT"
    qs, rs = [], []

    def runs(lens, tail):
        out = bytearray()
        for i, n in enumerate(lens):
            out += bytes([acgt[i & 3]]) * n
        return out + bytes(rng.choice(list(acgt), tail).tolist())

    for a in (1, 2, 4):
        for b in (1, 3, 5):
            for c in (1, 2, 6):
                for tail in (0, 3, 20, 60):
                    r = runs((a, b, c), tail)
                    L = len(r)
                    if L < 8:
                        continue
                    for t in range(0, min(L, a + b + c + 2)):
                        q = bytearray(r)                       # detour (a): diagonal +1 up to t, then the main diagonal
                        for k in range(min(t, L - 1)):
                            q[k] = r[k + 1]
                        qs.append(bytes(q)); rs.append(bytes(r))
                        q2 = bytearray(q)                      # ... and the inserted base itself mismatching
                        q2[min(t, L - 1)] = acgt[(acgt.index(q2[min(t, L - 1)]) + 1) & 3]
                        qs.append(bytes(q2)); rs.append(bytes(r))
                    rr = bytes(r[::-1])                        # detour (b): the mirror image, runs at the window's end
                    for t in range(max(0, L - (a + b + c + 2)), L):
                        q = bytearray(rr)
                        for k in range(t + 1, L):
                            q[k] = rr[k - 1]
                        qs.append(bytes(q)); rs.append(rr)
                        q2 = bytearray(q)
                        q2[t] = acgt[(acgt.index(q2[t]) + 1) & 3]
                        qs.append(bytes(q2)); rs.append(rr)
    for lq in (8, 9, 12, 16, 33, 40):
        for _ in range(1500):
            r = bytes(rng.choice(list(b"AC"), lq).tolist())
            q = bytearray(r)
            for p in rng.choice(lq, 2, replace=False):
                q[p] = b"C"[0] if q[p] == b"A"[0] else b"A"[0]
            qs.append(bytes(q)); rs.append(r)
    out = lib.sw_batch(*lib.pack_reads(qs), *lib.pack_reads(rs))
    ndiag = nother = 0
    for i in range(len(qs)):
        exp = orc.sw(qs[i], rs[i])
        if exp == "no_match":
            assert out["status"][i] == 1, (qs[i], rs[i])
            continue
        cig = out["cigar"][int(out["cig_off"][i]):int(out["cig_off"][i + 1])].decode()
        assert (out["status"][i], int(out["score"][i]), cig) == (0, exp[0], exp[1]), (qs[i], rs[i])
        if cig == "%dM" % len(qs[i]):
            ndiag += 1
        else:
            nother += 1
    assert ndiag > 2000 and nother > 300          # both outcomes are well represented


def test_edge_batches(gpu_index, orc_fm, ref_bytes):
    """Empty batch, reads shorter than MIN_LEN, a single read, and 600 bp reads (more than 64 sub-reads:
    the workgroup clustering path and the generic strip kernels)."""
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ref1 = E.make_ref1(ref_bytes)
    out = gpu_index.align_batch(*lib.pack_reads([]))
    assert len(out["hit_off"]) == 1 and len(out["score"]) == 0
    out = gpu_index.sga_batch(*lib.pack_reads([b"ACGTAC", b"", b"ACGTACG"]))
    assert out["clus_off"].tolist() == [0, 0, 0, 0]
    long_reads = synth_reads(ref_bytes, 40, 600, 301, max_subs=4) + synth_reads(ref_bytes, 10, 600, 302, indel_frac=1.0)
    for reads in ([ref_bytes[5000:5100]], long_reads):
        bases, off = lib.pack_reads(reads)
        out = gpu_index.align_batch(bases, off)
        for i, q in enumerate(reads):
            a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
            exp = orc.align_read(orc_fm, ref1, q)
            if exp is None:
                assert out["read_status"][i] == 1 and a == b
                continue
            got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                    int(out["pos"][h]), int(out["win_off"][h]), int(out["win_len"][h])) for h in range(a, b)]
            assert got == exp, (i, len(q))


def test_size_independent_properties_at_scale():
    """Properties that hold whatever the size (checked where the oracle cannot follow: 60 M-row index, 1.5 M reads of 100 bp
    through the fast seed kernel):
      * sharding: the hit tables of a batch are the concatenation of the hit tables of its slabs (what --gpus N relies on);
      * idempotence: a second run of the same resident batch gives the same tables;
      * every CIGAR consumes exactly Lq read bases (sum of M and I), every score >= 2 * ceil(Lq / 2) (sw.erl:62,69,79) and
        <= 2 Lq, pos = win_off - 200 + shift, win_len >= Lq;
      * a read copied from the genome without edits whose hit list is not empty holds a full-score hit whose window covers the
        read's own position."""
    import re
    from bwt_amd import lib
    rng = np.random.default_rng(77)
    n_ref, n_reads, lq, shift = 60_000_000, 1_500_000, 100, 5
    ref = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n_ref, dtype=np.uint8)]
    ix = lib.Index.build(ref, shift=shift)
    try:
        ix.set_ktable(15, 1)
        starts = rng.integers(0, n_ref - lq, n_reads)
        bases = ref[starts[:, None] + np.arange(lq)[None, :]].copy()
        edited = rng.random(n_reads) < 0.6
        rows = np.nonzero(edited)[0]
        cols = rng.integers(0, lq, len(rows))
        bases[rows, cols] = np.frombuffer(b"ACGT", np.uint8)[(np.searchsorted(np.frombuffer(b"ACGT", np.uint8), bases[rows, cols]) + rng.integers(1, 4, len(rows))) & 3]
        bases = bases.reshape(-1)
        off = np.arange(0, (n_reads + 1) * lq, lq, dtype=np.uint64)
        b = lib.Batch(ix)
        b.set_reads(bases, off)
        c = b.run(lib.STAGE_EXTEND)
        whole = b.fetch_hits()
        assert c["n_handed_back"] < n_reads // 100              # the fast kernel kept (nearly) every read
        b.run(lib.STAGE_EXTEND)
        again = b.fetch_hits()
        for k in whole:
            assert np.array_equal(np.asarray(whole[k]), np.asarray(again[k])) if k != "cigar" else whole[k] == again[k], k
        # sharding: three uneven slabs
        cuts = [0, 400_003, 1_100_001, n_reads]
        ho, co = 0, 0
        for a, e in zip(cuts[:-1], cuts[1:]):
            b.set_reads(bases[a * lq:e * lq], off[: e - a + 1])
            b.run(lib.STAGE_EXTEND)
            part = b.fetch_hits()
            nh = len(part["score"])
            assert np.array_equal(part["read_status"], whole["read_status"][a:e])
            assert np.array_equal(part["hit_off"].astype(np.int64) + ho, whole["hit_off"][a:e + 1].astype(np.int64))
            for k in ("score", "pos", "win_off", "win_len"):
                assert np.array_equal(part[k], whole[k][ho:ho + nh]), k
            assert part["cigar"] == whole["cigar"][co:co + len(part["cigar"])]
            ho += nh; co += len(part["cigar"])
        assert ho == len(whole["score"]) and co == len(whole["cigar"])
        b.destroy()
        # per-hit laws
        score, pos, woff, wlen = whole["score"], whole["pos"], whole["win_off"], whole["win_len"]
        assert score.min() >= 2 * ((lq + 1) // 2) and score.max() <= 2 * lq
        assert np.array_equal(pos, woff - 200 + shift) and wlen.min() >= lq
        cigs = whole["cigar"]
        co_ = whole["cig_off"].astype(np.int64)
        plain = (b"%dM" % lq)
        is_plain = np.array([cigs[co_[h]:co_[h + 1]] == plain for h in range(0, len(score), 97)])      # a sample of the texts
        assert is_plain.mean() > 0.9
        for h in np.nonzero(np.diff(co_) != len(plain))[0][:2000]:                                         # every other text form
            ops = re.findall(rb"(\d+)([MID])", cigs[co_[h]:co_[h + 1]])
            assert b"".join(n + o for n, o in ops) == cigs[co_[h]:co_[h + 1]]
            assert sum(int(n) for n, o in ops if o in b"MI") == lq
        # unedited reads with hits hold the full-score diagonal at their own position
        hoff = whole["hit_off"].astype(np.int64)
        clean = np.nonzero(~edited & (np.diff(hoff) > 0))[0]
        assert len(clean) > n_reads // 10
        first = hoff[clean]
        own = np.zeros(len(clean), bool)
        for d in range(int(np.diff(hoff).max())):
            h = np.minimum(first + d, hoff[clean + 1] - 1)
            # (Pos is the WINDOW start, src/worker_bwt.erl:94: a cluster with a stray seed in front of the read widens the window to the left)
            own |= (pos[h] <= starts[clean] + shift) & (starts[clean] + shift <= pos[h] + wlen[h] - lq) & (score[h] == 2 * lq)
        assert own.all()
    finally:
        ix.close()
