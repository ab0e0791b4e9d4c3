"""BASELINE configs[3] and configs[4] at their REAL index size: a whole-GRCh38-scale index -- 3.1e9 rows, row
numbers past 2^31, positions past 2^31, the 71.6 GB suffix-interval table the bench runs with -- built once per module, then
   * 200 000 reads of 100 bp with 0 / 1 / 2 substitutions            (configs[3]'s reads)
   * 200 000 mates of 150 bp with <= 2 substitutions and an indel    (configs[4]'s reads; mates aligned independently, as the
     of 1-5 bp with p = 0.2                                            reference does)
   * 40 000 reads of 100 bp from the LAST 10^6 bases of the contig, a few hundred of them ending exactly at the last base
     (u32 row / i64 position edges; windows that reach into the trailing 200-N halo of Ref1, src/worker_bwt.erl:54-55,86-94)
each with 1 % of the reads carrying an N / IUPAC base, through the host boundary the bench times (packed upload, bwt_pipe, compact
hit table), read by read against the oracle's digest of the complete answer.  No GRCh38 FASTA exists offline: the genome is the
bench's seeded i.i.d. contig.  (Its own module: the chr1-scale module's index and table are released before this one is built.)"""
import sys

import numpy as np
import pytest

from conftest import ROOT
from _fullsize import pipe_twice_against_digest, sprinkle_exceptions

pytestmark = pytest.mark.gpu

N_REF, N_READS = 3_100_000_000, 200_000


@pytest.fixture(scope="module")
def grch38_scale():
    sys.path.insert(0, ROOT)
    import bench
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ref = bench.synth_genome(N_REF, seed=5)                                   # the bench's genome
    ix = lib.Index.build(ref, shift=1, device=0)
    (pc, pg, pt, last), shift = ix.meta()
    assert last == N_REF + 1 and last > 2 ** 31 and pt > 2 ** 31 and ix.ktable == (15, True)
    L, SA = ix.export()
    ofm = orc.OrcFM.from_L_SA(L, SA, (pc, pg, pt, last), shift)
    del L, SA
    ref1 = E.make_ref1(ref.tobytes())
    yield bench, ix, ref, ofm, ref1
    ix.close()


def test_configs3_reads_at_whole_genome_scale(grch38_scale):
    bench, ix, ref, ofm, ref1 = grch38_scale
    length = 100
    bases, off = bench.synth_reads(ref, N_READS, length, seed=5000 + 77, max_subs=2)
    sprinkle_exceptions(bases, N_READS, length, seed=3)
    out, c, n_extra, n_exc = pipe_twice_against_digest(ix, ofm, ref1, bases, off, N_READS, length, options={"sw_band_min": 0})
    assert n_exc > 1000
    # the paths the headline takes: nearly every read stays in the fast seed kernel and is settled by the prefilter (plain hits)
    assert c["n_hits"] > 0.9 * N_READS and c["n_handed_back"] < N_READS // 20 and n_extra < 0.02 * N_READS
    assert int(out["pos"].max()) > 2 ** 31                                    # hits in the upper half of the position range


def test_configs4_mates_at_whole_genome_scale(grch38_scale):
    bench, ix, ref, ofm, ref1 = grch38_scale
    length = 150
    bases, off = bench.synth_reads(ref, N_READS, length, seed=6000 + 77, max_subs=2, indel_p=0.2)
    sprinkle_exceptions(bases, N_READS, length, seed=4)
    out, c, n_extra, n_exc = pipe_twice_against_digest(ix, ofm, ref1, bases, off, N_READS, length, options={"sw_band_min": 0})
    assert n_exc > 1000
    assert c["n_hits"] > 0.8 * N_READS and c["n_handed_back"] < N_READS // 20
    assert c["n_banded"] > 0.1 * N_READS and n_extra > 0.1 * N_READS          # the indel mates went through the banded DP, CIGARs with I / D


@pytest.mark.parametrize("length", [100, 150])
def test_far_end_of_the_contig(grch38_scale, length):
    """reads from the last 10^6 bases: SA values and positions just under the row count (> 2^31), and -- for the reads that end at
    the contig's last base -- windows that touch or enter the trailing halo (a cluster of width D > 0 there slices Ref1 past
    the last base: src/worker_bwt.erl:86-90), which the fp64 path scores with 1.6 per N (src/sw.erl:164-202)"""
    bench, ix, ref, ofm, ref1 = grch38_scale
    n = 40_000
    rng = np.random.default_rng(length)
    back = rng.integers(0, 1_000_000, n)
    back[:400] = 0                                                            # ... ending exactly at the last base
    back[400:800] = rng.integers(1, 6, 400)                                   # ... and a few bases before it
    st = (len(ref) - length - back).astype(np.int64)
    blk = ref[st[:, None] + np.arange(length, dtype=np.int64)[None, :]]
    code = np.zeros(256, np.uint8)
    code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    alpha = np.frombuffer(b"ACGT", np.uint8)
    k = rng.integers(0, 3, n)
    for s in range(2):
        rows = np.nonzero(k > s)[0]
        cols = rng.integers(0, length, len(rows))
        blk[rows, cols] = alpha[(code[blk[rows, cols]] + rng.integers(1, 4, len(rows), dtype=np.uint8)) & 3]
    # a deletion of 1-5 read bases in a tenth of them: the cluster widens (D > 0) and, at the far end, the window enters the halo
    bases = blk.reshape(-1).copy()
    cut = rng.choice(n, n // 10, replace=False)
    for r in cut.tolist():
        d = int(rng.integers(1, 6)); p = int(rng.integers(20, length - 20))
        row = bases[r * length:(r + 1) * length]
        row[p:length - d] = row[p + d:].copy()
        row[length - d:] = alpha[rng.integers(0, 4, d)]
    sprinkle_exceptions(bases, n, length, seed=length + 1)
    off = np.arange(0, (n + 1) * length, length, dtype=np.uint64)
    out, c, n_extra, n_exc = pipe_twice_against_digest(ix, ofm, ref1, bases, off, n, length, options={"sw_band_min": 0})
    assert c["n_hits"] > 0.8 * n
    assert int(out["pos"].min()) > 2 ** 31 and int(out["pos"].max()) >= len(ref) - length - 2
    assert int((out["win_off"] + out["win_len"]).max()) >= len(ref) + 200     # a window that ends at (or inside) the trailing halo
