"""Driver-visible full-size checks: BASELINE configs[2] (chr1-scale index, 100 bp reads with <= 2 substitutions)
and configs[4]'s read shape (150 bp mates with <= 2 substitutions and an indel of 1-5 bp) at the same index size, read by read
against the oracle's per-read digest of the COMPLETE answer (every hit's score, position, window, CIGAR text, list order:
oracle/bwt_oracle.c orc_batch_digest) -- 200 000 reads each, through the host boundary the bench times (packed upload, compact
hit tables, bwt_pipe).  No chr1 FASTA exists offline: the genome is the bench's seeded i.i.d. contig of 2.3e8 bp."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

N_REF, N_READS = 230_000_000, 200_000


@pytest.fixture(scope="module")
def chr1_scale():
    sys.path.insert(0, ROOT)
    import bench
    from bwt_amd import lib
    from oracle import orc, erl_literal as E
    ref = bench.synth_genome(N_REF, seed=3)
    ix = lib.Index.build(ref, shift=1, device=0)
    assert ix.meta()[0][3] == N_REF + 1 and ix.ktable == (15, True)          # the suffix-interval table the bench runs with
    L, SA = ix.export()
    ptr, shift = ix.meta()
    ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
    del L, SA
    ref1 = E.make_ref1(ref.tobytes())
    yield bench, ix, ref, ofm, ref1
    ix.close()


@pytest.mark.parametrize("length,indel_p,name", [(100, 0.0, "configs[2]: 100 bp, 0/1/2 substitutions"),
                                                 (150, 0.2, "configs[4] reads: 150 bp, 0/1/2 substitutions + an indel of 1-5 bp with p = 0.2")])
def test_full_size_index_against_oracle_digest(chr1_scale, length, indel_p, name):
    from bwt_amd import lib
    from oracle import orc
    bench, ix, ref, ofm, ref1 = chr1_scale
    bases, off = bench.synth_reads(ref, N_READS, length, seed=4000 + length, max_subs=2, indel_p=indel_p)
    rng = np.random.default_rng(length)
    dirty = rng.integers(0, N_READS, N_READS // 100)                           # a percent of the reads with an N / IUPAC base (exception list)
    bases[dirty * length + rng.integers(0, length, len(dirty))] = np.frombuffer(b"NNNNRYKM", np.uint8)[rng.integers(0, 8, len(dirty))]
    codes, exc = lib.pack_2bit(bases)
    assert len(exc) > 1000
    pipe = lib.Pipe(ix, depth=2)
    pipe.set_options(sw_band_min=0)            # (a DP queue of a few ten thousand candidates would be left to the wave kernel otherwise)
    try:
        # two submissions of the same batch: a steady pipe, and the second result must equal the first
        outs = []
        for _ in range(2):
            s, v = pipe.acquire_packed(len(bases), N_READS, len(exc))
            v["codes"][:len(codes)] = codes; v["exc"][:len(exc)] = exc
            pipe.submit_packed(s, N_READS, length, len(bases), len(exc))
        for _ in range(2):
            r = pipe.wait()
            out = pipe.expand(r)
            outs.append((out, r["counts"], len(r["extra"])))
            pipe.release(r["slot"])
    finally:
        pipe.close()
    (out, c, n_extra), (out2, _, _) = outs
    for k in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off"):
        assert np.array_equal(out[k], out2[k]), k
    assert out["cigar"] == out2["cigar"]
    gpu = orc.hits_digest(out)
    cpu = orc.batch_digest(ofm, ref1, bases, off.astype(np.uint32), os.cpu_count() or 1)
    bad = np.nonzero(gpu != cpu)[0]
    assert len(bad) == 0, (name, len(bad), bad[:10].tolist())
    # the run took the paths the bench takes: nearly every group stays in the fast seed kernel; 100 bp reads are settled by the
    # prefilter (plain hits, no extra record), the 150 bp reads with an indel go through the banded DP
    # (on a genome of this size about half of the reads end without a cluster: the reference's row ranges are shifted one row
    # down, DESIGN 2 -- the oracle agrees read by read)
    assert c["n_hits"] > 0.4 * N_READS and c["n_handed_back"] < N_READS // 20
    if indel_p:
        assert c["n_banded"] > 0.02 * N_READS and n_extra > 0.02 * N_READS
    else:
        assert n_extra < 0.02 * N_READS
