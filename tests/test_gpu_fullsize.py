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
from _fullsize import pipe_twice_against_digest, sprinkle_exceptions

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
    bench, ix, ref, ofm, ref1 = chr1_scale
    bases, off = bench.synth_reads(ref, N_READS, length, seed=4000 + length, max_subs=2, indel_p=indel_p)
    sprinkle_exceptions(bases, N_READS, length, seed=length)                  # a percent of the reads with an N / IUPAC base (exception list)
    # (sw_band_min = 0: a DP queue of a few ten thousand candidates would be left to the wave kernel otherwise)
    out, c, n_extra, n_exc = pipe_twice_against_digest(ix, ofm, ref1, bases, off, N_READS, length, options={"sw_band_min": 0})
    assert n_exc > 1000
    # the run took the paths the bench takes: nearly every group stays in the fast seed kernel; 100 bp reads are settled by the
    # prefilter (plain hits, no extra record), the 150 bp reads with an indel go through the banded DP
    # (on a genome of this size about half of the reads end without a cluster: the reference's row ranges are shifted one row
    # down, DESIGN 2 -- the oracle agrees read by read)
    assert c["n_hits"] > 0.4 * N_READS and c["n_handed_back"] < N_READS // 20
    if indel_p:
        assert c["n_banded"] > 0.02 * N_READS and n_extra > 0.02 * N_READS
    else:
        assert n_extra < 0.02 * N_READS
