"""Shared body of the full-size GPU checks (tests/test_gpu_fullsize*.py): one batch through the host boundary the bench times
(packed upload, bwt_pipe, compact hit table, host-side expansion), submitted twice, compared read by read with the oracle's
per-read digest of the COMPLETE answer (every hit's score, position, window, CIGAR text, list order: oracle/bwt_oracle.c
orc_batch_digest).  TEST INFRASTRUCTURE: the oracle is the checker here, never the thing run."""
import os

import numpy as np


def sprinkle_exceptions(bases, n_reads, length, seed, one_in=100):
    """an N / IUPAC base in one read of `one_in` (they travel in the exception list of the packed upload)"""
    rng = np.random.default_rng(seed)
    dirty = rng.integers(0, n_reads, max(1, n_reads // one_in))
    bases[dirty * length + rng.integers(0, length, len(dirty))] = np.frombuffer(b"NNNNRYKM", np.uint8)[rng.integers(0, 8, len(dirty))]
    return bases


def pipe_twice_against_digest(ix, ofm, ref1, bases, off, n_reads, length, options=None):
    """-> (expanded hit arrays, counts, number of extras); asserts both submissions agree and equal the oracle's digests"""
    from bwt_amd import lib
    from oracle import orc
    codes, exc = lib.pack_2bit(bases)
    pipe = lib.Pipe(ix, depth=2)
    if options:
        pipe.set_options(**options)
    try:
        outs = []
        for _ in range(2):
            s, v = pipe.acquire_packed(len(bases), n_reads, len(exc))
            v["codes"][:len(codes)] = codes; v["exc"][:len(exc)] = exc
            pipe.submit_packed(s, n_reads, length, len(bases), len(exc))
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
    assert len(bad) == 0, (len(bad), bad[:10].tolist())
    return out, c, n_extra, len(exc)
