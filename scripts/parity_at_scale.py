#!/usr/bin/env python3
"""At-scale parity: per-read digests of the COMPLETE answer (every hit's score, pos, window, CIGAR text, in list order) from
the GPU path against the CPU oracle, read by read, on a whole-GRCh38-scale index and millions of reads -- where
bench.py's cpu_baseline leg can only afford aggregates over a few hundred thousand.
    python scripts/parity_at_scale.py [--reads N] [--genome-len N] [--length L] > profiles/rN/parity_at_scale.json"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bwt_amd import lib  # noqa: E402
from oracle import orc, erl_literal as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=4_000_000)
    ap.add_argument("--genome-len", type=int, default=3_100_000_000)
    ap.add_argument("--length", type=int, default=100)
    ap.add_argument("--max-subs", type=int, default=2)
    ap.add_argument("--repeats", type=int, default=0, help="plant this many copies of 40 segments of 600 bp each (reads from them: many seeds, several clusters)")
    ap.add_argument("--spectrum", action="store_true", help="the bench's repeat-rich genome (bench.synth_genome_repeats: SINE / LINE-like families, mid-copy families, tandem and low-complexity tracts, segmental duplications)")
    ap.add_argument("--seed", type=int, default=91)
    a = ap.parse_args()
    t0 = time.perf_counter()
    composition = None
    if a.spectrum:
        ref, composition = bench.synth_genome_repeats(a.genome_len, seed=5)
    else:
        ref = bench.synth_genome(a.genome_len, seed=5)
    rep_starts = []
    if a.repeats:
        ref = ref.copy()
        rr = np.random.default_rng(a.seed + 7)
        for s_ in range(40):
            src = int(rr.integers(0, a.genome_len - 700))
            seg = ref[src:src + 600].copy()
            for _ in range(a.repeats):
                dst = int(rr.integers(0, a.genome_len - 700))
                ref[dst:dst + 600] = seg
                rep_starts.append(dst)
    ix = lib.Index.build(ref, shift=1, device=0)
    n, L = a.reads, a.length
    # three quarters like the bench batch (0/1/2 substitutions), the rest with an indel; a sprinkle of N and IUPAC bases on top
    b1, _ = bench.synth_reads(ref, n * 3 // 4, L, seed=a.seed, max_subs=a.max_subs)
    b2, _ = bench.synth_reads(ref, n - n * 3 // 4, L, seed=a.seed + 1, max_subs=a.max_subs, indel_p=1.0)
    bases = np.concatenate([b1, b2])
    rng = np.random.default_rng(a.seed + 2)
    if rep_starts:                                   # a tenth of the reads from the planted repeats
        k = n // 10
        st = np.array(rep_starts, np.int64)[rng.integers(0, len(rep_starts), k)] + rng.integers(0, 600 - L, k)
        rows = rng.choice(n, k, replace=False)
        blk = ref[st[:, None] + np.arange(L)[None, :]]
        bases.reshape(n, L)[rows] = blk
    dirty = rng.integers(0, n, n // 50)
    bases[dirty * L + rng.integers(0, L, len(dirty))] = np.frombuffer(b"NNNNRYKM", np.uint8)[rng.integers(0, 8, len(dirty))]
    off = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    print("setup %.1f s" % (time.perf_counter() - t0), file=sys.stderr, flush=True)
    # through the wire formats of the host boundary (round 3): packed upload, compact hit table, host-side expansion
    # (round 4: the table in its NARROW form -- what bench.py's pipe downloads -- decoded by bwt_hits_expand_narrow)
    batch = lib.Batch(ix)
    batch.set_options(hits_narrow=1)
    codes, exc = lib.pack_2bit(bases)
    batch.set_reads_packed(codes, exc, n, read_len=L)
    t0 = time.perf_counter()
    c = batch.run(lib.STAGE_EXTEND)
    t_gpu = time.perf_counter() - t0
    comp = batch.fetch_hits_narrow()
    hits = lib.hits_expand(n, L, None, ix.meta()[1], comp["hits"], comp["extra"], comp["cigar"], comp["read_status"], comp["narrow_off"], comp["narrow_score"])
    st = comp["read_status"].astype(np.int32)
    st[st == 2] = 0
    hits["read_status"] = st
    gpu = orc.hits_digest(hits)
    Lc, SA = ix.export()
    ptr, shift = ix.meta()
    ofm = orc.OrcFM.from_L_SA(Lc, SA, ptr, shift)
    ref1 = E.make_ref1(ref.tobytes())
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    cpu = np.zeros(n, np.uint64)
    step = 500_000
    for lo in range(0, n, step):
        hi = min(n, lo + step)
        cpu[lo:hi] = orc.batch_digest(ofm, ref1, bases[lo * L:hi * L], (off[lo:hi + 1] - off[lo]).astype(np.uint32), cores)
        print("cpu %d / %d" % (hi, n), file=sys.stderr, flush=True)
    t_cpu = time.perf_counter() - t0
    bad = np.nonzero(gpu != cpu)[0]
    out = {"repeat_spectrum": composition,
           "workload": "i.i.d. genome of %d bp%s, %d reads of %d bp: 3/4 with 0-%d substitutions, 1/4 with an indel of 1-5 bp "
                       "as well, 2 %% of the reads with an N or IUPAC base%s" % (a.genome_len, (" with 40 segments of 600 bp in %d copies each" % a.repeats) if a.repeats else "",
                                                                                 n, L, a.max_subs, ", a tenth of the reads from the repeats" if a.repeats else ""),
           "compared": "per-read digest of the complete hit list (score, pos, window offset and length, CIGAR text, list order); GPU side: packed upload (%d exceptions), narrow hit table (%d narrow reads, %d records, %d extras), expanded on the host; suffix-interval table K = %d, lower levels kept wide %s; head %s" % (len(exc), len(comp["narrow_off"]), len(comp["hits"]), len(comp["extra"]), ix.ktable[0], ix.ktable_wide_levels, os.environ.get("BWT_HEAD", "?")),
           "reads": n, "reads_with_hits": int((np.diff(hits["hit_off"].astype(np.int64)) > 0).sum()), "hits": int(c["n_hits"]),
           "reads_status_minus1": int((hits["read_status"] == -1).sum()), "dp_candidates": int(c["n_dp"]), "fp64_candidates": int(c["n_float"]),
           "handed_back_reads": int(c["n_handed_back"]), "banded": int(c.get("n_banded", 0)), "certified_without_dp": int(c.get("n_certified", 0)), "mismatching_reads": int(len(bad)), "first_mismatches": bad[:10].tolist(),
           "gpu_s": t_gpu, "cpu_s": t_cpu, "cpu_threads": cores}
    print(json.dumps(out))
    batch.destroy(); ix.close()
    sys.exit(1 if len(bad) else 0)


if __name__ == "__main__":
    main()
