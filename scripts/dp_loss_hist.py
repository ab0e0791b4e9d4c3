#!/usr/bin/env python3
"""Which DP tier does a candidate of the repeat-rich workload need?  Histogram of the loss 2 Lq - score over the hits of a batch of
the bench's grch38-repeats reads (scaled-down genome), split by window shape (D = W - Lq), plus the counters of the run.
    python scripts/dp_loss_hist.py [genome_len] [reads]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bwt_amd import lib  # noqa: E402

glen = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000_000
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
L = 100
ref, comp = bench.synth_genome_repeats(glen, seed=5)
ix = lib.Index.build(ref, shift=1, device=0)
bases, off = bench.synth_reads(ref, n, L, seed=11, max_subs=2)
b = lib.Batch(ix)
b.set_reads(bases, off)
counts = b.run(lib.STAGE_EXTEND)
t = b.fetch_hits()
score = np.asarray(t["score"]); wlen = np.asarray(t["win_len"])
loss = 2 * L - score
D = wlen - L
edges = [0, 1, 4, 7, 13, 25, 31, 43, 49, 73, 97, 101, 10**9]
out = {"genome_len": glen, "reads": n, "hits": int(len(score)), "counts": {k: int(v) for k, v in counts.items()} if isinstance(counts, dict) else str(counts)}
for name, sel in (("D=0", D == 0), ("D 1..13", (D > 0) & (D <= 13)), ("D 14..25", (D > 13) & (D <= 25)), ("D 26..57", (D > 25) & (D <= 57)), ("D>57", D > 57)):
    h, _ = np.histogram(loss[sel], bins=edges)
    out[name] = {"hits": int(sel.sum()), "loss_bins": {"%d..%d" % (edges[i], edges[i + 1] - 1): int(h[i]) for i in range(len(h))}}
wide = D[D > 154]          # W > 254: past the packed kernels' cap, the lane-per-alignment trace kernel's
out["W>254"] = {"hits": int(len(wide)), "D_percentiles": {str(q): int(np.percentile(wide, q)) for q in (50, 90, 99, 100)} if len(wide) else {},
                "cells": int(((wide.astype(np.int64) + L) * L).sum())}
mid = D[(D > 57) & (D <= 154)]
out["57<D<=154"] = {"hits": int(len(mid)), "cells": int(((mid.astype(np.int64) + L) * L).sum())}
# round 4: which candidates could a certificate for the plain diagonal take?  Square windows whose answer is "<L>M", by the number of
# mismatches m = loss / 3; square windows with another answer; wider windows by D, with the share of "<L>M" answers among them
cig = bytes(t["cigar"]); co = np.asarray(t["cig_off"])
plain_txt = ("%dM" % L).encode()
is_plain = np.fromiter((cig[int(co[h]):int(co[h + 1])] == plain_txt for h in range(len(score))), bool, len(score))
sq = D == 0
mm = loss[sq & is_plain] // 3
out["square_plain_by_mismatches"] = {str(k): int((mm == k).sum()) for k in range(0, 34)}
out["square_not_plain"] = int((sq & ~is_plain).sum())
out["wider_by_D"] = {str(d): {"hits": int((D == d).sum()), "answer_is_LM": int(((D == d) & is_plain).sum())} for d in range(1, 21)}
out["wider_D_over_20"] = {"hits": int((D > 20).sum()), "answer_is_LM": int(((D > 20) & is_plain).sum())}
print(json.dumps(out, indent=1))
