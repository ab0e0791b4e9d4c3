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
print(json.dumps(out, indent=1))
