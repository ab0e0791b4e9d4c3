#!/usr/bin/env python3
"""End-to-end rate of bwt_run_fastq: FASTQ text on disk (/tmp) -> threaded reader -> pinned-buffer pipeline -> pick -> sink, against
a GRCh38-scale index resident in HBM.  Two runs: threshold 265 (master.erl:149: no 100-bp read ever prints, the job is ingest +
alignment + pick) and threshold 100 (every cast prints: the sink's text dominates).
    python scripts/run_fastq_bench.py [reads] [genome_len]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bwt_amd import lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
glen = int(sys.argv[2]) if len(sys.argv) > 2 else 3_100_000_000
t0 = time.perf_counter()
ref = bench.synth_genome(glen, seed=5)
ix = lib.Index.build(ref, shift=1, device=0)
bases, off = bench.synth_reads(ref, n, 100, seed=77, max_subs=2)
path = "/tmp/run_fastq_%d.fastq" % os.getpid()
with open(path, "wb") as f:
    qual = b"I" * 100
    for lo in range(0, n, 250_000):
        m = min(250_000, n - lo)
        rows = bases[lo * 100:(lo + m) * 100].reshape(m, 100)
        f.write(b"".join(b"@SRR770176.%d %d length=100\n" % (lo + i, lo + i) + rows[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(m)))
size = os.path.getsize(path)
print("setup %.1f s: %d reads, FASTQ %.2f GB" % (time.perf_counter() - t0, n, size / 1e9), file=sys.stderr, flush=True)
res = {"reads": n, "fastq_bytes": size, "index_rows": glen + 1, "host_threads": os.cpu_count()}
lib.run_fastq(ix, path, "synthetic", 265, 0, None)                       # warm-up: page cache, pinned buffers
for name, thr, outp in (("count_only_threshold_265", 265, None), ("all_lines_threshold_100", 100, "/tmp/run_fastq_%d.out" % os.getpid())):
    st = lib.run_fastq(ix, path, "synthetic", thr, 0, outp)
    st["reads_per_s"] = st["reads"] / st["seconds"]
    st["fastq_GB_per_s"] = size / st["seconds"] / 1e9
    if outp:
        st["output_bytes"] = os.path.getsize(outp)
        os.remove(outp)
    res[name] = st
os.remove(path)
print(json.dumps(res))
