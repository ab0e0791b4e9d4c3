"""Latency of one bwt_align_batch-sized call for small batches (what a worker_bwt QseqList looks like)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from bwt_amd import lib
g = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests", "golden", "fm_indices", "GL000193.1_p1")
ix = lib.Index.open(g + ".fm", g + ".ref")
ref = np.frombuffer(open(g + ".ref", "rb").read(), np.uint8)
rng = np.random.default_rng(1)
for n in (200, 2000, 20000, 200000):
    st = rng.integers(0, len(ref) - 100, n)
    bases = ref[st[:, None] + np.arange(100)[None, :]].reshape(-1).copy()
    off = np.arange(0, (n + 1) * 100, 100, dtype=np.uint64)
    b = lib.Batch(ix)
    for _ in range(3):
        b.set_reads(bases, off); b.run(lib.STAGE_EXTEND); b.fetch_hits()
    t = time.perf_counter()
    reps = 20
    for _ in range(reps):
        b.set_reads(bases, off); b.run(lib.STAGE_EXTEND); out = b.fetch_hits()
    dt = (time.perf_counter() - t) / reps
    print("n=%d: %.3f ms per call, %.2f M reads/s (host in, host out)" % (n, dt * 1e3, n / dt / 1e6))
    b.destroy()
