#!/usr/bin/env python3
"""FASTQ ingest rate of the native reader (bwt_fastq_read_seqs): sequential against threaded, on a synthetic file in /tmp.
    python scripts/fastq_bench.py [records] [batch]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bwt_amd import fastq, lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
nb = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
path = "/tmp/fastq_bench_%d.fastq" % os.getpid()
rng = np.random.default_rng(1)
with open(path, "wb") as f:
    qual = b"I" * 100
    for lo in range(0, n, 200_000):
        m = min(200_000, n - lo)
        seqs = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (m, 100), dtype=np.uint8)]
        f.write(b"".join(b"@SRR770176.%d %d length=100\n" % (lo + i, lo + i) + seqs[i].tobytes() + b"\n+\n" + qual + b"\n" for i in range(m)))
size = os.path.getsize(path)
l = lib.load()
off = np.zeros(nb + 1, np.uint64); noff = np.zeros(nb + 1, np.uint64)
bases = np.zeros(nb * 128, np.uint8); names = np.zeros(nb * 64, np.uint8)
k = C.c_uint32(0); st = C.c_int(0); need = (C.c_uint64 * 2)()
print("%d records, %.0f MB, batches of %d, %d hardware threads" % (n, size / 1e6, nb, os.cpu_count()))
for threads in (1, 1, 4, 16, 64, 0):
    dev = fastq.Dev(path)
    dev.set_threads(threads, 1 if threads != 1 else 1 << 60)
    tt, tot = 0.0, 0
    while True:
        t0 = time.perf_counter()
        rc = l.bwt_fastq_read_seqs(dev.h, C.c_uint32(nb), lib._p(bases), C.c_uint64(len(bases)), lib._p(off), lib._p(names),
                                   C.c_uint64(len(names)), lib._p(noff), C.byref(k), C.byref(st), need)
        tt += time.perf_counter() - t0
        assert rc == 0, rc
        tot += k.value
        if st.value != 0:
            break
    print("threads %2d: %.3f s, %.1f M records/s, %.2f GB/s of text" % (threads, tt, tot / tt / 1e6, size / tt / 1e9), flush=True)
    dev.close()
os.remove(path)
