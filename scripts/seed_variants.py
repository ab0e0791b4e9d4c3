#!/usr/bin/env python3
"""GPU experiment driver: one GRCh38-scale index, the bench batch, and the seed stage timed under several library
switches (env, read per call) -- kernel times from the library's own hipEvents, parity by the hit-table aggregate.
    python scripts/seed_variants.py [--genome-len N] [--reads N] VAR=VAL[,VAR=VAL] ...   (each argument = one variant)"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genome-len", type=int, default=3_100_000_000)
    ap.add_argument("--reads", type=int, default=12_500_000)
    ap.add_argument("--length", type=int, default=100)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    t0 = time.perf_counter()
    ref = bench.synth_genome(a.genome_len, seed=5)
    ix = lib.Index.build(ref, shift=1, device=0)
    bases, off = bench.synth_reads(ref, a.reads, a.length, seed=5000, max_subs=2, indel_p=0.2 if a.length == 150 else 0.0)
    print("setup %.1f s, table %s" % (time.perf_counter() - t0, ix.ktable), flush=True)
    batch = lib.Batch(ix)
    batch.set_reads(bases, off)
    base_env = dict(os.environ)
    ref_sig = None
    for var in [""] + a.variants:
        os.environ.clear(); os.environ.update(base_env)
        for kv in filter(None, var.split(",")):
            k, v = kv.split("=")
            os.environ[k] = v
        batch.run(lib.STAGE_EXTEND)
        batch.profile(True)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            c = batch.run(lib.STAGE_EXTEND)
        dt = (time.perf_counter() - t0) / a.steps
        kt = batch.kernel_times()
        batch.profile(False)
        h = batch.fetch_hits()
        sig = (int(c["n_hits"]), int(c["n_clusters"]), int((h["pos"].astype(np.int64) * 31 + h["score"]).sum()),
               int(np.frombuffer(h["cigar"], np.uint8).astype(np.int64).sum()))
        if ref_sig is None:
            ref_sig = sig
        per = {k: round(v["ms"] / v["launches"], 3) for k, v in kt.items()}
        print(json.dumps({"variant": var or "default", "ms_per_step": round(dt * 1e3, 3), "Mreads_per_s": round(a.reads / dt / 1e6, 1),
                          "same_hits": sig == ref_sig, "handed_back": c["n_handed_back"],
                          "kernels": {k: per[k] for k in sorted(per) if per[k] >= 0.02}}), flush=True)
    batch.destroy(); ix.close()


if __name__ == "__main__":
    main()
