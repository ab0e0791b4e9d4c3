"""Latency of one call at the reference's message sizes (src/master.erl:59: 10 lists of 200 reads; src/worker_bwt.erl:62-69),
host buffers in -> hit tables out, on the bundled index: the general path (bwt_batch_set_reads + bwt_batch_run +
bwt_batch_fetch_hits: what bwt_align_batch does) stage by stage, and the small-batch entry point (bwt_small_align) when the
library has it.    python scripts/small_batch_latency.py > profiles/r4/small_batch_latency.json"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np          # noqa: E402
from bwt_amd import lib     # noqa: E402

g = os.path.join(ROOT, "tests", "golden", "fm_indices", "GL000193.1_p1")
ix = lib.Index.open(g + ".fm", g + ".ref")
if len(sys.argv) > 1:                       # "15,1": the suffix-interval table at K = 15 with a wide top level instead of the index's default
    ix.set_ktable(*[int(x) for x in sys.argv[1].split(",")])
ref = np.frombuffer(open(g + ".ref", "rb").read(), np.uint8)
rng = np.random.default_rng(1)
out = {"index": "bundled GL000193.1_p1 (K = %d, wide = %s)" % ix.ktable, "sizes": {}}
for n in (200, 2000, 4096, 20000):
    st = rng.integers(0, len(ref) - 100, n)
    bases = ref[st[:, None] + np.arange(100)[None, :]].reshape(-1).copy()
    k = rng.integers(0, 3, n)                                           # 0 / 1 / 2 substitutions, as the bench's reads
    for s in range(2):
        rows = np.nonzero(k > s)[0]
        bases.reshape(n, 100)[rows, rng.integers(0, 100, len(rows))] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, len(rows))]
    off = np.arange(0, (n + 1) * 100, 100, dtype=np.uint64)
    rec = {}
    b = lib.Batch(ix)
    for _ in range(3):
        b.set_reads(bases, off); c = b.run(lib.STAGE_EXTEND); b.fetch_hits()
    reps = 30
    ts = [0.0, 0.0, 0.0]
    for _ in range(reps):
        t0 = time.perf_counter(); b.set_reads(bases, off)
        t1 = time.perf_counter(); b.run(lib.STAGE_EXTEND)
        t2 = time.perf_counter(); ref_out = b.fetch_hits()
        t3 = time.perf_counter()
        ts[0] += t1 - t0; ts[1] += t2 - t1; ts[2] += t3 - t2
    b.profile(True)
    for _ in range(reps):
        b.set_reads(bases, off); b.run(lib.STAGE_EXTEND)
    kt = b.kernel_times()
    b.profile(False)
    rec["general_path_ms"] = {"set_reads": ts[0] / reps * 1e3, "run": ts[1] / reps * 1e3, "fetch_hits": ts[2] / reps * 1e3,
                              "total": sum(ts) / reps * 1e3}
    rec["general_path_device_ms_per_call"] = {k_: v["ms"] / reps for k_, v in sorted(kt.items())}
    rec["general_path_launch_groups_per_call"] = sum(v["launches"] for v in kt.values()) / reps
    rec["counts"] = {k_: c[k_] for k_ in ("n_clusters", "n_dp", "n_float", "n_hits", "n_banded")}
    b.destroy()
    if hasattr(lib, "SmallBatch") and n <= lib.SMALL_MAX_READS:
        sb = lib.SmallBatch(ix)
        for _ in range(5):
            got = sb.align(bases, off)
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            got = sb.align(bases, off, expand=False)
        rec["small_path_ms"] = (time.perf_counter() - t0) / (reps * 4) * 1e3
        t0 = time.perf_counter()
        for _ in range(reps * 4):
            sb.align_raw(bases, off)
        rec["small_path_ms_c_call_only"] = (time.perf_counter() - t0) / (reps * 4) * 1e3
        rec["small_path_stage_us"] = sb.stage_us()
        sb.profile(True)
        for _ in range(reps):
            sb.align_raw(bases, off)
        rec["small_path_device_us_per_call"] = {k_: round(v["ms"] / reps * 1e3, 2) for k_, v in sorted(sb.kernel_times().items())}
        rec["small_path_launches_per_call"] = sum(v["launches"] for v in sb.kernel_times().values()) / reps
        sb.profile(False)
        got = sb.align(bases, off)
        same = all(np.array_equal(got[k_], ref_out[k_]) for k_ in ("read_status", "hit_off", "score", "pos", "win_off", "win_len", "cig_off")) and got["cigar"] == ref_out["cigar"]
        rec["small_path_equals_general_path"] = bool(same)
        rec["small_path_fell_back"] = sb.fallbacks
        sb.close()
    out["sizes"][str(n)] = rec
    print("n=%d: %s" % (n, json.dumps(rec)), file=sys.stderr, flush=True)
print(json.dumps(out))
