#!/usr/bin/env python3
"""bench.py -- aligned reads/s of the FM-index seed search + clustering + SW extension hot path.

    python bench.py --gpus N --steps K --warmup W [--workload grch38|chr1|config2|config2-align]

One process per GPU (torch.distributed / RCCL for N > 1; the driver launches the ranks).  A "step"
is one pass of the hot path over one batch of synthetic reads that is already resident in HBM.
Rank 0 prints ONE JSON line.  Weak scaling: every GPU gets its own slab of reads, the index is
replicated, and the only collective is the per-step gather of the 16-byte hit records to rank 0.

Workloads (BASELINE.json configs):
  grch38 (default)  configs[3] at N GPUs: whole-GRCh38-SCALE FM index (3.1e9 rows) resident in HBM,
                    100 M reads / 8 = 12.5 M 100-bp reads per GPU per step with 0/1/2 substitutions
                    (p = 1/3 each), seed search + clustering + SW extension + CIGAR.  No GRCh38 FASTA
                    exists offline, so the genome is a seeded i.i.d. ACGT contig (flagged synthetic);
                    the index is built on the GPU by bwt_build_index at start-up (not timed).
  chr1              configs[2]: 2.3e8-row index, 10 M reads, same path.
  config2           configs[1]: bundled GL000193.1 index, 1 M exact reads, FM backward search only.
  config2-align     the same reads through the whole path.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_TOPS = 256 * 4 * 16 * 2.4e9 / 1e12   # wave64 integer ops issue at 4 cycles per op per SIMD (SQ counters)
HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s measured streaming copy


def log(msg):
    print("[bench %s] %s" % (time.strftime("%H:%M:%S"), msg), file=sys.stderr, flush=True)


def synth_genome(n, seed):
    rng = np.random.default_rng(seed)
    return np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n, dtype=np.uint8)]


def synth_reads(ref, n, length, seed, max_subs, indel_p=0.0):
    """uniform start, exactly k substitutions with k uniform in 0..max_subs (a substituted
    base always differs from the original); with probability indel_p one insertion or deletion of 1-5 bp.
    Generated in slices to bound host memory."""
    rng = np.random.default_rng(seed)
    out = np.empty(n * length, np.uint8)
    code = np.zeros(256, np.uint8)
    code[ord("C")] = 1; code[ord("G")] = 2; code[ord("T")] = 3
    alpha = np.frombuffer(b"ACGT", np.uint8)
    step = 1 << 20
    for lo in range(0, n, step):
        m = min(step, n - lo)
        starts = rng.integers(0, len(ref) - length - 8, size=m, dtype=np.int64)
        if indel_p:
            # column j of a read copies reference offset j + shift[j]: +k after a deletion of k ref bases, and an
            # insertion of k read bases re-reads k reference bases later on (the inserted bases are then randomised)
            has = rng.random(m) < indel_p
            k = rng.integers(1, 6, size=m) * has
            at = rng.integers(10, length - 10, size=m)
            dele = rng.random(m) < 0.5
            cols = np.arange(length, dtype=np.int64)[None, :]
            after = cols >= at[:, None]
            shift = np.where(dele[:, None], after * k[:, None], -np.minimum(np.maximum(cols - at[:, None], 0), k[:, None]))
            blk = ref[starts[:, None] + cols + shift]
            ins = (~dele[:, None]) & (cols >= at[:, None]) & (cols < (at + k)[:, None])
            blk[ins] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=int(ins.sum()), dtype=np.uint8)]
        else:
            blk = ref[starts[:, None] + np.arange(length, dtype=np.int64)[None, :]]
        if max_subs:
            k = rng.integers(0, max_subs + 1, size=m)
            for s in range(max_subs):
                rows = np.nonzero(k > s)[0]
                cols = rng.integers(0, length, size=len(rows))
                old = code[blk[rows, cols]]
                blk[rows, cols] = alpha[(old + rng.integers(1, 4, size=len(rows), dtype=np.uint8)) & 3]
        out[lo * length:(lo + m) * length] = blk.reshape(-1)
    off = np.arange(0, (n + 1) * length, length, dtype=np.uint64)
    return out, off


def parity_sample(batch, ofm, ref1, bases, off, length, n=2000):
    """Hits of the LAST timed step for the first n reads, compared one by one with the CPU oracle."""
    from oracle import orc
    out = batch.fetch_hits()
    bad = 0
    for i in range(min(n, len(off) - 1)):
        q = bases[i * length:(i + 1) * length].tobytes()
        a, b = int(out["hit_off"][i]), int(out["hit_off"][i + 1])
        got = [(int(out["score"][h]), out["cigar"][int(out["cig_off"][h]):int(out["cig_off"][h + 1])].decode(),
                int(out["pos"][h])) for h in range(a, b)]
        try:
            exp = orc.align_read(ofm, ref1, q)
        except RuntimeError:
            exp = None
        want = [] if exp is None else [(s, c, p) for (s, c, p, _, _) in exp]
        bad += got != want
    return {"reads": min(n, len(off) - 1), "mismatching_reads": bad}


def cpu_baseline(ix, ref, bases, off, length, mode, budget_s=12.0, batch=None):
    """The oracle's C restatement (literal linear-scan se_down/se_up over the L and I columns,
    full-rectangle SW with the per-row abort) timed on this host's cores on a bounded sample of the
    same reads.  kind = "port": the reference is Erlang and cannot run here.  The index COLUMNS (L, SA)
    are exported from the GPU-built index; the I column and every step of the timed path are the oracle's."""
    from oracle import orc, erl_literal as E
    t0 = time.perf_counter()
    L, SA = ix.export()
    ptr, shift = ix.meta()
    ofm = orc.OrcFM.from_L_SA(L, SA, ptr, shift)
    ref1 = E.make_ref1(ref.tobytes() if isinstance(ref, np.ndarray) else ref)
    log("cpu baseline: index columns on the host in %.1f s" % (time.perf_counter() - t0))
    cores = os.cpu_count() or 1
    nmax = len(off) - 1
    n = min(nmax, 20000)
    reps, dt, done, res = 0, 0.0, 0, None
    while dt < budget_s and reps < 64:
        t0 = time.perf_counter()
        res = orc.batch(ofm, ref1, bases[: n * length], off[: n + 1].astype(np.uint32), mode, cores)
        d = time.perf_counter() - t0
        dt += d; done += n; reps += 1
        if d < budget_s / 8 and n < nmax:            # grow the sample until one pass is a good part of the budget
            n = min(nmax, n * 4)
    out = {"value": done / dt, "unit": "reads/s", "cores": cores, "kind": "port",
           "sample": "first %d reads of the bench batch (%d passes, %d reads in all), %s, %d threads, %.1f s" %
                     (n, reps, done, "sga (search+cluster)" if mode == 0 else "search+cluster+SW", cores, dt),
           "checksum": int(res[3])}
    if batch is not None and mode == 1:
        out["parity_sample"] = parity_sample(batch, ofm, ref1, bases, off, length)
        # at-scale parity: the oracle's aggregate over the whole CPU sample (reads with a hit, hits kept,
        # sum of 31*pos + score) against the same aggregate of the GPU's hit table for those reads
        hits = batch.fetch_hits()
        a, b = int(hits["hit_off"][0]), int(hits["hit_off"][n])
        gpu = [int((np.diff(hits["hit_off"][: n + 1].astype(np.int64)) > 0).sum()), b - a,
               int((hits["pos"][a:b].astype(np.int64) * 31 + hits["score"][a:b].astype(np.int64)).sum())]
        cpu = [int(res[0]), int(res[1]), int(res[3])]
        out["parity_aggregate"] = {"reads": n, "cpu": cpu, "gpu": gpu, "equal": cpu == gpu}
    return out


def host_pipeline(ix, read_sets, n_reads, depth=3, rounds=12):
    """The host-buffer boundary as a pipeline (bwt_pipe): reads in PINNED host memory in, hit tables in pinned host memory
    out, H2D + kernels + D2H of consecutive batches overlapped on `depth` slots.  Each slot's input buffers are filled once
    (a FASTQ parser would write there); every submit uploads them again.  Never `value`."""
    from bwt_amd import lib
    pipe = lib.Pipe(ix, depth=depth)
    nb = len(read_sets[0][0])
    filled = set()

    def submit_next():
        s, pb, po = pipe.acquire(nb, n_reads)
        if s not in filled:
            bases, off = read_sets[s % len(read_sets)]
            pb[:nb] = bases
            po[:] = off
            filled.add(s)
        pipe.submit(s, n_reads)

    def run(k):
        submitted = done = hits = 0
        while done < k:
            while submitted < k and submitted - done < depth:
                submit_next(); submitted += 1
            r = pipe.wait()
            hits += int(r["counts"]["n_hits"])
            pipe.release(r["slot"])
            done += 1
        return hits
    try:
        run(depth)                                         # warm-up: device and pinned buffers sized, inputs in place
        t0 = time.perf_counter()
        hits = run(rounds)
        dt = time.perf_counter() - t0
        return {"reads_per_s": rounds * n_reads / dt, "ms_per_batch": dt / rounds * 1e3, "depth": depth, "batches": rounds,
                "hits_per_batch": hits // rounds,
                "what": "pinned host reads in -> pinned host hit tables + CIGAR text out, H2D / kernels / D2H overlapped (bwt_pipe)"}
    finally:
        pipe.close()


def load_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/traffic.json), or None."""
    p = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(p):
        try:
            return json.load(open(p)).get(workload)
        except Exception:
            return None
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="grch38", choices=["grch38", "chr1", "paired150", "config2", "config2-align"])
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU per step (0 = the workload's own)")
    ap.add_argument("--genome-len", type=int, default=0, help="override the synthetic genome length")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-batch", action="store_true", help="one resident batch instead of two alternating ones")
    ap.add_argument("--no-pipeline", action="store_true", help="skip the host-buffer pipeline measurement")
    ap.add_argument("--overlap", action="store_true", help="also measure the two resident batches in flight at once (experiment)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)

    # stdout carries ONE JSON line: whatever libraries print there on the way (RCCL's version banner at communicator set-up)
    # goes to stderr instead
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the hot path has no CPU fallback", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dist = None
    force_gather = os.environ.get("BWT_BENCH_FORCE_GATHER") == "1"    # rehearse the N>1 exchange step on one GPU
    if world > 1 or force_gather:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ["MASTER_ADDR"] = "127.0.0.1"
            os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from bwt_amd import lib, shard
    length = 100
    wl = args.workload
    if wl.startswith("config2"):
        g = os.path.join(ROOT, "tests", "golden", "fm_indices", "GL000193.1_p1")
        ref = np.frombuffer(open(g + ".ref", "rb").read(), np.uint8)
        ix = lib.Index.open(g + ".fm", g + ".ref", local_rank)
        n_reads = args.reads or 1_000_000
        bases, off = synth_reads(ref, n_reads, length, seed=2 + rank, max_subs=0)
        stages = lib.STAGE_EXTEND if wl.endswith("align") else lib.STAGE_SEARCH
        desc = ("BASELINE configs[1]: bundled GL000193.1_p1 index, %d exact-match 100 bp reads per GPU, %s"
                % (n_reads, "whole path" if stages == lib.STAGE_EXTEND else "FM backward-search stage only"))
    else:
        if wl == "paired150":
            length = 150
        glen = args.genome_len or (230_000_000 if wl == "chr1" else 3_100_000_000)
        n_reads = args.reads or (10_000_000 if wl == "chr1" else 12_500_000)
        t0 = time.perf_counter()
        ref = synth_genome(glen, seed=3 if wl == "chr1" else 5)
        log("rank %d: synthetic genome of %d bp in %.1f s" % (rank, glen, time.perf_counter() - t0))
        t0 = time.perf_counter()
        ix = lib.Index.build(ref, shift=1, device=local_rank)
        log("rank %d: FM index built on the GPU in %.1f s (%s)" % (rank, time.perf_counter() - t0, ix.device_bytes()))
        t0 = time.perf_counter()
        bases, off = synth_reads(ref, n_reads, length, seed={"grch38": 5, "chr1": 4, "paired150": 6}[wl] * 1000 + rank,
                                 max_subs=2, indel_p=0.2 if wl == "paired150" else 0.0)
        log("rank %d: %d reads in %.1f s" % (rank, n_reads, time.perf_counter() - t0))
        stages = lib.STAGE_EXTEND
        desc = ("BASELINE configs[%d]: %s synthetic i.i.d. ACGT genome of %d bp (no GRCh38 FASTA offline), FM index "
                "(%d rows) built on the GPU and resident in HBM, %d reads of %d bp per GPU per step with 0/1/2 "
                "substitutions (p=1/3 each)%s, seed search + clustering + SW extension + CIGAR"
                % ({"grch38": 3, "chr1": 2, "paired150": 4}[wl], "chr1-scale" if wl == "chr1" else "whole-GRCh38-scale",
                   glen, glen + 1, n_reads, length,
                   " and one indel of 1-5 bp with p=0.2 (2x150 bp mates aligned independently, as the reference does)"
                   if wl == "paired150" else ""))

    # two distinct read batches, both resident in HBM before the timed region; the steps alternate between them
    batch = lib.Batch(ix)
    batch.set_reads(bases, off)
    batches = [batch]
    if not args.single_batch:
        if wl.startswith("config2"):
            bases2, off2 = synth_reads(ref, n_reads, length, seed=1002 + rank, max_subs=0)
        else:
            bases2, off2 = synth_reads(ref, n_reads, length, seed={"grch38": 5, "chr1": 4, "paired150": 6}[wl] * 1000 + 500 + rank,
                                       max_subs=2, indel_p=0.2 if wl == "paired150" else 0.0)
        b2 = lib.Batch(ix)
        b2.set_reads(bases2, off2)
        batches.append(b2)
    step_no = [0]
    gather = (world > 1 or force_gather) and stages == lib.STAGE_EXTEND
    lo, _ = shard.slab(n_reads * world, world, rank)
    gatherer = shard.HitGatherer(dist, rank, world, torch.device("cuda", local_rank)) if gather else None
    gathered = [0]

    def step():
        batch = batches[step_no[0] % len(batches)]
        step_no[0] += 1
        c = batch.run(stages)
        if gather:
            # RCCL hit gather (the path's one exchange step): 16-byte records to rank 0, double-buffered so that the gather of
            # this step overlaps the next step's kernels; the library writes the records (global read ids) straight into the
            # gatherer's send buffer
            nh = c["n_hits"]
            parts, buf = gatherer.begin(nh)
            batch.hits_to_device_global(buf.data_ptr(), buf.shape[0], lo)
            gatherer.commit()
            if rank == 0 and parts is not None:
                gathered[0] = int(sum(p.shape[0] for p in parts))
        return c

    def drain():
        if gather:
            parts = gatherer.wait()
            if rank == 0 and parts is not None:
                gathered[0] = int(sum(p.shape[0] for p in parts))

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # untimed set-up: every resident batch runs once so that its device buffers exist at their final sizes (hipMalloc, the
    # CIGAR-slot retry) before anything is timed -- the W warm-up steps alone would leave the second batch cold
    for b_ in batches:
        b_.run(stages)
    for _ in range(args.warmup):
        step()
    drain()
    for b_ in batches:
        b_.profile(True)                                # hipEvents around every kernel, on the library's own streams
    step_no[0] = 0
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        counts = step()
    drain()                                             # the last gather is inside the timed region
    sync()
    dt = time.perf_counter() - t0
    if gather and rank == 0:
        counts["gathered_hits"] = gathered[0]
    kt = {}
    for b_ in batches:
        for k, v in b_.kernel_times().items():
            a = kt.setdefault(k, {"ms": 0.0, "launches": 0})
            a["ms"] += v["ms"]; a["launches"] += v["launches"]
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        value = n_reads * world * args.steps / dt
        per = {k: v["ms"] / v["launches"] for k, v in kt.items()}           # avg ms per launch
        dom = max(per, key=lambda k: kt[k]["ms"])
        # Per-kernel rooflines (algorithmic bytes per launch, SURVEY 8d terms, over the launch time from hipEvents):
        #   fm_lookup   k_seed_tasks: Lq per read in (plane windows), one table entry per task (64 B when the top level
        #               carries the SA values, else 8 B), 4 B per task out (sub-read table) + the rows of the reads it does
        #               not cluster itself (16 B per task), 8 B per cluster and 16 B per read out
        #   fm_search   128 B per LF step (two 64-B rank blocks) [+ the read/result terms when there is no table]
        #   cluster     4 B per SA entry gathered, 16 B per task-table row in, 8 B per cluster out
        #   sw_extend   integer VALU (not HBM, not MFMA): 7.4 lane-ops per DP cell against 256 CU x 4 SIMD x 16 lanes x 2.4 GHz
        traffic = load_traffic(wl) or {}
        total_ms = sum(v["ms"] for v in kt.values())
        io_terms = counts["n_reads"] * length + 16 * counts["n_tasks"]
        kdepth, kwide = ix.ktable
        unfused = counts["n_small"] / max(counts["n_reads"], 1) if kwide else 1.0
        fast = "fm_lookup_rest" in per                     # k_seed_fast took the ordinary groups, k_seed_tasks the handed-back ones
        rw_ = 64 // max(1, length // 8)
        back = min(1.0, counts.get("n_handed_back", 0) * rw_ / max(counts["n_reads"], 1)) if fast else 1.0
        # (of the groups k_seed_fast hands back, its two-register pass over that list keeps most; the general kernel takes the rest)
        back2 = min(back, counts.get("n_general_groups", counts.get("n_handed_back", 0)) * rw_ / max(counts["n_reads"], 1)) if fast else 1.0
        narrow = counts["n_reads"] * sum(1 for k in range(length // 8) if 8 <= length - 8 * k < kdepth) if kdepth else 0
        wide_l = max(0, counts["n_lookups"] - narrow) if kwide else 0
        narrow_l = counts["n_lookups"] - wide_l
        # seed stage, both kernels: Lq per read in, one table entry per task (64 B / 8 B), 16 B per read (r_pend, r_P, r_qty) + 4 B (r_K)
        # + 8 B per cluster out; the general kernel also writes the sub-read table (4 B per task) and the rows of the reads it
        # does not cluster itself (16 B per task)
        # (the fast kernel writes r_qty and r_K only -- the host clears r_pend / r_P before it: 8 B per read, not 20)
        seed_common = counts["n_reads"] * (length + (8 if fast else 20)) + 64 * wide_l + 8 * narrow_l + 8 * counts["n_clusters"]
        general_extra = 4 * counts["n_tasks"] + int(16 * counts["n_tasks"] * unfused)
        models = {
            "fm_lookup": (("k_seed_fast", int(seed_common * (1.0 - back))) if fast else ("k_seed_tasks", seed_common + general_extra)),
            "fm_lookup_big": ("k_seed_fast_list", int(seed_common * (back - back2))),
            "fm_lookup_rest": ("k_seed_tasks", int((seed_common + general_extra) * back2)),
            "fm_search": ("k_search_pending" if "fm_lookup" in per else "k_search",
                          128 * counts["lf_steps"] + (0 if "fm_lookup" in per else io_terms)),
            "cluster_small": ("k_cluster_small", 4 * counts["sa_reads"] + 16 * counts["n_tasks"] + 8 * counts["n_clusters"]),
        }
        roofs = {}
        for key, (kname, nbytes) in models.items():
            if key in per and per[key] > 0:
                gbs = nbytes / (per[key] / 1e3) / 1e9
                roofs[key] = {"kernel": kname, "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": gbs / HBM_PEAK_GBS, "traffic": traffic.get(kname), "bytes_per_launch": nbytes,
                              "avg_launch_ms": per[key], "share_of_step": kt[key]["ms"] / total_ms}
        if "fm_lookup" in roofs:
            # random 64-byte reads are rated per line (profiles/r1/rand_access_microbench.txt: 47-55 G lines/s on this part,
            # 4..16 bytes used per line or all 64 alike): one table line per task
            glines = counts["n_lookups"] * ((1.0 - back) if fast else 1.0) / (per["fm_lookup"] / 1e3) / 1e9
            roofs["fm_lookup"]["random_lines"] = {"achieved_Glines_per_s": glines, "measured_ceiling_Glines_per_s": 47.0,
                                                  "frac": glines / 47.0}
        # (cells of the full-rectangle kernels: the DP candidates the banded kernel did not settle)
        dp_left = counts["n_dp"] - counts.get("n_banded", 0)
        dp_cells = counts["sw_cells"] * dp_left / max(counts["n_clusters"], 1)
        if "sw_extend" in per and dp_left > 32768:
            ops_per_cell = 7.4 if "sw_trace_pk" in per else 9.4      # score-only pass / the traced DP run on the whole DP queue
            ops = dp_cells * ops_per_cell / (per["sw_extend"] / 1e3) / 1e12
            roofs["sw_extend"] = {"kernel": "k_sw_pk", "bound": "valu", "achieved": ops, "peak": VALU_PEAK_TOPS,
                                  "unit": "Tlane-op/s", "frac": ops / VALU_PEAK_TOPS, "traffic": traffic.get("k_sw"),
                                  "cells_per_launch": dp_cells, "valu_lane_ops_per_cell": ops_per_cell, "avg_launch_ms": per["sw_extend"],
                                  "share_of_step": kt["sw_extend"]["ms"] / total_ms}
        if "sw_band" in per and counts.get("n_banded", 0) > 0:
            # banded DP: 20 cells per row in the first tier (nearly everything it settles is settled there), 17.6 instructions per
            # cell PAIR (two candidates per lane) = 8.8 lane-ops per cell
            bcells = counts["n_banded"] * length * 20.0
            ops = bcells * 8.8 / (per["sw_band"] / 1e3) / 1e12
            roofs["sw_band"] = {"kernel": "k_sw_band", "bound": "valu", "achieved": ops, "peak": VALU_PEAK_TOPS, "unit": "Tlane-op/s",
                                "frac": ops / VALU_PEAK_TOPS, "traffic": traffic.get("k_sw_band"), "cells_per_launch": bcells,
                                "valu_lane_ops_per_cell": 8.8, "avg_launch_ms": per["sw_band"], "share_of_step": kt["sw_band"]["ms"] / total_ms,
                                "settled": counts["n_banded"], "of_dp_candidates": counts["n_dp"]}
        roof = roofs.get(dom) or roofs[max(roofs, key=lambda k: kt[k]["ms"])]
        line = {
            "metric": "aligned reads/sec (100 bp, GRCh38 FM-index) at 1/2/4/8 MI355X",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": desc, "reads_per_gpu": n_reads, "read_len": length,
                       "index_rows": ix.meta()[0][3], "index_bytes": ix.device_bytes(),
                       "suffix_interval_table": {"K": ix.ktable[0], "top_level_carries_sa": ix.ktable[1]},
                       "parallelism": "reads sharded x%d (contiguous slabs), index replicated, RCCL hit gather" % world},
            "roofline": roof,
            "rooflines": roofs,
            "dominant_kernel": dom,
            "kernels_ms_per_step": {k: v["ms"] / args.steps for k, v in sorted(kt.items())},
            "counts_per_step": counts,
        }
        if "sw_extend" in per or "sw_wave" in per:
            banded = counts.get("n_banded", 0) > 0
            left = counts["n_dp"] - counts.get("n_banded", 0)
            t_dp = ((per.get("sw_band", 0.0) if banded else 0.0) + (per.get("sw_wave", 0.0) if left <= 32768 else per.get("sw_extend", 0.0))) / 1e3
            line["extend"] = {"candidates": counts["n_clusters"],
                              "settled_by_prefilter": counts["n_clusters"] - counts["n_dp"] - counts["n_float"],
                              "dp_candidates": counts["n_dp"],
                              "settled_by_banded_dp": counts.get("n_banded", 0),
                              "dp_kernel": ("k_sw_band (20, then 32 diagonals + certificate), then " if banded else "") +
                                           ("k_sw_wave (wave per alignment: short queue)" if left <= 32768 else "k_sw_pk"),
                              # cells of the reference's full rectangles for these candidates, per second of DP kernels (the banded
                              # kernel fills 24 diagonals of them)
                              "dp_cells_per_s": counts["sw_cells"] * counts["n_dp"] / max(counts["n_clusters"], 1) / t_dp if t_dp else None}
        # PCIe-inclusive rate (never `value`): host buffers in, host hit tables out, one untimed-extra pass
        t0 = time.perf_counter()
        batch.set_reads(bases, off)
        batch.run(stages)
        if stages == lib.STAGE_EXTEND:
            batch.fetch_hits()
        line["pcie_inclusive_reads_per_s"] = n_reads / (time.perf_counter() - t0)
        if len(batches) > 1 and args.overlap:
            # experiment (never `value`): the two resident batches in flight at once, each on its own stream and host thread
            import threading
            for b_ in batches:
                b_.profile(False)
            def runner(b_, k):
                for _ in range(k):
                    b_.run(stages)
            k_each = max(1, args.steps // 2)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            th = [threading.Thread(target=runner, args=(b_, k_each)) for b_ in batches]
            for t_ in th:
                t_.start()
            for t_ in th:
                t_.join()
            torch.cuda.synchronize()
            dt2 = time.perf_counter() - t0
            line["two_batches_in_flight"] = {"reads_per_s": n_reads * k_each * len(batches) / dt2, "ms_per_step": dt2 / (k_each * len(batches)) * 1e3,
                                             "what": "both resident batches run concurrently (own stream + host thread each)"}
        if stages == lib.STAGE_EXTEND and not args.no_pipeline:
            line["host_pipeline"] = host_pipeline(ix, [(bases, off)] + ([(bases2, off2)] if len(batches) > 1 else []), n_reads)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(ix, ref, bases, off, length, 0 if stages == lib.STAGE_SEARCH else 1,
                                                batch=batch)
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(json.dumps(line), flush=True)
        os.dup2(2, 1)
    batch.destroy()
    ix.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
