"""world_size-2 gloo test of the multi-GPU path's host logic (bwt_amd/shard.py): slabs cover the
read stream exactly once and rank 0 ends up with every hit record in global read order."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bwt_amd import shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard.slab(n_total, world, rank)
    # fake per-rank hits: read i yields (i % 3) hits; score/pos derived from the global id
    g = torch.Generator().manual_seed(1234)
    reads, scores, poss = [], [], []
    for i in range(lo, hi):
        for k in range(i % 3):
            reads.append(i - lo); scores.append(100 + (i * 7 + k) % 101); poss.append(i * 1000 - k)
    rec = shard.pack_records(torch.tensor(reads, dtype=torch.int64), torch.tensor(scores, dtype=torch.int64),
                             torch.tensor(poss, dtype=torch.int64)) if reads else torch.zeros((0, 2), dtype=torch.int64)
    rec = shard.globalize(rec, lo)
    parts = shard.gather_hits(dist, rec, rank, world)
    if rank == 0:
        allrec = torch.cat(parts)
        r, s, p = shard.unpack_records(allrec)
        q.put((r.tolist(), s.tolist(), p.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_slabs_partition():
    for n in (0, 1, 7, 100, 1_000_003):
        for w in (1, 2, 3, 8):
            edges = [shard.slab(n, w, r) for r in range(w)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))


def test_gather_world2():
    world, n_total = 2, 101
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_r, exp_s, exp_p = [], [], []
    for i in range(n_total):
        for k in range(i % 3):
            exp_r.append(i); exp_s.append(100 + (i * 7 + k) % 101); exp_p.append(i * 1000 - k)
    assert got == (exp_r, exp_s, exp_p)


def _worker_async(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = shard.HitGatherer(dist, rank, world, torch.device("cpu"))
    results = []
    for step in range(4):                      # batch sizes change; step 3 outgrows the agreed capacity
        n = [5, 3, 6, 40][step] + rank
        rec = torch.stack([torch.arange(n, dtype=torch.int64) + 1000 * rank + 100000 * step,
                           torch.full((n,), step, dtype=torch.int64)], dim=1)
        prev = g.submit(rec)
        if prev is not None and rank == 0:
            results.append(torch.cat(prev).tolist())
    last = g.wait()
    if rank == 0:
        results.append(torch.cat(last).tolist())
        q.put(results)
    dist.barrier()
    dist.destroy_process_group()


def test_async_double_buffered_gather_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_async, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == 4
    for step, res in enumerate(got):
        exp = []
        for rank in range(world):
            n = [5, 3, 6, 40][step] + rank
            exp += [[i + 1000 * rank + 100000 * step, step] for i in range(n)]
        assert res == exp


def _fake_table(rank, step):
    """a rank's compact table of one step: k hits (global read ids), every third one with an extra record and text"""
    k = [5, 0, 9, 70][step] + 2 * rank
    hits = torch.stack([(torch.arange(k, dtype=torch.int64) // 2 + 1000 * rank) | ((150 + torch.arange(k, dtype=torch.int64)) << 32),
                        torch.arange(k, dtype=torch.int64) * 31 + step], dim=1)
    xs, text = [], b""
    for h in range(0, k, 3):
        t = ("%dM1D%dM" % (h + 1, 99 - h)).encode()
        xs.append([h, 101, len(text), len(t)])
        text += t
    extra = torch.tensor(xs, dtype=torch.int32).reshape(-1, 4)
    return hits, extra, torch.frombuffer(bytearray(text), dtype=torch.uint8) if text else torch.zeros(0, dtype=torch.uint8)


def _worker_tables(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = shard.TableGatherer(dist, rank, world, torch.device("cpu"))
    results = []

    def keep(parts):
        h, x, t = shard.merge_tables(parts)
        results.append((h.tolist(), x.tolist(), bytes(t.tolist())))
    for step in range(4):                      # sizes change from step to step; step 3 outgrows every agreed capacity
        prev = g.submit(*_fake_table(rank, step))
        if prev is not None and rank == 0:
            keep(prev)
    last = g.wait()
    if rank == 0:
        keep(last)
        q.put(results)
    dist.barrier()
    dist.destroy_process_group()


def test_compact_table_gather_world2():
    """TableGatherer: rank 0 reassembles, step by step, the compact table a single rank would have produced for the whole batch --
    records, extras (hit indices and text offsets rebased) and the CIGAR bytes -- with capacities that grow on the way."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_tables, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(got) == 4
    for step, (h, x, t) in enumerate(got):
        eh, ex, et, hb = [], [], b"", 0
        for rank in range(world):
            fh, fx, ft = _fake_table(rank, step)
            eh += fh.tolist()
            for (hit, wl, co, cl) in fx.tolist():
                ex.append([hit + hb, wl, co + len(et), cl])
            et += bytes(ft.tolist())
            hb += fh.shape[0]
        assert h == eh and x == ex and t == et
        # every extra's text is where it says
        for (hit, wl, co, cl) in x:
            assert t[co:co + cl].endswith(b"M") and wl == 101
