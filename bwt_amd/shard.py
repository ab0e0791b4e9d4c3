"""Multi-GPU sharding of the read stream: reads are independent units, the index is
replicated on every GPU, each rank aligns one contiguous slab, and the only exchange is the
gather of the compact hit tables -- fixed-size hit records {read:u32, score:i32, pos:i64} (16 B) plus, for the hits that are
not plain, window length and CIGAR text -- to rank 0 (TableGatherer; HitGatherer / gather_hits carry the records alone).

The reference shards the same way -- every worker_bwt holds the whole index and pulls batches of
reads from the master (/root/reference/src/master.erl:123,140-147, src/worker_bwt.erl:46-48).
Backend-agnostic: RCCL ("nccl") on GPUs, "gloo" in the CPU tests."""
import torch


def slab(n_total, world, rank):
    """Contiguous slab [lo, hi) of rank `rank`; global read id = lo + local index."""
    lo = (n_total * rank) // world
    hi = (n_total * (rank + 1)) // world
    return lo, hi


def pack_records(read, score, pos):
    """-> int64 tensor [k, 2]: word0 = read | score << 32, word1 = pos (the HitRec layout of the library)."""
    w0 = (read.to(torch.int64) & 0xFFFFFFFF) | (score.to(torch.int64) << 32)
    return torch.stack([w0, pos.to(torch.int64)], dim=1).contiguous()


def unpack_records(rec):
    w0 = rec[:, 0]
    read = w0 & 0xFFFFFFFF
    score = w0 >> 32
    return read, score, rec[:, 1]


def globalize(rec, lo):
    """Turn slab-local read ids into global ones."""
    out = rec.clone()
    out[:, 0] = (rec[:, 0] & ~0xFFFFFFFF) | ((rec[:, 0] & 0xFFFFFFFF) + lo)
    return out


def gather_hits(dist, rec, rank, world, dst=0):
    """Variable-length gather of hit records to `dst`: all_gather of the counts, then a gather of
    max-padded buffers (one collective each).  Returns on dst the list of per-rank tensors (in rank
    order = global read order), elsewhere None."""
    dev = rec.device
    n = torch.tensor([rec.shape[0]], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    counts = [int(c.item()) for c in counts]
    mx = max(max(counts), 1)
    buf = torch.zeros((mx, 2), dtype=torch.int64, device=dev)
    buf[: rec.shape[0]] = rec
    out = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return [out[r][: counts[r]] for r in range(world)]


def _round16(x):
    return (int(x) + 15) // 16 * 16


def _host_sync(device):
    """After work.wait(): under RCCL that call only makes torch's CURRENT STREAM wait for the collective -- the host goes on.
    The gatherers' buffers are written by the library on ITS OWN stream (bwt_batch_compact_to_device) and read by consumers
    that may not be on torch's stream either, so a slot may only be handed back to the producer (or its recv views handed
    out) once the HOST knows the collective is done: block on the stream that now carries the wait.  (gloo's wait() blocks the
    host by itself; there is nothing to do for a CPU device.)"""
    if torch.device(device).type == "cuda":
        torch.cuda.current_stream(device).synchronize()


class TableGatherer:
    """Asynchronous, double-buffered gather of COMPACT HIT TABLES (include/bwt_gpu.h) to rank `dst`: everything the worker's cast
    {cigar, Name, {Score, Cigar}, Pos, RefSeq} needs (src/worker_bwt.erl:100-105) -- 16-byte hit records with global read ids,
    and for the hits that are not plain their window length and CIGAR text (extras).  A plain hit's CIGAR is "<Lq>M" on a window
    of Lq bases; RefSeq is Ref1[pos - Shift + 200, win_len) on any replica of the index.

    Per batch and rank ONE collective on the device: a gather of one byte buffer [hits | extras | text], the three sections at
    capacities every rank computes alike.  The counts that size them are exchanged on the HOST (a gloo group next to RCCL:
    the counts are host integers as soon as a batch's kernels are done, so no rank waits for its GPU to learn them -- round 2's
    all_gather on the device followed by .tolist() was a stream synchronisation per step).  `begin(counts)` hands out the send
    buffer's three sections for the producer to fill (bwt_batch_compact_to_device: device to device, on the batch's stream);
    `commit()` starts the gather; results come one batch late (`begin` returns the previous batch's, `wait()` the last one's):
    on dst a list of per-rank (hits [k,2] int64, extras [x,4] int32, text uint8) views in rank order, None elsewhere.
    Buffers are allocated with torch.empty -- nothing is ever launched on them by torch (round 2 zero-filled them on torch's
    stream, unordered with the library's stream: ADVICE r2)."""

    def __init__(self, dist, rank, world, device, cpu_group=None, dst=0):
        self.dist, self.rank, self.world, self.device, self.dst = dist, rank, world, device, dst
        self.cpu_group = cpu_group
        self.caps = [0, 0, 0]          # hits, extras (records), text (bytes, multiple of 16)
        self.send = [None, None]
        self.recv = [None, None]
        self.pending = None            # (slot, work handle, counts per rank)
        self.cur = None
        self.k = 0

    def _nbytes(self):
        return 16 * self.caps[0] + 16 * self.caps[1] + self.caps[2]

    def _alloc(self):
        nb = max(self._nbytes(), 16)
        for s in (0, 1):
            self.send[s] = torch.empty(nb, dtype=torch.uint8, device=self.device)
            if self.rank == self.dst:
                self.recv[s] = [torch.empty(nb, dtype=torch.uint8, device=self.device) for _ in range(self.world)]

    def _sections(self, buf, counts=None):
        ch, cx, cc = self.caps
        nh, nx, nc = (ch, cx, cc) if counts is None else counts
        hits = buf[: 16 * ch].view(torch.int64).view(-1, 2)[:nh]
        extra = buf[16 * ch: 16 * (ch + cx)].view(torch.int32).view(-1, 4)[:nx]
        text = buf[16 * (ch + cx): 16 * (ch + cx) + cc][:nc]
        return hits, extra, text

    def begin(self, n_hits, n_extra, cig_bytes):
        """-> (result of the previous batch, (hits, extras, text) sections of this batch's send buffer, at full capacity)."""
        prev = self.wait()                                   # (also: both slots are idle, so they may be reallocated)
        mine = torch.tensor([int(n_hits), int(n_extra), int(cig_bytes)], dtype=torch.int64)
        allc = [torch.zeros(3, dtype=torch.int64) for _ in range(self.world)]
        self.dist.all_gather(allc, mine, group=self.cpu_group)                 # host integers over the host group: no device sync
        counts = [tuple(int(v) for v in c.tolist()) for c in allc]
        need = [max(c[j] for c in counts) for j in range(3)]
        if any(need[j] > self.caps[j] for j in range(3)) or self.send[0] is None:
            # the same decision on every rank; 1/16 head-room because the padding travels too
            self.caps = [max(self.caps[0], need[0] * 17 // 16 + 1024), max(self.caps[1], need[1] * 17 // 16 + 1024),
                         max(self.caps[2], _round16(need[2] * 17 // 16 + 1024))]
            self._alloc()
        s = self.k & 1
        self.k += 1
        self.cur = (s, counts)
        return prev, self._sections(self.send[s])

    def commit(self):
        s, counts = self.cur
        self.cur = None
        w = self.dist.gather(self.send[s], self.recv[s] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.pending = (s, w, counts)

    def submit(self, hits, extra, text):
        """Tables that already sit in tensors ([k,2] int64 with global read ids, [x,4] int32, uint8).  Returns the PREVIOUS result."""
        prev, (h, x, t) = self.begin(hits.shape[0], extra.shape[0], text.shape[0])
        h[: hits.shape[0]] = hits
        x[: extra.shape[0]] = extra
        t[: text.shape[0]] = text
        self.commit()
        return prev

    def wait(self):
        """Finish the outstanding gather, if any, and return its result."""
        if self.pending is None:
            return None
        s, work, counts = self.pending
        self.pending = None
        work.wait()
        _host_sync(self.device)
        if self.rank != self.dst:
            return None
        return [self._sections(self.recv[s][r], counts[r]) for r in range(self.world)]


def merge_tables(parts):
    """The per-rank sections of a gathered batch as ONE compact table in global read order: hit records concatenated (their
    read ids are global already), extras with the sender's hit base and text base added.  -> (hits [k,2], extras [x,4], text)."""
    hits = torch.cat([p[0] for p in parts])
    text = torch.cat([p[2] for p in parts])
    ex, hb, tb = [], 0, 0
    for h, x, t in parts:
        x = x.clone()
        x[:, 0] += hb
        x[:, 2] += tb
        ex.append(x)
        hb += h.shape[0]; tb += t.shape[0]
    return hits, torch.cat(ex), text


class HitGatherer:
    """Asynchronous, double-buffered form of gather_hits for a steady stream of batches: the gather of batch b runs on the
    collective's own stream while batch b+1 is being computed.

    Per batch: one small all_gather of the hit counts (every rank learns every count, so every rank takes the same decisions:
    the padded capacity grows when a count outgrows it, +1/16 head-room because the padding travels too -- at ~7 ms per
    12.5 M-read batch rank 0 takes in more than 150 GB/s), then one asynchronous gather of the padded send buffers.
    `begin(n)` hands out the send buffer of the batch so that the producer writes its records straight into it
    (bwt_batch_hits_to_device_global: device-to-device, read ids made global on the way); `commit()` starts the gather;
    `submit(rec)` is both for records that already sit in a tensor.  Results come one batch late (`begin` / `submit` return
    the previous batch's, `wait()` the last one's): on dst the list of per-rank tensors in rank order, None elsewhere."""

    def __init__(self, dist, rank, world, device, dst=0):
        self.dist, self.rank, self.world, self.device, self.dst = dist, rank, world, device, dst
        self.cap = 0
        self.send = [None, None]
        self.recv = [None, None]
        self.cnt_in = torch.zeros(1, dtype=torch.int64, device=device)
        self.cnt_out = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
        self.pending = None          # (slot, work handle, counts)
        self.cur = None              # (slot, counts) between begin and commit
        self.k = 0

    def _grow(self, need):
        self.cap = max(need * 17 // 16 + 1024, 1)
        for s in (0, 1):
            self.send[s] = torch.empty((self.cap, 2), dtype=torch.int64, device=self.device)   # (not zeros: nothing of torch's may run on it later than the library's writes)
            if self.rank == self.dst:
                self.recv[s] = [torch.empty((self.cap, 2), dtype=torch.int64, device=self.device) for _ in range(self.world)]

    def begin(self, n):
        """-> (result of the previous batch, send buffer [cap, 2] int64 to fill with this batch's n records)."""
        prev = self.wait()                                   # (also: both slots are idle, so they may be reallocated)
        self.cnt_in[0] = n
        self.dist.all_gather(self.cnt_out, self.cnt_in)
        counts = torch.cat(self.cnt_out).tolist()            # one synchronisation
        if max(counts) > self.cap:
            self._grow(max(counts))                          # the same decision on every rank
        s = self.k & 1
        self.k += 1
        self.cur = (s, counts)
        return prev, self.send[s]

    def commit(self):
        s, counts = self.cur
        self.cur = None
        w = self.dist.gather(self.send[s], self.recv[s] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.pending = (s, w, counts)

    def submit(self, rec):
        """Start gathering `rec` ([k,2] int64, global read ids).  Returns the result of the PREVIOUS submit."""
        n = rec.shape[0]
        prev, buf = self.begin(n)
        buf[:n] = rec
        self.commit()
        return prev

    def wait(self):
        """Finish the outstanding gather, if any, and return its result."""
        if self.pending is None:
            return None
        s, work, counts = self.pending
        self.pending = None
        work.wait()
        _host_sync(self.device)
        if self.rank != self.dst:
            return None
        return [self.recv[s][r][: counts[r]] for r in range(self.world)]
