"""Multi-GPU sharding of the read stream: reads are independent units, the index is
replicated on every GPU, each rank aligns one contiguous slab, and the only exchange is the
gather of the fixed-size hit records {read:u32, score:i32, pos:i64} (16 B) to rank 0.

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


class HitGatherer:
    """Asynchronous, double-buffered form of gather_hits for a steady stream of batches: the gather of
    batch b runs on the collective's own stream while batch b+1 is being computed.  All ranks agree once on a
    padded capacity (all_reduce MAX of the first batch's hit count, +1/16: the padding travels too, and at ~10 ms
    per 12.5 M-read batch rank 0 takes in more than 100 GB/s); a batch that outgrows it falls back to the synchronous path."""

    def __init__(self, dist, rank, world, device, dst=0):
        self.dist, self.rank, self.world, self.device, self.dst = dist, rank, world, device, dst
        self.cap = 0
        self.send = [None, None]
        self.recv = [None, None]
        self.cnt_in = [None, None]
        self.cnt_out = [None, None]
        self.pending = None          # (slot, [work handles])
        self.k = 0

    def _agree_capacity(self, n):
        t = torch.tensor([n], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.cap = max(int(t.item()) * 17 // 16 + 1024, 1)
        for s in (0, 1):
            self.send[s] = torch.zeros((self.cap, 2), dtype=torch.int64, device=self.device)
            self.cnt_in[s] = torch.zeros(1, dtype=torch.int64, device=self.device)
            self.cnt_out[s] = [torch.zeros(1, dtype=torch.int64, device=self.device) for _ in range(self.world)]
            if self.rank == self.dst:
                self.recv[s] = [torch.empty((self.cap, 2), dtype=torch.int64, device=self.device) for _ in range(self.world)]

    def submit(self, rec):
        """Start gathering `rec` ([k,2] int64, global read ids).  Returns the result of the PREVIOUS submit
        (list of per-rank tensors on dst, None elsewhere, None for the very first call)."""
        n = rec.shape[0]
        if self.cap == 0:
            self._agree_capacity(n)
        prev = self.wait()
        over = torch.tensor([1 if n > self.cap else 0], dtype=torch.int64, device=self.device)
        self.dist.all_reduce(over, op=self.dist.ReduceOp.MAX)          # every rank takes the same path
        if int(over.item()):
            self.sync_result = gather_hits(self.dist, rec, self.rank, self.world, self.dst)
            self.pending = ("sync", None)
            return prev
        s = self.k & 1
        self.k += 1
        self.send[s][:n] = rec
        self.cnt_in[s][0] = n
        w1 = self.dist.all_gather(self.cnt_out[s], self.cnt_in[s], async_op=True)
        w2 = self.dist.gather(self.send[s], self.recv[s] if self.rank == self.dst else None, dst=self.dst, async_op=True)
        self.pending = (s, [w1, w2])
        return prev

    def wait(self):
        """Finish the outstanding gather, if any, and return its result."""
        if self.pending is None:
            return None
        s, works = self.pending
        self.pending = None
        if s == "sync":
            return self.sync_result
        for w in works:
            w.wait()
        if self.rank != self.dst:
            return None
        counts = [int(c.item()) for c in self.cnt_out[s]]
        return [self.recv[s][r][: counts[r]] for r in range(self.world)]
