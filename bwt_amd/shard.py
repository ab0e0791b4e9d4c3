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
