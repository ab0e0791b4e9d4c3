#!/usr/bin/env python3
"""What the host link carries: one pinned 312.5 MB buffer (a 12.5 M-read batch of packed 100-bp reads) host -> device, as one copy
and split over 2 / 4 streams; alone and with a device -> host copy of 205 MB (a batch's hit table) running beside it.
    python scripts/h2d_probe.py"""
import json
import time

import torch

N = 312_500_000
D = 205_000_000
dev = torch.device("cuda", 0)
src = torch.empty(N, dtype=torch.uint8).pin_memory()
dst = torch.empty(N, dtype=torch.uint8, device=dev)
dsrc = torch.empty(D, dtype=torch.uint8, device=dev)
ddst = torch.empty(D, dtype=torch.uint8).pin_memory()
streams = [torch.cuda.Stream(dev) for _ in range(4)]
down = torch.cuda.Stream(dev)
out = {}
for parts in (1, 2, 4):
    for with_down in (False, True):
        best = 1e9
        for rep in range(6):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if with_down:
                with torch.cuda.stream(down):
                    ddst.copy_(dsrc, non_blocking=True)
            step = N // parts
            for p in range(parts):
                with torch.cuda.stream(streams[p]):
                    dst[p * step:(p + 1) * step].copy_(src[p * step:(p + 1) * step], non_blocking=True)
            for p in range(parts):
                streams[p].synchronize()
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            best = min(best, t1 - t0)
        out["h2d_%d_streams%s" % (parts, "_with_d2h" if with_down else "")] = {"ms": round(best * 1e3, 3), "GB_per_s": round(N / best / 1e9, 2)}
print(json.dumps(out, indent=1))
