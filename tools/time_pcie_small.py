#!/usr/bin/env python3
"""PCIe rate of pinned-memory copies by size (the 20-MB columns of tdt_cluster_columns vs the ingest's 256-MB spans)."""
import torch, time
dev = torch.device("cuda", 0)
for mb in (1, 5, 20, 80, 256):
    n = mb << 20
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device=dev)
    for direction in ("h2d", "d2h"):
        ts = []
        for rep in range(8):
            torch.cuda.synchronize()
            t = time.perf_counter()
            if direction == "h2d": d.copy_(h, non_blocking=True)
            else: h.copy_(d, non_blocking=True)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t)
        best = min(ts[2:])
        print("%4d MB %s %.3f ms %.1f GB/s" % (mb, direction, best * 1e3, n / best / 1e9))
