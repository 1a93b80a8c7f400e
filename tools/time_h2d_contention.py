#!/usr/bin/env python3
"""What slows the reader's PCIe copies down when the whole pipeline runs?  (3-Gb `tiddit --sv`: the spans' host-to-device copies sum to
1.0 s in a process's first pass, where the file reads are slow, and to 1.65-1.73 s in warm passes — 32 GB/s instead of 54 — which is
then as long as the inflate kernels.)  A pinned ring like bamio.DeviceBamReader's, filled by N threads with os.preadv from a file in the
page cache, while the previous span crosses PCIe:
    python tools/time_h2d_contention.py [file_MB=4096] [span_MB=448]
prints the copy rate alone, beside 1..32 reading threads, beside reading threads pinned to one NUMA node or the other, and the topology
(GPU's NUMA node, nodes' CPUs)."""
import glob
import os
import sys
import threading
import time
from concurrent.futures import ThreadPoolExecutor

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch


def topology():
    out = {}
    for p in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
        try:
            out[p.split("/")[4]] = open(p).read().strip()
        except OSError:
            pass
    nodes = {}
    for p in sorted(glob.glob("/sys/devices/system/node/node*/cpulist")):
        nodes[p.split("/")[5]] = open(p).read().strip()
    return out, nodes


def parse_cpulist(s):
    cpus = []
    for part in s.split(","):
        if "-" in part:
            a, b = part.split("-")
            cpus.extend(range(int(a), int(b) + 1))
        elif part:
            cpus.append(int(part))
    return cpus


def main():
    file_mb = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    span_mb = int(sys.argv[2]) if len(sys.argv) > 2 else 448
    gpus, nodes = topology()
    print("GPU numa nodes:", gpus)
    print("nodes:", {k: v[:60] for k, v in nodes.items()})
    print("process affinity: %d cpus" % len(os.sched_getaffinity(0)))
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "h2d_contention.bin")
    span = span_mb << 20
    nsp = max(2, (file_mb << 20) // span)
    if not os.path.exists(path) or os.path.getsize(path) != nsp * span:
        blk = np.random.default_rng(1).integers(0, 256, span, dtype=np.uint8).tobytes()
        with open(path, "wb") as f:
            for _ in range(nsp):
                f.write(blk)
    fd = os.open(path, os.O_RDONLY)
    for i in range(nsp):                                     # page cache warm
        os.pread(fd, span, i * span)
    dev = torch.device("cuda:0")
    ring = [torch.empty(span, dtype=torch.uint8).pin_memory() for _ in range(3)]
    views = [memoryview(r.numpy()) for r in ring]
    dbuf = [torch.empty(span, dtype=torch.uint8, device=dev) for _ in range(2)]
    cs = torch.cuda.Stream(device=dev)

    def fill(pool, nthreads, i, k):
        mv = views[k]
        piece = (span + nthreads - 1) // nthreads
        piece = (piece + 4095) & ~4095

        def one(o):
            n, end = 0, min(o + piece, span)
            while o + n < end:
                g = os.preadv(fd, [mv[o + n:end]], (i % nsp) * span + o + n)
                if g <= 0:
                    break
                n += g
        list(pool.map(one, range(0, span, piece)))

    def copy_alone(reps=12):
        ts = []
        for r in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(cs):
                a.record(cs)
                dbuf[r & 1].copy_(ring[r % 3], non_blocking=True)
                b.record(cs)
            cs.synchronize()
            ts.append(a.elapsed_time(b))
        return span / (np.median(ts) * 1e-3) / 1e9

    def pipeline(nthreads, affinity=None, spans=16):
        """reader fills ring[k+1] while ring[k] is copied; -> (copy GB/s from events, read GB/s, wall GB/s)"""
        def init():
            if affinity:
                os.sched_setaffinity(0, affinity)
        pool = ThreadPoolExecutor(nthreads, initializer=init)
        fill(pool, nthreads, 0, 0)
        copies, reads = [], []
        t_wall = time.perf_counter()
        for i in range(spans):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            with torch.cuda.stream(cs):
                a.record(cs)
                dbuf[i & 1].copy_(ring[i % 3], non_blocking=True)
                b.record(cs)
            t0 = time.perf_counter()
            fill(pool, nthreads, i + 1, (i + 1) % 3)
            reads.append(time.perf_counter() - t0)
            cs.synchronize()
            copies.append(a.elapsed_time(b) * 1e-3)
        wall = time.perf_counter() - t_wall
        pool.shutdown()
        return span / np.median(copies) / 1e9, span / np.median(reads) / 1e9, spans * span / wall / 1e9

    print("copy alone: %.1f GB/s" % copy_alone())
    for n in (1, 2, 4, 8, 16, 32):
        c, r, w = pipeline(n)
        print("readers %2d: copy %.1f GB/s, read %.1f GB/s, pipeline %.1f GB/s" % (n, c, r, w))
    for name, cl in nodes.items():
        cpus = set(parse_cpulist(cl)) & os.sched_getaffinity(0)
        if not cpus:
            continue
        for n in (4, 16):
            c, r, w = pipeline(n, cpus)
            print("readers %2d on %s: copy %.1f GB/s, read %.1f GB/s, pipeline %.1f GB/s" % (n, name, c, r, w))
    # the pinned ring allocated while the process is bound to each node (first touch decides where the pages live)
    full = os.sched_getaffinity(0)
    for name, cl in nodes.items():
        cpus = set(parse_cpulist(cl)) & full
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        ring[:] = [torch.empty(span, dtype=torch.uint8).pin_memory() for _ in range(3)]
        views[:] = [memoryview(r.numpy()) for r in ring]
        for v in views:
            np.frombuffer(v, dtype=np.uint8)[::4096] = 1
        os.sched_setaffinity(0, full)
        print("ring pinned from %s: copy alone %.1f GB/s" % (name, copy_alone()))
        for n in (4, 16):
            c, r, w = pipeline(n)
            print("   readers %2d: copy %.1f GB/s, read %.1f GB/s, pipeline %.1f GB/s" % (n, c, r, w))
    os.close(fd)
    os.unlink(path)


if __name__ == "__main__":
    main()
