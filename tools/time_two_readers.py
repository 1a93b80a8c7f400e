#!/usr/bin/env python3
"""Would two device readers in ONE process — each on its own library context (own streams), each over one half of the file's byte range —
finish a BAM sooner than one reader over all of it?  (One launch's persistent inflate waves leave the chip part empty in their last round
of blocks, and a reader's small kernels and host round trips sit between its launches: a second pipeline could fill both.)
    python tools/time_two_readers.py [Mb per contig = 120] [level = 1]
prints the wall of a pass (records into HBM, nothing consumed) with one reader and with two concurrent ones (threads), three rounds."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import _native, bamio, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 120
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "chunks_%d_%d.bam" % (mb, level))
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", mb * 1_000_000), ("chr2", mb * 1_000_000)], depth=30, threads=32, level=level, realistic=True)
size = os.path.getsize(path)
ctxs = [_native.default_context(), _native.Context(_native.default_context().device)]


def one_pass(ctx, shard, out):
    r = bamio.DeviceBamReader(path, ctx=ctx, shard=shard)
    n = sum(len(b) for b in r.batches())
    ctx.sync()
    r.close()
    out.append(n)


for rep in range(4):
    res = []
    t0 = time.perf_counter()
    one_pass(ctxs[0], None, res)
    t1 = time.perf_counter() - t0
    res2 = []
    t0 = time.perf_counter()
    th = [threading.Thread(target=one_pass, args=(ctxs[k], (k, 2), res2)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    t2 = time.perf_counter() - t0
    if rep:
        print("one reader: %d records in %.3f s (%.1f GB/s of BGZF) | two readers, halves of the file: %d records in %.3f s (%.1f GB/s) -> %.2fx" % (
            res[0], t1, size / t1 / 1e9, sum(res2), t2, size / t2 / 1e9, t1 / t2), flush=True)
