#!/usr/bin/env python3
"""How the size of a span (the BGZF bytes one inflate launch decodes) moves the device ingest: a launch's persistent waves take the span's
blocks off one counter, 8192 at a time, and the launch ends with its slowest wave — a 448-MB span is ~2.7 "rounds" of blocks, so its last round
runs part empty.   python tools/time_ingest_chunks.py [Mb per contig = 120] [level = 1] [chunk_MB ...]
Per chunk size: wall of a pass over the file (records in HBM, nothing consumed) and the pushes' own stage sums (TIDDIT_INGEST_TIMING)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["TIDDIT_INGEST_TIMING"] = "1"
from tiddit_amd import _native, bamio, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 120
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
chunks = [int(x) for x in sys.argv[3:]] or [224, 448, 672, 896]
path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "chunks_%d_%d.bam" % (mb, level))
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", mb * 1_000_000), ("chr2", mb * 1_000_000)], depth=30, threads=32, level=level, realistic=True)
print("file: %.0f MB" % (os.path.getsize(path) / 1e6), flush=True)
ctx = _native.default_context()
for rep in range(3):
    for c in chunks:
        r = bamio.DeviceBamReader(path, ctx=ctx, chunk=c << 20)
        t0 = time.perf_counter()
        n = sum(len(b) for b in r.batches())
        ctx.sync()
        wall = time.perf_counter() - t0
        tm = r.timings
        r.close()
        if rep:
            print("chunk %4d MB: %d records in %.3f s = %.1f M records/s, %.2f GB/s of BGZF | %d pushes: inflate+crc %.1f ms, find %.1f, decode %.1f, h2d %.1f, push walls %.1f" % (
                c, n, wall, n / wall / 1e6, os.path.getsize(path) / wall / 1e9, len(tm), sum(t["inflate_crc_ms"] for t in tm), sum(t["find_records_ms"] for t in tm),
                sum(t["decode_ms"] for t in tm), sum(t["h2d_ms"] for t in tm), sum(t["push_wall_ms"] for t in tm)), flush=True)
