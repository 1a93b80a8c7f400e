"""Pass-by-pass wall of the device reader over bench.py's ingest file (4.8 M records, 259 MB of BGZF): open, batches, close timed apart,
and the helper threads' own sums — where a pass's time goes beyond its batches.   python tools/time_ingest_passes.py [passes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401
from tiddit_amd import _native, bamio, synth_bam
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
path = "/tmp/tiddit_bench_real6_8.bam"
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path + ".tmp", [("chr1", 8_000_000), ("chr2", 8_000_000)], depth=30, threads=min(16, os.cpu_count() or 1), level=6, realistic=True)
    os.replace(path + ".tmp", path)
ctx = _native.default_context()
for p in range(n):
    t0 = time.perf_counter()
    r = bamio.DeviceBamReader(path, ctx=ctx)
    t1 = time.perf_counter()
    k, marks = 0, []
    for b in r.batches():
        k += len(b)
        marks.append(time.perf_counter())
    t2 = time.perf_counter()
    rs = dict(r.reader_seconds)
    r.close()
    t3 = time.perf_counter()
    print("pass %2d: open %.2f  batches %.2f (%s)  close %.2f  total %.2f ms | %d records | %s" % (
        p, 1e3 * (t1 - t0), 1e3 * (t2 - t1), " ".join("%.1f" % (1e3 * (m - t1)) for m in marks), 1e3 * (t3 - t2), 1e3 * (t3 - t0), k,
        {a: round(1e3 * v, 1) if isinstance(v, float) else v for a, v in rs.items()}), flush=True)
    if p == n // 2:
        time.sleep(0.5)            # (an idle gap: does the next pass pay for it?)
