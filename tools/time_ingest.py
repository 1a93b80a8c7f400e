"""Stage timing of the device ingest (run on the GPU box)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, bamio, synth_bam
path = sys.argv[1] if len(sys.argv) > 1 else "/tmp/bulk_10_2.bam"
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", 10_000_000), ("chr2", 10_000_000)], depth=30, threads=16)
ctx = _native.default_context(); lib = ctx.lib
for rep in range(2):
    t0 = time.perf_counter()
    r = bamio.DeviceBamReader(path)
    t1 = time.perf_counter()
    n = 0
    sp = r._spans()
    while True:
        item = sp.next()
        if item is None:
            break
        n += item[1]
    t2 = time.perf_counter()
    r.close()
    print("open %.3f s; read spans only %.3f s (%.0f MB)" % (t1 - t0, t2 - t1, n / 1e6))
for rep in range(2):
    r = bamio.DeviceBamReader(path)
    tp = 0.0; k = 0; first = True
    t0 = time.perf_counter()
    sp = r._spans()
    while True:
        item = sp.next()
        if item is None:
            break
        buf, consumed, _abs = item
        ta = time.perf_counter()
        nn = ctypes.c_size_t(0)
        _native.check(lib.tdt_ingest_push(r._h, _native.ptr(buf), consumed, r._skip if first else 0, ctypes.byref(nn)))
        first = False
        tp += time.perf_counter() - ta
        k += nn.value
    t1 = time.perf_counter()
    r.close()
    print("spans + push: %.3f s, of which push calls %.3f s (%d records)" % (t1 - t0, tp, k))
