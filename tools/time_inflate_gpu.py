"""Device inflate on a bulk BAM (run on the GPU box, under rocprofv3 for kernel times): python tools/time_inflate_gpu.py [Mb] [level]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
level = int(sys.argv[2]) if len(sys.argv) > 2 else 1
realistic = "--realistic" in sys.argv
path = "/tmp/bulk_inf_%d_%d%s.bam" % (mb, level, "_real" if realistic else "")
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", mb * 1_000_000), ("chr2", mb * 1_000_000)], depth=30, threads=16, level=level, realistic=realistic)
ctx = _native.default_context(); lib = ctx.lib
comp = np.fromfile(path, dtype=np.uint8)
nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
_native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
import torch
d_out = torch.empty(produced.value + 256, dtype=torch.uint8, device="cuda")
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), consumed.value, d_out.data_ptr(), produced.value, 1)
    dt = time.perf_counter() - t0
    print("rc %d  %.3f s incl. H2D of %.0f MB  (%.0f MB out, %d blocks) -> %.1f GB/s out" % (rc, dt, consumed.value / 1e6, produced.value / 1e6, nb.value, produced.value / dt / 1e9))
if "--check" in sys.argv:
    want = np.empty(produced.value, dtype=np.uint8)
    _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(want), len(want), 16))
    print("match" if np.array_equal(d_out[:produced.value].cpu().numpy(), want) else "MISMATCH")
