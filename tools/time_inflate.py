"""Where the BGZF ingest time goes (run on the GPU box after tools/time_cli.py wrote /tmp/bulk_10_2.bam)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, synth_bam
path = "/tmp/bulk_10_2.bam"
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", 10_000_000), ("chr2", 10_000_000)], depth=30, threads=os.cpu_count())
lib = _native.load()
t0 = time.perf_counter(); comp = np.fromfile(path, dtype=np.uint8); print("read file %.3f s" % (time.perf_counter() - t0))
nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
t0 = time.perf_counter()
_native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
print("scan %.4f s  blocks %d  out %.1f MB" % (time.perf_counter() - t0, nb.value, produced.value / 1e6))
out = np.empty(produced.value, dtype=np.uint8)
for th in (16, 32, 64, 128, 256):
    for rep in range(2):
        t0 = time.perf_counter()
        _native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(out), len(out), th))
        dt = time.perf_counter() - t0
        print("inflate threads %3d %s: %.3f s  %.1f GB/s" % (th, "cold" if (th == 16 and rep == 0) else "warm", dt, len(out) / dt / 1e9))
cap = len(out) // 36 + 1
names = {"tid": np.int32, "pos": np.int32, "end": np.int32, "mapq": np.uint8, "flag": np.uint16, "mate_tid": np.int32, "mate_pos": np.int32,
         "tlen": np.int32, "l_seq": np.int32, "cigar_first": np.uint32, "cigar_last": np.uint32, "rec_off": np.uint64, "sa_off": np.int64}
# skip header: find first record via BamReader header length
from tiddit_amd import bamio
r = bamio.BamReader(path); r.close()
import struct
l_text = struct.unpack_from("<i", out, 4)[0]; o = 8 + l_text; nref = struct.unpack_from("<i", out, o)[0]; o += 4
for _ in range(nref):
    ln = struct.unpack_from("<i", out, o)[0]; o += 4 + ln + 4
raw = out[o:]
arrs = {k: np.empty(cap, dtype=dt) for k, dt in names.items()}
for th in (1, 16, 64, 128):
    lib.tdt_host_threads(th)
    for rep in range(2):
        c, n = ctypes.c_size_t(0), ctypes.c_size_t(0)
        t0 = time.perf_counter()
        _native.check(lib.tdt_bam_decode(_native.ptr(raw), len(raw), cap, ctypes.byref(c), ctypes.byref(n), *[_native.ptr(arrs[k]) for k in names]))
        dt = time.perf_counter() - t0
        print("decode threads %3d: %.3f s  %.1f M records/s" % (th, dt, n.value / dt / 1e6))
