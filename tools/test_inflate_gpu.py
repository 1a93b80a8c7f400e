"""Device inflate vs zlib on assorted streams (run on the GPU box)."""
import ctypes, os, sys, time, zlib, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, bamio

ctx = _native.default_context()
lib = ctx.lib
rng = np.random.default_rng(1)


def bgzf(data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    out = b""
    for o in range(0, max(1, len(data)), 0xff00):
        d = data[o:o + 0xff00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
        comp = c.compress(d) + c.flush()
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return out


def check(name, data, level, strategy=zlib.Z_DEFAULT_STRATEGY):
    comp = np.frombuffer(bgzf(data, level, strategy) + bamio._BGZF_EOF, dtype=np.uint8)
    out = np.zeros(len(data), dtype=np.uint8)
    rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), len(comp), _native.ptr(out), len(out), 0)
    if rc:
        print("FAIL", name, level, _native.last_error() if hasattr(_native, "last_error") else lib.tdt_last_error())
        return False
    ok = out.tobytes() == data
    if not ok:
        w = np.frombuffer(data, dtype=np.uint8)
        bad = np.flatnonzero(out != w)
        print("MISMATCH", name, level, "first at", bad[0], "count", len(bad), "of", len(data))
    return ok


cases = {
    "empty": b"",
    "one": b"A",
    "three": b"abc",
    "zeros": bytes(200000),
    "text": (b"the quick brown fox jumps over the lazy dog. " * 5000),
    "random": rng.integers(0, 256, 150000, dtype=np.uint8).tobytes(),
    "dna": np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, 300000)].tobytes(),
    "skew": np.minimum(255, rng.geometric(0.05, 300000)).astype(np.uint8).tobytes(),
    "runs": b"".join(bytes([int(v)]) * int(n) for v, n in zip(rng.integers(0, 256, 3000), rng.integers(1, 400, 3000))),
    "period": (b"abcdefg" * 30000) + (b"xy" * 20000) + (b"0123456789ABCDEF" * 9000),
}
allok = True
for name, data in cases.items():
    for level in (0, 1, 6, 9):
        allok &= check(name, data, level)
    allok &= check(name + "/fixed", data, 6, zlib.Z_FIXED)
    allok &= check(name + "/huff", data, 6, zlib.Z_HUFFMAN_ONLY)
    allok &= check(name + "/rle", data, 6, zlib.Z_RLE)
print("small cases", "OK" if allok else "FAILED")
path = "/tmp/bulk_4_2.bam"
if not os.path.exists(path):
    from tiddit_amd import synth_bam
    synth_bam.write_bulk_bam(path, [("chr1", 4_000_000), ("chr2", 4_000_000)], depth=30, threads=16)
comp = np.fromfile(path, dtype=np.uint8)
nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
_native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
want = np.empty(produced.value, dtype=np.uint8)
_native.check(lib.tdt_bgzf_inflate(_native.ptr(comp), consumed.value, _native.ptr(want), len(want), 16))
got = np.zeros(produced.value, dtype=np.uint8)
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), consumed.value, _native.ptr(got), len(got), 0)
    dt = time.perf_counter() - t0
    print("bam: rc", rc, "%.3f s incl. copies (%.1f MB out, %d blocks)" % (dt, len(got) / 1e6, nb.value), "match" if np.array_equal(got, want) else "MISMATCH")
if rc:
    print(lib.tdt_last_error())
# corrupt one byte in the middle -> must fail
bad = comp.copy(); bad[len(bad) // 2] ^= 0x10
rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(bad), consumed.value, _native.ptr(got), len(got), 0)
print("corrupted input rc", rc, lib.tdt_last_error())
