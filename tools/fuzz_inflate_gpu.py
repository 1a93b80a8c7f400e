"""Damage fuzz of the device inflate: random bit flips / byte smashes inside BGZF payloads must end in an error (or, when the
flip happens to be harmless, in the right bytes) — never in a hang or a crash.  Run under `timeout` on the GPU box."""
import os, struct, sys, time, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, bamio
ctx = _native.default_context(); lib = ctx.lib
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300


def bgzf(data, level):
    out, spans = b"", []
    for o in range(0, len(data), 0xff00):
        d = data[o:o + 0xff00]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        comp = c.compress(d) + c.flush()
        spans.append((len(out) + 18, len(comp)))
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", len(comp) + 25) + comp +
                struct.pack("<II", zlib.crc32(d) & 0xffffffff, len(d)))
    return out + bamio._BGZF_EOF, spans


srcs = [np.minimum(255, rng.geometric(0.04, 400000)).astype(np.uint8).tobytes(),
        (b"the quick brown fox jumps over the lazy dog. " * 9000),
        np.array(list(b"ACGT"), np.uint8)[rng.integers(0, 4, 300000)].tobytes()]
bad = ok_same = 0
t0 = time.time()
for it in range(N):
    data = srcs[it % 3]
    comp, spans = bgzf(data, (1, 6, 9)[it % 3])
    buf = np.frombuffer(comp, dtype=np.uint8).copy()
    kind = it % 4
    for _ in range(1 + it % 5):
        o, n = spans[rng.integers(0, len(spans))]
        p = o + int(rng.integers(0, n))
        if kind == 0:
            buf[p] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            buf[p:p + 8] = rng.integers(0, 256, len(buf[p:p + 8]), dtype=np.uint8)
        elif kind == 2:
            buf[p:o + n] = 0
        else:
            buf[p:p + 64] = 0xff
    out = np.zeros(len(data), dtype=np.uint8)
    rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(buf), len(buf), _native.ptr(out), len(out), 0)
    if rc == 0:
        assert out.tobytes() == data, "silent corruption at iteration %d" % it
        ok_same += 1
    else:
        bad += 1
# and the context still inflates a clean stream
comp, _ = bgzf(srcs[1], 6)
out = np.zeros(len(srcs[1]), dtype=np.uint8)
cb = np.frombuffer(comp, dtype=np.uint8)
_native.check(lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(cb), len(cb), _native.ptr(out), len(out), 0))
assert out.tobytes() == srcs[1]
print("fuzz ok: %d damaged streams rejected, %d harmless, %.1f s, kernel=%s" % (bad, ok_same, time.time() - t0, "seq" if os.environ.get("TIDDIT_INFLATE_SEQ") else "lanes"))
