import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TIDDIT_ALLOW_VARIANT", "1")      # (a measurement build on purpose)
from tiddit_amd import _native
ctx = _native.default_context(); lib = ctx.lib
path = sys.argv[1]
comp = np.fromfile(path, dtype=np.uint8)
nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
_native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
got = np.zeros(produced.value, dtype=np.uint8)
raw = ctypes.CDLL(os.environ["TIDDIT_HIP_LIB"])
raw.tdt_debug_bz_stats(None, 1)
rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), consumed.value, _native.ptr(got), len(got), 0)
st = (ctypes.c_ulonglong * 64)()
raw.tdt_debug_bz_stats(st, 0)
st = list(st)
print("rc", rc, "out", len(got), "literals", st[0], "matches", st[1], "match bytes", st[2], "avg len %.1f" % (st[2] / max(1, st[1])))
tot = st[1]
cum = 0
for k in range(0, 17):
    cum += st[8 + k]
    print("dist < 2^%-2d : %5.1f%%" % (k, 100.0 * cum / tot))
cum = 0
for k in range(0, 10):
    cum += st[32 + k]
    print("len  < 2^%-2d : %5.1f%%" % (k, 100.0 * cum / tot))
