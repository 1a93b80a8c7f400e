"""What the 64-bit windows of bgzf_inflate_lanes are made of (a -DB2_STATS build: VARIANT_SRC=tdt_inflate2 tools/build_variant.sh b2stats -Iinclude -DB2_STATS;
run with TIDDIT_HIP_LIB=variants/lib_b2stats.so): python tools/inflate_stats2.py <bam>"""
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
raw.tdt_debug_b2_stats(None, 1)
rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), consumed.value, _native.ptr(got), len(got), 0)
st = (ctypes.c_ulonglong * 16)()
raw.tdt_debug_b2_stats(st, 0)
w, sym, lit, par, lng, dep, stops, mb, rb = list(st)[:9]
print("rc", rc, "blocks", nb.value, "in", consumed.value, "out", produced.value)
print("windows %d: %.1f input bits, %.1f output bytes, %.2f symbols each (%.2f literals, %.2f matches)" % (w, 8.0 * consumed.value / w, produced.value / w, sym / w, lit / w, (par + lng + dep) / w))
print("matches: %.1f bytes on average; own-lane copies %.1f%%, replayed because longer than 16 bytes %.1f%%, replayed because the source lies in the window's own output %.1f%%"
      % (mb / max(1, par + lng + dep), 100.0 * par / max(1, par + lng + dep), 100.0 * lng / max(1, par + lng + dep), 100.0 * dep / max(1, par + lng + dep)))
print("replayed matches per window %.3f (%.1f bytes each); stops at a code longer than the LUT: one per %.1f windows" % ((lng + dep) / w, rb / max(1, lng + dep), w / max(1, stops)))
