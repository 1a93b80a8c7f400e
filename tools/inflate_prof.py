"""Where a window of bgzf_inflate_lanes spends its cycles (a -DB2_PROF build: VARIANT_SRC=tdt_inflate2 tools/build_variant.sh b2prof -Iinclude -DB2_PROF;
run with TIDDIT_HIP_LIB=variants/lib_b2prof.so): python tools/inflate_prof.py <bam>.  The marks read the shader clock (s_memtime) and the
build waits for the copies' memory operations where it charges them, so the kernel itself runs slower than the shipped one: the SHARES are the result."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
os.environ.setdefault("TIDDIT_ALLOW_VARIANT", "1")      # (a measurement build on purpose)
from tiddit_amd import _native
ctx = _native.default_context(); lib = ctx.lib
comp = np.fromfile(sys.argv[1], dtype=np.uint8)
nb, consumed, produced = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_size_t(0)
_native.check(lib.tdt_bgzf_scan(_native.ptr(comp), len(comp), 1 << 40, ctypes.byref(nb), ctypes.byref(consumed), ctypes.byref(produced)))
got = np.zeros(produced.value, dtype=np.uint8)
raw = ctypes.CDLL(os.environ["TIDDIT_HIP_LIB"])
raw.tdt_debug_b2_prof(None, 1)
rc = lib.tdt_bgzf_inflate_hbm(ctx.handle, _native.ptr(comp), consumed.value, _native.ptr(got), len(got), 0)
st = (ctypes.c_ulonglong * 16)()
raw.tdt_debug_b2_prof(st, 0)
v = list(st)
names = ["gather of the stream bits + the two LUT lookups (three dependent LDS round trips)", "per-lane lengths / distances (VALU)", "chain walk (scalar) + long codes",
         "prefix sum, positions, checks", "literal store + own-lane copies (load, wait, stores complete)", "replayed matches (load, store, complete)",
         "cursor, ring refill", "tables, block headers, between windows"]
tot = float(sum(v[:8]) + v[9] + v[11]) or 1.0
print("rc", rc, "windows", v[8], "cycles per window %.0f (shader clock)" % (tot / max(1, v[8])))
for n, c in zip(names, v[:8]):
    print("  %5.1f %%  %7.0f cycles/window  %s" % (100.0 * c / tot, c / max(1, v[8]), n))
print("  %5.1f %%  %7.0f cycles/window  the hops of the chain walk alone (the line 'chain walk' above then holds only the long-code path: %.3f long codes per window, %.0f cycles each)"
      % (100.0 * v[9] / tot, v[9] / max(1, v[8]), v[10] / max(1, v[8]), v[2] / max(1, v[10])))
print("  %5.1f %%  %7.0f cycles/window  DEFLATE block headers, code lengths and the two table builds (the line 'tables, block headers, between windows' above then holds only the rest)"
      % (100.0 * v[11] / tot, v[11] / max(1, v[8])))
