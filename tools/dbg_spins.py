import ctypes, os, sys, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["TIDDIT_HIP_LIB"] = os.path.join(sys.path[0], "variants/lib_dbg.so")
import torch
from tiddit_amd import _native, synth
ctx = _native.default_context(0)
L = ctx.lib
L.tdt_debug_words.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
pts = synth.gen_points(5_000_000)
dev = torch.device("cuda:0")
x = torch.from_numpy(pts[:, 0].astype(np.uint32).view(np.int32)).to(dev); y = torch.from_numpy(pts[:, 1].astype(np.uint32).view(np.int32)).to(dev)
lab = torch.empty(len(pts), dtype=torch.float64, device=dev); lid = torch.empty(1, dtype=torch.int64, device=dev)
off = np.array([0, len(pts)], dtype=np.int64)
torch.cuda.synchronize()
w = np.zeros(16, dtype=np.uint32)
for it in range(3):
    L.tdt_debug_words(ctx.handle, w.ctypes.data_as(ctypes.c_void_p), 1)
    _native.check(L.tdt_dbscan_device(ctx.handle, x.data_ptr(), y.data_ptr(), len(pts), off.ctypes.data_as(ctypes.c_void_p), 1, ctypes.c_uint64(500), 3, 0, lab.data_ptr(), lid.data_ptr()))
    L.tdt_debug_words(ctx.handle, w.ctypes.data_as(ctypes.c_void_p), 0)
    print("iter", it, "err", w[0], "sum_spins", w[4], "max_spins", w[5], "sum_hops", w[6], "lookbacks", w[7])
L.tdt_debug_ts.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
ts = np.zeros((2, 8192, 4), dtype=np.uint64)
L.tdt_debug_ts(ts.ctypes.data_as(ctypes.c_void_p), ts.nbytes)
t = ts[0, :1221].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0   # us (100 MHz)
print("tile: start, agg_ready, prefix_done, end (us)")
for k in list(range(0, 1221, 61)) + [1220]:
    print(k, np.round(t[k], 2))
print("max end", t[:, 3].max(), "mean agg_ready-start", (t[:, 1] - t[:, 0]).mean(), "mean lookback", (t[:, 2] - t[:, 1]).mean(), "mean finish", (t[:, 3] - t[:, 2]).mean())
print("start spread", t[:, 0].max(), "sorted starts sample", np.round(np.sort(t[:, 0])[::122], 2))
