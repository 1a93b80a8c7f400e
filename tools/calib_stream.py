"""What a plain streaming read of 4.85 GB (the headline launch's algorithmic bytes) reaches on this device: tdt_calib_stream_read over workgroup
counts and the two walks, torch's sum and copy beside it.  python tools/calib_stream.py"""
import ctypes, os, sys, torch
sys.path.insert(0, "/root/repo")
from tiddit_amd import _native
ctx = _native.default_context(0)
cal = torch.empty(4848000000 // 8, dtype=torch.int64, device="cuda"); cal.random_(0, 1 << 40); torch.cuda.synchronize()
for blocked in (0, 1):
    for g in (1, 2, 3, 4, 8, 16, 32, 64):
        cb, cm = ctypes.c_double(0), ctypes.c_double(0)
        _native.check(ctx.lib.tdt_calib_stream_read(ctx.handle, cal.data_ptr(), cal.numel() * 8, 10, g, blocked, ctypes.byref(cb), ctypes.byref(cm)))
        print("%s workgroups/CU %2d: best %.4f ms mean %.4f ms -> %.0f GB/s" % ("blocked    " if blocked else "grid-stride", g, cb.value, cm.value, cal.numel() * 8 / cm.value / 1e6))
x = torch.empty_like(cal)
for _ in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); s = cal.sum(); t1.record(); torch.cuda.synchronize()
    print("torch sum: %.4f ms -> %.0f GB/s" % (t0.elapsed_time(t1), cal.numel() * 8 / t0.elapsed_time(t1) / 1e6))
for _ in range(3):
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); x.copy_(cal); t1.record(); torch.cuda.synchronize()
    print("torch copy (read + write): %.4f ms -> %.0f GB/s of traffic" % (t0.elapsed_time(t1), 2 * cal.numel() * 8 / t0.elapsed_time(t1) / 1e6))
