#!/usr/bin/env python3
"""A/B of the coverage launch layouts on BASELINE configs[1] (600 M reads): binned / packed / four arrays, 500-bp and 50-bp bins,
interleaved rounds, HIP events on the launch stream.  usage (GPU box): python tools/ab_cov_binned.py [rounds]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tiddit_amd import _native, synth, tiddit_coverage
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
dev = torch.device("cuda:0")
ctx = _native.default_context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
C, L = 24, 125_000_000
with torch.cuda.stream(stream):
    reads = [synth.gen_reads_device(L, 30, dev, seed=synth.SEED + c) for c in range(C)]
torch.cuda.synchronize()
n = [int(r[0].numel()) for r in reads]
res = {}
for z, q in ((500, 20), (50, 5)):
    h = tiddit_coverage.CoverageHistogram([("s%02d" % c, L) for c in range(C)], z, ctx=ctx)
    pk = [torch.empty(n[c], dtype=torch.int64, device=dev) for c in range(C)]
    bn = [torch.empty(n[c], dtype=torch.int64, device=dev) for c in range(C)]
    torch.cuda.synchronize()
    for c in range(C):
        _native.check(ctx.lib.tdt_cov_pack_device(ctx.handle, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c], pk[c].data_ptr()))
        h.pack_binned_device(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c], bn[c].data_ptr())
    ctx.sync()
    calls = {
        "binned": lambda: h.push_binned_device_multi([(c, bn[c].data_ptr(), reads[c][0].data_ptr(), reads[c][1].data_ptr(), n[c]) for c in range(C)], q),
        "packed": lambda: h.push_packed_device_multi([(c, pk[c].data_ptr(), reads[c][1].data_ptr(), n[c]) for c in range(C)], q),
        "four": lambda: h.push_device_multi([(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c]) for c in range(C)], q),
    }
    outs = {}
    out = torch.empty(h.total_bins(), dtype=torch.float64, device=dev)
    times = {k: [] for k in calls}
    for r in range(rounds + 2):
        for k, f in calls.items():
            h.reset()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            f()
            b.record(stream)
            h.finish_all_device(out.data_ptr())
            ctx.sync()
            torch.cuda.synchronize()
            if r >= 2:
                times[k].append(a.elapsed_time(b))
            if r == 0:
                outs[k] = (out.clone(), h.kept())
    for k in calls:
        assert torch.equal(outs[k][0], outs["four"][0]) and outs[k][1] == outs["four"][1], (z, k)
        t = sorted(times[k])
        print("z=%d %-7s mean %.3f  median %.3f  min %.3f ms" % (z, k, sum(t) / len(t), t[len(t) // 2], t[0]), flush=True)
    h.close()
    del pk, bn, out
