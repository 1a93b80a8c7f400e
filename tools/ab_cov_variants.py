#!/usr/bin/env python3
"""Ceilings of the binned 500-bp coverage launch: the in-tree library and measurement variants (variants/lib_<name>.so built by
tools/build_variant.sh with -DCOV_EXP_*), each in its own process, same box.  usage: python tools/ab_cov_variants.py name ..."""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, %r)
import torch
from tiddit_amd import _native, synth, tiddit_coverage
dev = torch.device("cuda:0")
ctx = _native.default_context(0)
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)
C, L = 24, 125_000_000
with torch.cuda.stream(stream):
    reads = [synth.gen_reads_device(L, 30, dev, seed=synth.SEED + c) for c in range(C)]
torch.cuda.synchronize()
n = [int(r[0].numel()) for r in reads]
for z, q in ((500, 20), (50, 5)):
    h = tiddit_coverage.CoverageHistogram([("s%%02d" %% c, L) for c in range(C)], z, ctx=ctx)
    bn = [torch.empty(n[c], dtype=torch.int64, device=dev) for c in range(C)]
    torch.cuda.synchronize()
    for c in range(C):
        h.pack_binned_device(c, reads[c][0].data_ptr(), reads[c][1].data_ptr(), reads[c][2].data_ptr(), reads[c][3].data_ptr(), n[c], bn[c].data_ptr())
    ctx.sync()
    ts = []
    for r in range(14):
        h.reset()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(stream)
        h.push_binned_device_multi([(c, bn[c].data_ptr(), reads[c][0].data_ptr(), reads[c][1].data_ptr(), n[c]) for c in range(C)], q)
        b.record(stream)
        ctx.sync(); torch.cuda.synchronize()
        if r >= 2: ts.append(a.elapsed_time(b))
    ts.sort()
    print("z=%%d mean %%.3f median %%.3f min %%.3f" %% (z, sum(ts)/len(ts), ts[len(ts)//2], ts[0]), flush=True)
    h.close(); del bn
''' % REPO
for rep in range(2):
    for name in ["in-tree"] + sys.argv[1:]:
        env = dict(os.environ)
        if name != "in-tree":
            env["TIDDIT_HIP_LIB"] = os.path.join(REPO, "variants", "lib_%s.so" % name)
            env["TIDDIT_ALLOW_VARIANT"] = "1"
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print("%-10s %s" % (name, " | ".join(l for l in out.stdout.strip().splitlines() if l.startswith("z="))), flush=True)
        if out.returncode:
            print(out.stderr[-600:])
