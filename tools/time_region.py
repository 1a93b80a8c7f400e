"""Time tdt_region_counts_device (the get_region loop) on a 125 Mb / 30x contig with 200k candidates; parity-check a sample
against the literal oracle loop.  Run on the GPU box:  python tools/time_region.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tiddit_amd import _native, synth

dev = torch.device("cuda", 0)
ctx = _native.default_context(0)
L, NQ, MAX_INS, MIN_Q = 125_000_000, 200_000, 600, 5
start, end, mapq, flag = synth.gen_reads_device(L, 30.0, dev)
n = start.numel()
g = torch.Generator(device=dev); g.manual_seed(7)
ins = (torch.randn(n, generator=g, device=dev) * 50 + 350).to(torch.int32)
far = torch.rand(n, generator=g, device=dev) < 0.02
mate_pos = torch.where(far, torch.randint(0, L, (n,), generator=g, device=dev, dtype=torch.int32), torch.clamp(start + ins, 0, L - 1))
tlen = (mate_pos - start + 150).to(torch.int32)
mate_tid = torch.where(torch.rand(n, generator=g, device=dev) < 0.01, 3, 0).to(torch.int32)
has_sa = (torch.rand(n, generator=g, device=dev) < 0.01).to(torch.uint8)
qs = torch.randint(0, L - 5000, (NQ,), generator=g, device=dev, dtype=torch.int32)
qe = qs + torch.randint(0, 2000, (NQ,), generator=g, device=dev, dtype=torch.int32)
qb = torch.where(torch.rand(NQ, generator=g, device=dev) < 0.5, qs, qe)
out = torch.zeros(NQ, 7, dtype=torch.int64, device=dev)
max_span = int((end - start).max().item())
stream = torch.cuda.Stream(device=dev)
ctx.set_stream(stream.cuda_stream)


def launch():
    _native.check(ctx.lib.tdt_region_counts_device(ctx.handle, start.data_ptr(), end.data_ptr(), mapq.data_ptr(), flag.data_ptr(),
                                                   mate_tid.data_ptr(), mate_pos.data_ptr(), tlen.data_ptr(), has_sa.data_ptr(), n, 0,
                                                   max_span, L, qs.data_ptr(), qe.data_ptr(), qb.data_ptr(), NQ, MIN_Q, MAX_INS,
                                                   out.data_ptr()))


with torch.cuda.stream(stream):
    for _ in range(3):
        launch()
    stream.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(stream)
    for _ in range(20):
        launch()
    b.record(stream)
    stream.synchronize()
ms = a.elapsed_time(b) / 20
o = out.cpu().numpy()
reads_seen = float(o[:, 1].sum())
print("n_reads %d  queries %d  kernel %.3f ms  -> %.1f M candidates/s, ~%.2f G read-visits/s (n_reads sum %.3g)" %
      (n, NQ, ms, NQ / ms / 1e3, reads_seen / ms / 1e6, reads_seen))
if "--check" in sys.argv:
    import oracle
    tab = dict(start=start.cpu().numpy(), end=end.cpu().numpy(), mapq=mapq.cpu().numpy(), flag=flag.cpu().numpy().view(np.uint16),
               mate_tid=mate_tid.cpu().numpy(), mate_pos=mate_pos.cpu().numpy(), tlen=tlen.cpu().numpy(), has_sa=has_sa.cpu().numpy())
    h_qs, h_qe, h_qb = qs.cpu().numpy(), qe.cpu().numpy(), qb.cpu().numpy()
    t0 = time.perf_counter()
    for q in range(0, NQ, NQ // 24):
        want = oracle.get_region_counts(tab, 0, L, int(h_qs[q]), int(h_qe[q]), int(h_qb[q]), MIN_Q, MAX_INS)
        assert np.array_equal(o[q], want), (q, o[q], want)
    print("24 sampled candidates match the literal loop (%.2f s/candidate on the CPU, full-contig scan)" % ((time.perf_counter() - t0) / 24))
