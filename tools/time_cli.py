"""End-to-end `tiddit --cov` ingest timing on a bulk synthetic BAM (BGZF inflate -> record decode -> device histogram).
Run on the GPU box:  python tools/time_cli.py [Mb_per_contig] [contigs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import synth_bam, tiddit_coverage, _native
from tiddit_amd import bamio

mb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nc = int(sys.argv[2]) if len(sys.argv) > 2 else 2
contigs = [("chr%d" % (i + 1), mb * 1_000_000) for i in range(nc)]
path = "/tmp/bulk_%d_%d.bam" % (mb, nc)
t0 = time.perf_counter()
n = synth_bam.write_bulk_bam(path, contigs, depth=30, threads=os.cpu_count() or 8)
print("wrote %s: %d records, %.1f MB compressed, %.1f s (host cores %d)" % (path, n, os.path.getsize(path) / 1e6, time.perf_counter() - t0,
                                                                        os.cpu_count()))
_native.default_context()
# stage 1: inflate only
t0 = time.perf_counter()
ub = 0
with open(path, "rb", buffering=0) as f:
    for piece in bamio.inflate_pieces(f):
        ub += len(piece) - (1 << 20)
t_inf = time.perf_counter() - t0
print("inflate only: %.2f s  (%.0f MB/s uncompressed, %.1f M records/s)" % (t_inf, ub / t_inf / 1e6, n / t_inf / 1e6))
# stage 2: inflate + decode
t0 = time.perf_counter()
r = bamio.BamReader(path)
k = 0
for b in r.batches():
    k += len(b)
r.close()
t_dec = time.perf_counter() - t0
assert k == n, (k, n)
print("inflate + decode: %.2f s (%.1f M records/s)" % (t_dec, n / t_dec / 1e6))
# stage 3: the whole --cov command, host ingest vs device ingest
from tiddit_amd import __main__ as cli
for mode in ("host", "device", "device"):
    os.environ["TIDDIT_HOST_INGEST"] = "1" if mode == "host" else "0"
    t0 = time.perf_counter()
    args = cli._cov_parser().parse_args(["--cov", "--bam", path, "-o", "/tmp/bulk_cov_" + mode, "-z", "500"])
    cli.run_cov(args)
    t_all = time.perf_counter() - t0
    print("tiddit --cov end to end, %s ingest: %.2f s (%.1f M records/s, %.0f MB/s of BAM)" % (mode, t_all, n / t_all / 1e6, os.path.getsize(path) / t_all / 1e6))
print("identical .bed:", open("/tmp/bulk_cov_host.bed").read() == open("/tmp/bulk_cov_device.bed").read())
t0 = time.perf_counter()
r = bamio.DeviceBamReader(path)
k = 0
for b in r.batches():
    k += len(b)
r.close()
print("device ingest only: %.2f s (%.1f M records/s), host chases %d" % (time.perf_counter() - t0, k / (time.perf_counter() - t0) / 1e6, r.host_chases))
