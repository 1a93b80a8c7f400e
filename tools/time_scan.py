"""tiddit_signal.scan_signals on a bulk BAM (run on the GPU box): python tools/time_scan.py [Mb per contig]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import synth_bam, tiddit_signal, _native
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 20
path = "/tmp/bulk_%d_2.bam" % mb
if not os.path.exists(path):
    synth_bam.write_bulk_bam(path, [("chr1", mb * 1_000_000), ("chr2", mb * 1_000_000)], depth=30, threads=16)
_native.default_context()
for rep in range(2):
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    header, chroms, cov, data, splits, clips = tiddit_signal.scan_signals(path, 5, 600, 10000, 30, 20)
    pr.disable()
    dt = time.perf_counter() - t0
    n = mb * 2 * 1_000_000 * 30 // 100
    print("scan_signals: %.2f s (%.1f M records/s)" % (dt, n / dt / 1e6))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
