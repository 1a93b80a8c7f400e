#!/usr/bin/env python3
"""Where the first (cold) pass of the device ingest over a file differs from the following ones: per-batch stage timers of three
passes over the sv_e2e bench BAM of <Mb> (generated under TIDDIT_BENCH_TMP if missing).  usage: tools/time_first_pass.py [Mb]"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime in the process)
from tiddit_amd import bamio, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 1200
os.environ["TIDDIT_INGEST_TIMING"] = "1"
base = os.environ.get("TIDDIT_BENCH_TMP") or tempfile.gettempdir()
d = os.path.join(base, "tiddit_bench_sv_%d" % mb)
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam + ".tmp", contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
    os.replace(bam + ".tmp", bam)
for rep in range(3):
    t0 = time.perf_counter()
    r = bamio.open_bam(bam)
    t1 = time.perf_counter()
    n = 0
    marks = []
    for b in r.batches():
        n += len(b)
        marks.append(time.perf_counter())
    t2 = time.perf_counter()
    tm = list(r.timings)
    r.close()
    gaps = [marks[0] - t1] + [marks[i] - marks[i - 1] for i in range(1, len(marks))]
    keys = sorted({k for row in tm for k in row if isinstance(row[k], (int, float))})
    sums = {k: round(sum(row.get(k, 0) for row in tm), 1) for k in keys if k.endswith("_ms")}
    print("pass %d: open %.3f s, %d batches in %.3f s (%d records); slowest batches %s ms" % (rep, t1 - t0, len(marks), t2 - t1, n, [round(1e3 * g) for g in sorted(gaps)[-4:]]))
    print("   first 6 batch walls ms:", [round(1e3 * g, 1) for g in gaps[:6]])
    print("   sums ms:", sums, flush=True)
