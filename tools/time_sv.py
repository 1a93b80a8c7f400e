"""Stage timing of `tiddit --sv --skip_assembly` on a synthetic SV BAM (run on the GPU box): python tools/time_sv.py [Mb]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import synth, synth_bam, fasta
mb = float(sys.argv[1]) if len(sys.argv) > 1 else 2
d = "/tmp/sv_t"; os.makedirs(d, exist_ok=True)
contigs = [("chr1", int(mb * 1e6)), ("chr2", int(mb * 0.6e6)), ("chrM", 16000)]
bam, fa = d + "/s.bam", d + "/s.fa"
t0 = time.perf_counter()
info = synth_bam.write_synthetic_bam(bam, contigs, depth=30, seed=9, n_events=40)
with open(fa, "w") as f:
    for n, l in contigs:
        f.write(">%s\n" % n)
        s = synth.gen_sequence(l, seed=len(n) + l % 97).tobytes().decode()
        for o in range(0, l, 60):
            f.write(s[o:o + 60] + "\n")
print("wrote %d records in %.1f s" % (info["n_records"], time.perf_counter() - t0))
from tiddit_amd import __main__ as cli
import shutil
for mode in ("1", "0", "0"):
    os.environ["TIDDIT_HOST_INGEST"] = mode
    shutil.rmtree(d + "/out_tiddit", ignore_errors=True)
    sys.argv = ["tiddit", "--sv", "--bam", bam, "--ref", fa, "-o", d + "/out", "--skip_assembly", "--threads", "1"]
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    try:
        cli.main()
    except SystemExit:
        pass
    pr.disable()
    print("=== %s ingest: tiddit --sv %.2f s" % ("host" if mode == "1" else "device", time.perf_counter() - t0))
    pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
