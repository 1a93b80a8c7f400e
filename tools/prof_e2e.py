#!/usr/bin/env python3
"""cProfile of `tiddit --sv --skip_assembly` (tiddit_amd.__main__) on the sv_e2e bench BAM: where the HOST time of the end-to-end
run goes.  usage (GPU box): python tools/prof_e2e.py [--sv-mb 240]"""
import argparse, contextlib, cProfile, io, os, pstats, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (one HIP runtime in the process)
from tiddit_amd import __main__ as cli, synth_bam

ap = argparse.ArgumentParser()
ap.add_argument("--sv-mb", type=int, default=240)
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
d = "/tmp/tiddit_bench_sv_%d" % a.sv_mb
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(a.sv_mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam, contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
out = os.path.join(d, "prof")
argv = ["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"]
for rep in range(2):
    shutil.rmtree(out + "_tiddit", ignore_errors=True)
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        if rep:
            pr.runcall(cli.main, argv)
        else:
            cli.main(argv)
    print("pass %d: %.3f s" % (rep, time.perf_counter() - t0))
print({k: round(v, 3) for k, v in cli.STAGE_SECONDS.items()})
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(a.top)
print(s.getvalue()[:12000])
for f in ("discordants_WGS.tab", "splits_WGS.tab", "clips_WGS.fa"):
    p = os.path.join(out + "_tiddit", f)
    print(f, os.path.getsize(p), sum(1 for _ in open(p)))
