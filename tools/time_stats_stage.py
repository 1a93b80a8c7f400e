"""Where the `library statistics` stage of `tiddit --sv` spends its time: tiddit_stats.STAGE_SECONDS of back-to-back runs on the bench's file.
python tools/time_stats_stage.py [Mb] [reps]"""
import contextlib, io, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import __main__ as cli, synth_bam, tiddit_stats
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
d = os.path.join(os.environ.get("TIDDIT_BENCH_TMP", "/tmp"), "tiddit_bench_sv_%d" % mb)
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam, contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
out = os.path.join(d, "stats_stage")
for rep in range(-1, reps):
    shutil.rmtree(out + "_tiddit", ignore_errors=True)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"])
    wall = time.perf_counter() - t0
    print("rep %d wall %.3f stats %.4f |" % (rep, wall, cli.STAGE_SECONDS["library statistics"]),
          {k: round(v, 4) for k, v in tiddit_stats.STAGE_SECONDS.items()}, flush=True)
