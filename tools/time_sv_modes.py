"""`tiddit --sv --skip_assembly` on the bench's WGS-shaped file (python tools/time_sv_modes.py [Mb]): wall and stage seconds of four
back-to-back runs per setting of TIDDIT_GC_OVERLAP (1 = GC thread beside the scan, after = behind it, 0 = in sequence)."""
import contextlib, io, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import __main__ as cli, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
d = os.path.join(os.environ.get("TIDDIT_BENCH_TMP", "/tmp"), "tiddit_bench_sv_%d" % mb)
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam, contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
out = os.path.join(d, "modes")
for mode in ("1", "1", "1", "1", "1", "1"):
    os.environ["TIDDIT_GC_OVERLAP"] = mode
    for rep in range(4):
        shutil.rmtree(out + "_tiddit", ignore_errors=True)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"])
        wall = time.perf_counter() - t0
        T = cli.STAGE_SECONDS
        print("GC_OVERLAP=%-5s rep %d wall %.3f | stats %.3f signal %.3f (scan %.3f) gc wait %.3f ploidy %.3f clustering %.3f" % (
            mode, rep, wall, T["library statistics"], T["signal extraction + coverage"], T.get("  scan (ingest, coverage, predicates, signal tables)", 0),
            T["GC bins"], T["ploidy (masked medians)"], T["clustering"]), flush=True)
        if T["library statistics"] > 0.25:
            from tiddit_amd import tiddit_stats
            print("   slow statistics:", {k: round(v, 3) for k, v in tiddit_stats.STAGE_SECONDS.items()}, flush=True)
