"""`tiddit --sv --skip_assembly` on the bench's WGS-shaped file, the settings of one environment switch interleaved on the same file and box:
python tools/time_sv_modes.py [Mb] [SWITCH=v1,v2,...] [reps]      (default: 240 TIDDIT_GC_OVERLAP=1,after,0 4)
e.g. TIDDIT_BENCH_TMP=/dev/shm python tools/time_sv_modes.py 3000 TIDDIT_SCAN_PIPELINE=1,0 3 — wall and stage seconds of every run."""
import contextlib, io, os, shutil, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import __main__ as cli, synth_bam
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
switch, values = (sys.argv[2].split("=", 1) if len(sys.argv) > 2 else ("TIDDIT_GC_OVERLAP", "1,after,0"))
values = values.split(",")
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = os.path.join(os.environ.get("TIDDIT_BENCH_TMP", "/tmp"), "tiddit_bench_sv_%d" % mb)
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam, contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
out = os.path.join(d, "modes")
for rep in range(-1, reps):                       # rep -1: the file's first read, not printed as a result
    for v in values:
        os.environ[switch] = v
        shutil.rmtree(out + "_tiddit", ignore_errors=True)
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"])
        wall = time.perf_counter() - t0
        T = cli.STAGE_SECONDS
        print("%s%s=%-5s rep %d wall %.3f | stats %.3f signal %.3f (scan %.3f, ingest %.3f) gc wait %.3f ploidy %.3f clustering %.3f" % (
            "(first pass) " if rep < 0 else "", switch, v, rep, wall, T["library statistics"], T["signal extraction + coverage"],
            T.get("  scan (ingest, coverage, predicates, signal tables)", 0), T.get("    ingest (inflate + decode, device)", 0),
            T["GC bins"], T["ploidy (masked medians)"], T["clustering"]), flush=True)
        from tiddit_amd import tiddit_signal
        if tiddit_signal.READER_SECONDS:
            print("      reader thread:", {k: (round(v, 3) if isinstance(v, float) else v) for k, v in tiddit_signal.READER_SECONDS.items()}, flush=True)
        if rep < 0:
            break
