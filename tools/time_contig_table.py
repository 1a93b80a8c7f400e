"""What a GRCh38-shaped contig table costs: `tiddit --sv --skip_assembly` on the two ~24-Mb fixtures' files — 27 contigs
(tests/golden/sv_e2e.json) against 3 366 (tests/golden/sv_e2e_grch38.json) — in one process, warm repetitions, stage seconds of each.
python tools/time_contig_table.py [reps]"""
import contextlib, io, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from sv_e2e_common import load_fixture, materialise
from tiddit_amd import __main__ as cli
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
files = {}
td = tempfile.mkdtemp(dir=os.environ.get("TIDDIT_BENCH_TMP", "/tmp"))
for name in ("sv_e2e.json", "sv_e2e_grch38.json"):
    fx = load_fixture(os.path.join(R, "tests", "golden"), name)
    d = os.path.join(td, name[:-5])
    os.makedirs(d)
    bam, fa, contigs = materialise(fx, d, threads=min(16, os.cpu_count() or 1))
    files[name] = (fx, bam, fa, contigs, d)
    print("%s: %d contigs, %d records, BAM %.0f MB" % (name, len(contigs), fx["n_records"], os.path.getsize(bam) / 1e6), flush=True)
for rep in range(-1, reps):
    for name, (fx, bam, fa, contigs, d) in files.items():
        out = os.path.join(d, "run")
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            cli.main(["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite", "-s", str(fx["params"]["n_reads_stats"])])
        wall = time.perf_counter() - t0
        if rep >= 0:
            T = cli.STAGE_SECONDS
            print("%-20s rep %d wall %.3f | %s" % (name, rep, wall, ", ".join("%s %.3f" % (k.strip()[:34], v) for k, v in T.items() if isinstance(v, float) and v >= 0.002)), flush=True)
