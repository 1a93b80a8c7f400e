"""Wall time per C-ABI entry point over one warm `tiddit --sv --skip_assembly` run on the bench's file (main thread and helper threads summed):
which fixed costs a 0.2-s job pays.  python tools/time_c_calls.py [Mb]"""
import contextlib, io, os, shutil, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tiddit_amd import __main__ as cli, synth_bam, _native
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 240
d = os.path.join(os.environ.get("TIDDIT_BENCH_TMP", "/tmp"), "tiddit_bench_sv_%d" % mb)
bam, fa = os.path.join(d, "WGS.bam"), os.path.join(d, "ref.fa")
if not (os.path.exists(bam) and os.path.exists(fa)):
    os.makedirs(d, exist_ok=True)
    contigs = synth_bam.wgs_contigs(mb)
    seqs = synth_bam.write_fasta(fa, contigs)
    synth_bam.write_wgs_sv_bam(bam, contigs, threads=min(32, os.cpu_count() or 1), ref_seqs=seqs)
out = os.path.join(d, "ccalls")
argv = ["--sv", "--bam", bam, "--ref", fa, "-o", out, "--skip_assembly", "--force_overwrite"]
lib = _native.load()
acc, lock = {}, threading.Lock()
main_id = threading.get_ident()


class Timed:
    def __init__(self, name, fn):
        self.name, self.fn = name, fn

    def __call__(self, *a):
        t0 = time.perf_counter()
        r = self.fn(*a)
        dt = time.perf_counter() - t0
        key = (self.name, threading.get_ident() == main_id)
        with lock:
            c = acc.setdefault(key, [0, 0.0])
            c[0] += 1
            c[1] += dt
        return r


class Proxy:
    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)
        object.__setattr__(self, "_cache", {})

    def __getattr__(self, name):
        c = self._cache
        if name not in c:
            f = getattr(self._lib, name)
            c[name] = Timed(name, f) if name.startswith("tdt_") else f
        return c[name]


proxy = Proxy(lib)
_native._lib = proxy
ctx = _native.default_context()
ctx.lib = proxy
for rep in range(3):
    shutil.rmtree(out + "_tiddit", ignore_errors=True)
    acc.clear()
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        cli.main(argv)
    wall = time.perf_counter() - t0
print("wall %.4f s; C calls on the main thread %.4f s" % (wall, sum(v[1] for k, v in acc.items() if k[1])))
for (name, on_main), (n, t) in sorted(acc.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%-40s %-6s calls %5d  %8.3f ms" % (name, "main" if on_main else "helper", n, 1e3 * t))
