"""tiddit_gc.main end to end on a synthetic FASTA: device-side line-end handling vs stripping on the host (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from tiddit_amd import _native, synth, tiddit_gc
from tiddit_amd.fasta import FastaFile
mb = int(sys.argv[1]) if len(sys.argv) > 1 else 400
path = "/tmp/gc_%d.fa" % mb
contigs = [("chr%d" % i, mb * 1_000_000 // 4) for i in range(1, 5)]
if not os.path.exists(path):
    with open(path, "wb") as f:
        for name, ln in contigs:
            s = synth.gen_sequence(ln, seed=len(name) + ln % 89)
            f.write((">%s\n" % name).encode())
            body = np.full((ln // 60, 61), 10, dtype=np.uint8)
            body[:, :60] = s[:ln // 60 * 60].reshape(-1, 60)
            f.write(body.tobytes())
            if ln % 60:
                f.write(s[ln // 60 * 60:].tobytes() + b"\n")
    if os.path.exists(path + ".fai"):
        os.remove(path + ".fai")
_native.default_context()
t0 = time.perf_counter(); FastaFile(path); print("index (build .fai if missing): %.2f s" % (time.perf_counter() - t0))
names = [c[0] for c in contigs]
for rep in range(2):
    t0 = time.perf_counter(); a = tiddit_gc.main(path, names, 1, 50, 0.5); t1 = time.perf_counter() - t0
    print("gc main, file bytes straight to the device: %.3f s (%.2f G bases/s end to end)" % (t1, mb * 1e6 / t1 / 1e9))
fa = FastaFile(path)
t0 = time.perf_counter()
b = {n: tiddit_gc.binned_gc_array(fa.fetch_array(n), 50, 0.5) for n in names}
t2 = time.perf_counter() - t0
print("gc main, line ends stripped on the host first: %.3f s" % t2)
print("identical:", all(np.array_equal(a[n], b[n]) for n in names))
