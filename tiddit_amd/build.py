"""Build libtiddit_hip.so (hipcc, gfx950 only) in-tree: python -m tiddit_amd.build [--force]"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libtiddit_hip.so")
SOURCES = ["tdt_ctx.hip", "tdt_coverage.hip", "tdt_gc.hip", "tdt_dbscan.hip", "tdt_dbscan_yseg.hip", "tdt_sort.hip", "tdt_bam.hip", "tdt_format.hip", "tdt_bgzf.hip", "tdt_inflate.hip", "tdt_inflate2.hip", "tdt_ingest.hip", "tdt_signal.hip", "tdt_median.hip", "tdt_region.hip", "tdt_comm.hip", "tdt_means.hip", "tdt_stats.hip", "tdt_sigtab.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [os.path.join(os.path.dirname(HERE), "include", "tiddit_hip.h")]
    relink = force
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(CSRC, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _newer(s, o) or any(_newer(d, o) for d in deps):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((src, subprocess.Popen(cmd)))
            relink = True
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed on " + src)
    if relink or not os.path.exists(SO):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", SO, "-lz", "-lpthread", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    build_pycand(force=force, verbose=verbose)
    return SO


def pycand_path():
    import sysconfig
    return os.path.join(HERE, "_pycand" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pycand(force=False, verbose=False):
    """the CPython extension that builds the candidate dictionaries of tiddit_cluster.main (csrc/tdt_pycand.c; gcc, host only).  Optional:
    without the interpreter's headers tiddit_cluster keeps its Python loop."""
    import sysconfig
    src, out = os.path.join(CSRC, "tdt_pycand.c"), pycand_path()
    inc = sysconfig.get_paths().get("include") or ""
    if not os.path.exists(os.path.join(inc, "Python.h")):
        return None
    if force or _newer(src, out):
        cmd = [os.environ.get("CC", "gcc"), "-O2", "-shared", "-fPIC", "-Wall", "-I" + inc, src, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
