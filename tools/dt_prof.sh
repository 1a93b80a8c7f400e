#!/bin/bash
# per-phase shader-clock cycles of dbt_tile (variant built with -DDT_PROF: VARIANT_SRC=tdt_dbscan tools/build_variant.sh prof -Iinclude -DDT_PROF)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$R/variants/lib_${1:-prof}.so python - <<'PY'
import sys, ctypes, atexit
import torch  # before the library: one HIP runtime in the process
sys.argv = ["bench.py", "--full-line", "--no-gc", "--no-ingest", "--no-next", "--no-cov-sv", "--no-sv-e2e", "--no-cpu-baseline", "--steps", "30", "--warmup", "5", "--contigs", "1"]
import bench
from tiddit_amd import _native
lib = ctypes.CDLL(_native.SO_PATH)
def report():
    out = (ctypes.c_ulonglong * 16)()
    lib.tdt_debug_dt_prof(out)
    v = list(out)
    tot = sum(v) or 1
    names = ["stage", "x window masks", "lane=word runs", "info/extents", "ext regs", "rank", "y window", "lane=word subruns", "sub/cHead", "extra", "scan extra", "write"]
    for k, nme in enumerate(names):
        print("%-18s %6.2f %%  (%.3g cycles)" % (nme, 100.0 * v[k] / tot, v[k]), file=sys.stderr)
atexit.register(report)
bench.main()
PY
