#!/bin/bash
# kernel-trace stats of the clustering kernels for several builds: tools/trace_db_variants.sh <variant> ...
for v in in-tree "$@"; do
  if [ "$v" = in-tree ]; then unset TIDDIT_HIP_LIB; else export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  echo "== $v"; tools/trace_db.sh v_$v > /dev/null; python - <<PY
import csv
for r in csv.DictReader(open("gpurun_out/prof_v_$v/trace/t_kernel_stats.csv")):
    if any(k in r["Name"] for k in ("dbt_", "dbm_", "dby_", "db_signal")):
        print("   %-28s calls %4s  avg %8.1f us" % (r["Name"].split("(")[0][:28], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
