#!/bin/bash
# per-kernel rocprof stats of the clustering pass for each variant lib
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp; export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = "base" ]; then unset TIDDIT_HIP_LIB; else export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$R/variants/lib_$v.so; fi
  rm -rf $R/gpurun_out/pv_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pv_$v -o t -- python $R/bench.py --full-line --steps 5 --warmup 1 --no-gc --contigs 1 --no-cpu-baseline >/dev/null 2>&1
  echo "== $v"; grep -E "^.(db|tile_scan)" $R/gpurun_out/pv_$v/t_kernel_stats.csv | awk -F'",' '{split($1,a,"("); n=split($2,b,","); print a[1], b[3]}' 
done
