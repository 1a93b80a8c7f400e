#!/bin/bash
# rocprofv3 kernel-trace stats of the clustering section only: tools/trace_db.sh <tag>
TAG=${1:-db}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --full-line --steps 10 --warmup 2 --no-cpu-baseline --no-gc --no-ingest --no-next --no-cov-sv --no-sv-e2e --contigs 1 "$@" > $OUT/trace.log 2>&1
grep -E "db|sd_|rs_|scan" $OUT/trace/t_kernel_stats.csv | cut -c1-200
