#!/bin/bash
# rocprofv3 kernel trace of `tiddit --sv --skip_assembly` on the bench's WGS-shaped file (second pass of one process: buffers warm) and the
# timeline of what ran under each inflate launch (tools/kernel_timeline.py).   usage (GPU box): tools/trace_sv.sh <tag> [Mb]
TAG=${1:-r06}
MB=${2:-240}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/trace_sv_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export TIDDIT_BENCH_TMP=${TIDDIT_BENCH_TMP:-/dev/shm}
python $R/tools/time_sv_modes.py $MB TIDDIT_INGEST_AHEAD=1 1 > $OUT/warm.log 2>&1       # writes the file
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/time_sv_modes.py $MB TIDDIT_INGEST_AHEAD=1 1 > $OUT/trace.log 2>&1
K=$(find $OUT/trace -name "*kernel_trace.csv" | head -1)
python $R/tools/kernel_timeline.py $K 40 > $R/gpurun_out/${TAG}_sv${MB}_timeline.txt
S=$(find $OUT/trace -name "*kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/${TAG}_sv${MB}_kernel_stats.csv
head -30 $R/gpurun_out/${TAG}_sv${MB}_timeline.txt | cut -c1-220
