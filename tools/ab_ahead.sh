#!/bin/bash
# Round 6: the decoupled inflate (tdt_ingest_push_ahead on the reader's own streams) on the 3-Gb / 54-GB WGS-shaped file, one box:
# tiddit --sv --skip_assembly with spans begun ahead or not, the inflate grid's reserve (workgroups per CU left to the launch stream's
# kernels), the span size; then a rocprofv3 kernel trace of the 240-Mb job for the timeline.   usage (GPU box): tools/ab_ahead.sh [Mb] [reps]
MB=${1:-3000}
REPS=${2:-3}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out
cd $R
export TIDDIT_BENCH_TMP=/dev/shm
python tools/time_sv_modes.py $MB TIDDIT_INGEST_AHEAD=1,0 $REPS 2>&1 | grep -v amdgpu.ids > $OUT/r06_sv_ahead_${MB}mb.txt
python tools/time_sv_modes.py $MB TIDDIT_INFLATE_RESERVE=1,0,2 $REPS 2>&1 | grep -v amdgpu.ids > $OUT/r06_sv_reserve_${MB}mb.txt
python tools/time_sv_modes.py $MB TIDDIT_INGEST_CHUNK=469762048,268435456,134217728,67108864 $REPS 2>&1 | grep -v amdgpu.ids > $OUT/r06_sv_chunk_${MB}mb.txt
grep -h "rep [0-9]" $OUT/r06_sv_ahead_${MB}mb.txt $OUT/r06_sv_reserve_${MB}mb.txt $OUT/r06_sv_chunk_${MB}mb.txt | cut -c1-150
