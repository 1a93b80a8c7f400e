#!/bin/bash
# rocprofv3 kernel-trace stats of the device ingest (file -> packed arrays) + PMC passes of bgzf_inflate.
# usage (GPU box): tools/prof_ingest.sh <tag>
TAG=${1:-ing}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/time_ingest.py > $OUT/run.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/tools/time_ingest.py > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/tools/time_inflate_gpu.py 16 1 > $OUT/pmc$i.log 2>&1
done
tail -3 $OUT/run.log
head -9 $OUT/trace/t_kernel_stats.csv | cut -c1-160
