#!/bin/bash
# rocprofv3 evidence for the hot-path kernels of bench.py (run on the GPU box via gpurun): kernel-trace stats, then PMC passes
# (each in its own run; never combined with sys/hip/hsa tracing).   usage: tools/profile_all.sh <tag>
TAG=${1:-r02}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 20 --warmup 3 --no-cpu-baseline --no-next --no-ingest --no-sv-e2e $@"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --full-line $ARGS > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_FLAT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/bench.py --full-line $ARGS > $OUT/pmc$i.log 2>&1
done
python $R/tools/collect_profiles.py $TAG $ARGS
