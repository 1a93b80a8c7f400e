#!/bin/bash
# rocprofv3 evidence for the device ingest as bench.py's `ingest` section runs it (4.8 M-record BAM, file -> records in HBM):
# kernel-trace stats, then four PMC passes (FETCH_SIZE and WRITE_SIZE each in its own run: together rocprofv3 aborted; every pass under a timeout) over the ingest kernels (one counter group per run, kernel-trace only).
# usage (GPU box): tools/prof_ingest_bench.sh <tag>   -> gpurun_out/prof_<tag>/{trace,pmc*}, summaries printed
TAG=${1:-ing}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--no-gc --no-next --no-cov-sv --no-dbscan --no-sv-e2e --no-cpu-baseline --contigs 1"
python $R/bench.py --full-line $ARGS > $OUT/run.json 2> $OUT/run.err          # writes the BAM, warms the page cache
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python $R/bench.py --full-line $ARGS > $OUT/trace.log 2>&1
i=0
for PMC in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d $OUT/pmc$i -o p -- python $R/bench.py --full-line $ARGS > $OUT/pmc$i.log 2>&1
done
grep -E "Name|bgzf_|bam_|sig_" $OUT/trace/t_kernel_stats.csv > $OUT/ingest_kernel_stats.csv
( echo "# rocprofv3 --pmc passes over bench.py's ingest section (one counter group per run, kernel-trace only), mean per launch";
  python $R/tools/pmc_summary.py $OUT bgzf_; python $R/tools/pmc_summary.py $OUT bam_ ) > $OUT/pmc_ingest.txt
cut -c1-170 $OUT/ingest_kernel_stats.csv; head -30 $OUT/pmc_ingest.txt
