#!/bin/bash
# device inflate on BAM-shaped content (reads cut from a common reference, quality runs): both kernels, zlib levels 6 and 1.
# usage (GPU box): tools/prof_inflate_real.sh [out file under gpurun_out/]
OUT=${GRAFT_REPO_ROOT:-/root/repo}/gpurun_out/${1:-inflate_real.txt}
cd /tmp && export TMPDIR=/tmp
: > $OUT
for args in "16 6 --realistic" "16 1 --realistic" "16 1"; do
  echo "== tools/time_inflate_gpu.py $args  (29707 BGZF blocks, 1.94 GB out)" >> $OUT
  python /root/repo/tools/time_inflate_gpu.py $args --check 2>&1 | grep -E "rc 0|match|MISMATCH" | tail -2 >> $OUT
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o t -- python /root/repo/tools/time_inflate_gpu.py $args > /dev/null 2>&1
  grep -E "Name|bgzf_inflate|bgzf_crc" /tmp/pr/t_kernel_stats.csv >> $OUT
  TIDDIT_INFLATE_SEQ=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr2 -o t -- python /root/repo/tools/time_inflate_gpu.py $args > /dev/null 2>&1
  grep -E "bgzf_inflate" /tmp/pr2/t_kernel_stats.csv >> $OUT
done
cat $OUT
