cd /tmp && export TMPDIR=/tmp
for args in "16 6 --realistic" "16 1 --realistic"; do
  python /root/repo/tools/time_inflate_gpu.py $args --check 2>&1 | tail -2
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr -o t -- python /root/repo/tools/time_inflate_gpu.py $args > /dev/null 2>&1
  grep -E "bgzf_inflate|bgzf_crc" /tmp/pr/t_kernel_stats.csv | cut -c1-130
  TIDDIT_INFLATE_SEQ=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pr2 -o t -- python /root/repo/tools/time_inflate_gpu.py $args > /dev/null 2>&1
  grep -E "bgzf_inflate" /tmp/pr2/t_kernel_stats.csv | cut -c1-130
done
