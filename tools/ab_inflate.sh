#!/bin/bash
# same-box A/B of inflate kernel builds on BAM-shaped data (zlib level 6 and level 1): tools/ab_inflate.sh [<variant> ...]
# (variants/lib_<name>.so vs the in-tree library); prints the rocprofv3 average of bgzf_inflate_lanes per build, two rounds, interleaved.
cd /tmp && export TMPDIR=/tmp
one() {  # $1 = label, $2 = lib ("" = in-tree), $3.. = time_inflate_gpu.py args
  local label=$1 lib=$2; shift 2
  if [ -n "$lib" ]; then export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$lib; else unset TIDDIT_HIP_LIB; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abi -o t -- python /root/repo/tools/time_inflate_gpu.py "$@" --check > /tmp/abi.log 2>&1
  echo "$label [$*]: $(grep -c '^match' /tmp/abi.log) match, lanes avg ns $(grep bgzf_inflate_lanes /tmp/abi/t_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f3)"
}
for rep in 1 2; do
  for args in "16 6 --realistic" "16 1 --realistic"; do
    one in-tree "" $args
    for v in "$@"; do one $v /root/repo/variants/lib_$v.so $args; done
  done
done
