#!/bin/bash
# time bgzf_inflate for experimental variants built by tools/build_variant.sh: tools/run_inflate_variants.sh v1 v2 ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=/root/repo/variants/lib_$v.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv_$v -o t -- python /root/repo/tools/time_inflate_gpu.py 16 1 --check > /tmp/pv_$v.log 2>&1
  echo "$v: match=$(grep -c match /tmp/pv_$v.log) $(grep bgzf_inflate /tmp/pv_$v/t_kernel_stats.csv | sed 's/.*)",//' | cut -d, -f1-3)"
done
