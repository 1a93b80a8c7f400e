#!/bin/bash
# run bench (coverage only) for each variant lib; prints avg launch ms (parity checked on 1 contig)
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "base" ]; then unset TIDDIT_HIP_LIB; else export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  python bench.py --full-line --steps 5 --warmup 2 --no-dbscan --no-gc --cpu-contigs 1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'launch_ms', round(d['roofline']['avg_launch_ms'],4), 'GB/s', round(d['roofline']['achieved'],1), 'step_ms', round(d['ms_per_step'],3), 'parity', d.get('parity_checked'))"
done
