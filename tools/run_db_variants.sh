#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  if [ "$v" = "base" ]; then unset TIDDIT_HIP_LIB; else export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so; fi
  python bench.py --full-line --steps 10 --warmup 2 --no-gc --contigs 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$v', 'pass_ms', round(d['dbscan']['roofline']['avg_pass_ms'],4))"
done
