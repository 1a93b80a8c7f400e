#!/bin/bash
# same-box A/B of ingest builds (variants/lib_<name>.so vs the in-tree library) on bench.py's ingest section: tools/ab_ingest.sh [<variant> ...]
# prints records/s and the per-stage sums of the one-batch pass (find_records / chain_check / decode are what the segment size and the decode kernel move)
cd /tmp && export TMPDIR=/tmp
ARGS="--no-gc --no-next --no-cov-sv --no-dbscan --no-sv-e2e --no-cpu-baseline --contigs 1"
one() {
  local label=$1 lib=$2
  if [ -n "$lib" ]; then export TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$lib; else unset TIDDIT_HIP_LIB; fi
  python /root/repo/bench.py --full-line $ARGS 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['ingest']
s=d['per_batch']['span_448MB']['sum_ms']
print('$label: %.1f M records/s, %.2f ms/step | one batch: inflate+crc %.2f find %.3f chain %.3f decode %.3f push wall %.2f' % (d['value']/1e6, d['ms_per_step'], s['inflate_crc_ms'], s['find_records_ms'], s['chain_check_ms'], s['decode_ms'], s['push_wall_ms']))"
}
one warmup ""
for rep in 1 2; do
  one in-tree ""
  for v in "$@"; do one $v /root/repo/variants/lib_$v.so; done
done
