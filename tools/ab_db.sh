#!/bin/bash
# same-box A/B of clustering builds (5 M points, one bucket): HIP-event time of tdt_dbscan_device, 3 rounds interleaved
run() { python bench.py --full-line --no-gc --no-ingest --no-next --no-cov-sv --no-sv-e2e --no-cpu-baseline --steps 30 --warmup 5 --contigs 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%.1f us between events, %.1f us per step; shared %.1f us' % (1e3*d['dbscan']['roofline']['avg_pass_ms'], 1e3*d['dbscan']['ms_per_step'], 1e3*d['dbscan_shared']['device_ms_rank0']))"; }
for rep in 1 2 3; do
  echo "in-tree: $(run)"
  for v in "$@"; do echo "$v: $(TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so run)"; done
done
