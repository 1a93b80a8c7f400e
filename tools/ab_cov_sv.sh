#!/bin/bash
# same-box A/B of coverage kernel builds at 500-bp and 50-bp bins: tools/ab_cov_sv.sh [<variant> ...]  (variants/lib_<name>.so vs the in-tree library;
# TIDDIT_COV_MODE=0 forces the run-merged flavour at 50 bp).  Prints mean/median/min launch times of 20 steps, three rounds, builds interleaved.
run() { python bench.py --full-line --no-dbscan --no-gc --no-ingest --no-next --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r,v=d['roofline'],d['coverage_sv']['roofline']
print('z500 mean %.3f med %.3f min %.3f | z50 mean %.3f med %.3f min %.3f' % (r['avg_launch_ms'], r['median_launch_ms'], r['min_launch_ms'], v['avg_launch_ms'], v['median_launch_ms'], v['min_launch_ms']))"; }
for rep in 1 2 3; do
  echo "in-tree: $(run)"
  for v in "$@"; do echo "$v: $(TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so run)"; done
done
