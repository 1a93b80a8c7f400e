#!/bin/bash
# same-box A/B of coverage builds: packed and four-array 500-bp launches and the 50-bp launch: tools/ab_cov4.sh [<variant> ...]
run() { python bench.py --full-line --no-dbscan --no-gc --no-ingest --no-next --no-cpu-baseline --no-sv-e2e --steps 20 --warmup 3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
r,v,f=d['roofline'],d['coverage_sv']['roofline'],d['four_array_layout']
print('packed %.3f | four arrays %.3f | z50 %.3f ms' % (r['avg_launch_ms'], f['avg_launch_ms'], v['avg_launch_ms']))"; }
for rep in 1 2 3; do
  echo "in-tree: $(run)"
  for v in "$@"; do echo "$v: $(TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so run)"; done
done
