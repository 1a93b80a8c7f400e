#!/bin/bash
# same-box A/B of coverage kernel builds: tools/ab_cov.sh <variant> [<variant> ...]  (variants/lib_<name>.so vs the in-tree library)
run() { python bench.py --full-line --no-dbscan --no-gc --no-ingest --no-next --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_ms'])"; }
for rep in 1 2; do
  line="in-tree: $(run)"
  for v in "$@"; do line="$line   $v: $(TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so run)"; done
  echo "$line"
done
