#!/bin/bash
# same-box A/B of ingest builds (device BGZF inflate + decode of a BAM-shaped file): tools/ab_ingest.sh [<variant> ...]
run() { python bench.py --no-dbscan --no-gc --no-next --no-cov-sv --no-sv-e2e --no-cpu-baseline --contigs 1 --ingest-mb ${INGEST_MB:-40} --steps 3 --warmup 1 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])['ingest']
print('%.1f M records/s, %.0f MB/s of BGZF' % (d['value']/1e6, d['bam_MB_per_sec']))"; }
for rep in 1 2; do
  echo "in-tree: $(run)"
  for v in "$@"; do echo "$v: $(TIDDIT_HIP_LIB=$PWD/variants/lib_$v.so run)"; done
done
