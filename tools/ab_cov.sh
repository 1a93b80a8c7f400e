#!/bin/bash
# same-box A/B of two coverage kernel builds: variants/lib_$1.so vs the in-tree library
run() { python bench.py --no-dbscan --no-gc --no-ingest --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do echo "$1: $(TIDDIT_HIP_LIB=$PWD/variants/lib_$1.so run)   in-tree: $(run)"; done
