#!/bin/bash
# A/B of the GC kernel in one box: variants/lib_gc_prev.so (build_variant.sh on the stashed tree) vs the in-tree library
run() { python bench.py --full-line --no-dbscan --no-ingest --no-cpu-baseline --steps 40 --warmup 3 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['gc']['roofline']['avg_launch_ms'])"; }
for rep in 1 2 3; do echo "prev: $(TIDDIT_ALLOW_VARIANT=1 TIDDIT_HIP_LIB=$PWD/variants/lib_gc_prev.so run)   new: $(run)"; done
python -m pytest tests/test_gpu_parity.py -q -x -k gc 2>&1 | tail -1
