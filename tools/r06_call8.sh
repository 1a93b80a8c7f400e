cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py tests/test_gpu_sv_e2e.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -m pytest tests/test_gpu_parity.py -x -q -k "gc" 2>&1 | grep -E "passed|failed|rror" | tail -3
python tools/time_contig_table.py 3 2>&1 | grep -v amdgpu.ids | cut -c1-420
for i in 1 2 3; do python bench.py --steps 3 --warmup 1 --no-cov-sv --no-dbscan --no-gc --no-next --no-cpu-baseline --contigs 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ingest', d['roofline'].get('ingest_records_per_sec'), 'sv_e2e', d['roofline'].get('sv_e2e_wall_s'), len(json.dumps(d)))"; done
