cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_ingest.py tests/test_gpu_pipeline.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python -m pytest tests/test_gpu_sv_e2e.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
tools/ab_trees.sh 3000 4 variants/r05_tree > gpurun_out/r06_ab_trees_3000mb.txt 2>&1; cat gpurun_out/r06_ab_trees_3000mb.txt | cut -c1-260
for i in 1 2 3; do python bench.py --steps 3 --warmup 1 --no-cov-sv --no-dbscan --no-gc --no-next --no-cpu-baseline --contigs 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ingest', d['roofline'].get('ingest_records_per_sec'), 'sv_e2e', d['roofline'].get('sv_e2e_wall_s'))"; done
