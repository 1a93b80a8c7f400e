cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
python tools/time_contig_table.py 3 2>&1 | grep -v amdgpu.ids | cut -c1-420
