#!/bin/bash
# Same-box A/B of two whole TREES (library + Python side): `tiddit --sv --skip_assembly` on the bench's WGS-shaped file, each tree in its own
# process (one cold pass, then warm repetitions), the two trees interleaved.   usage (GPU box): tools/ab_trees.sh <Mb> <pairs> <other tree dir>
# e.g. tools/ab_trees.sh 3000 4 variants/r05_tree   (variants/r05_tree = `git archive <round-5 commit> tiddit_amd include tools/time_sv_modes.py` + its build)
MB=${1:-3000}
PAIRS=${2:-4}
OTHER=${3:-variants/r05_tree}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
export TIDDIT_BENCH_TMP=${TIDDIT_BENCH_TMP:-/dev/shm}
python tools/time_sv_modes.py $MB TIDDIT_INGEST_AHEAD=1 0 > /dev/null 2>&1          # writes the file
for p in $(seq 1 $PAIRS); do
  for T in $R $R/$OTHER; do
    echo "== tree $(basename $T) pair $p"
    python $T/tools/time_sv_modes.py $MB TIDDIT_INGEST_AHEAD=1 2 2>&1 | grep -E "rep [0-9]|reader thread" | cut -c1-330
  done
done
