#!/bin/bash
# run a pytest selection repeatedly until it fails; print the failure: tools/flaky_hunt.sh <file> [n] [-k expr]
F=${1:-tests/test_gpu_ingest.py}; N=${2:-12}; shift; shift
for i in $(seq 1 $N); do
  timeout 600 python -m pytest $F -q -x -p no:cacheprovider "$@" > /tmp/fh.log 2>&1
  if ! grep -q " passed" /tmp/fh.log || grep -q "failed" /tmp/fh.log; then echo "run $i FAILED"; grep -v "amdgpu.ids" /tmp/fh.log | tail -80; exit 1; fi
  echo "run $i: $(tail -1 /tmp/fh.log)"
done
