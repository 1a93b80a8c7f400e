# round 6: the candidate dictionaries built with the CPython API (tiddit_amd/_pycand) against the Python loop (TIDDIT_PY_CANDIDATES=1), interleaved,
# on the 240-Mb and the 3-Gb job; then the e2e GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TIDDIT_BENCH_TMP=/dev/shm
python tools/time_sv_modes.py 240 TIDDIT_PY_CANDIDATES=0,1 5 2>&1 | grep -v "amdgpu.ids\|reader thread" > gpurun_out/r06_pycand_240mb.txt; grep "rep [0-9]" gpurun_out/r06_pycand_240mb.txt | cut -c1-170
timeout 1500 python tools/time_sv_modes.py 3000 TIDDIT_PY_CANDIDATES=0,1 4 2>&1 | grep -v "amdgpu.ids\|reader thread" > gpurun_out/r06_pycand_3000mb.txt; grep "rep [0-9]" gpurun_out/r06_pycand_3000mb.txt | cut -c1-170
rm -rf /dev/shm/tiddit_bench_sv_3000
timeout 600 python -m pytest tests/test_gpu_sv_e2e.py -x -q 2>&1 | tail -2
