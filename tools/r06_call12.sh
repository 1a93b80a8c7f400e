# round 6: contig bins brought home during the scan (TIDDIT_EARLY_BINS=1/0) and the file blocks placed by four threads (TIDDIT_WRITE_THREADS=4/1), each interleaved
# with its own off-switch on the 240-Mb and the 3-Gb job; then the pipeline and e2e GPU tests
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TIDDIT_BENCH_TMP=/dev/shm
for sw in TIDDIT_EARLY_BINS=1,0 TIDDIT_WRITE_THREADS=4,1; do
  python tools/time_sv_modes.py 240 $sw 4 2>&1 | grep -v "amdgpu.ids\|reader thread" >> gpurun_out/r06_tail_240mb.txt
done
grep "rep [0-9]" gpurun_out/r06_tail_240mb.txt | cut -c1-170
for sw in TIDDIT_EARLY_BINS=1,0 TIDDIT_WRITE_THREADS=4,1; do
  timeout 1200 python tools/time_sv_modes.py 3000 $sw 3 2>&1 | grep -v "amdgpu.ids\|reader thread" >> gpurun_out/r06_tail_3000mb.txt
done
grep "rep [0-9]" gpurun_out/r06_tail_3000mb.txt | cut -c1-170
rm -rf /dev/shm/tiddit_bench_sv_3000
timeout 600 python -m pytest tests/test_gpu_sv_e2e.py tests/test_gpu_pipeline.py -x -q 2>&1 | tail -2
