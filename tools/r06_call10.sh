# round 6, final tree: GPU suite, rocprofv3 kernel stats + PMC passes (profiles/r06_kernel_stats.csv, r06_pmc.txt, traffic.json), the driver's bench line,
# the kernel timeline of the 240-Mb job, and the 3-Gb job four times (TIDDIT_INGEST_AHEAD=1,0 interleaved)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
timeout 900 tools/profile_all.sh r06 2>&1 | tail -5
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_n1.json 2> gpurun_out/r06_bench_n1.err; cp gpurun_out/bench_detail_n1.json gpurun_out/r06_bench_n1_detail.json
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_n1.json").read().strip().splitlines()[-1])
print({k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,str))})
print("cluster_columns", d["dbscan"].get("cluster_columns")); print("ingest", d.get("ingest")); print("sv", {k:v for k,v in d.get("sv_e2e",{}).items() if k in ("wall_s","serial_s")}, d.get("run_s"))
P
timeout 600 tools/trace_sv.sh r06 240 2>&1 | tail -12 | cut -c1-200
export TIDDIT_BENCH_TMP=/dev/shm
timeout 1500 python tools/time_sv_modes.py 3000 TIDDIT_INGEST_AHEAD=1,0 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sv_final_3000mb.txt; grep "rep [0-9]" gpurun_out/r06_sv_final_3000mb.txt | cut -c1-150
