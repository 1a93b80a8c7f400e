cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_sv_e2e.py -x -q -k "grch38" 2>&1 | tail -12
python -m pytest tests/test_gpu_parity.py -x -q -k "sort or cluster or dbscan" 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err; tail -c 300 gpurun_out/r06_bench_b.err
python - <<'P'
import json
d=json.loads(open("gpurun_out/r06_bench_b.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"])
print({k:v for k,v in d["roofline"].items() if not isinstance(v,(dict,str))})
print("cluster_columns", d["dbscan"].get("cluster_columns")); print("ingest", d.get("ingest")); print("sv", {k:v for k,v in d.get("sv_e2e",{}).items() if k in ("wall_s","serial_s")})
P
tools/trace_db.sh r06b 2>&1 | grep -E "rs_|sc_|sd_" | cut -c1-40,150-260
