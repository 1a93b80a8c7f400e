cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -6
tools/trace_db.sh r06 2>&1 | grep -E "rs_|sc_|sd_|dbt_|Name" | cut -c1-160
python - <<'P'
import json
d=json.loads([l for l in open("gpurun_out/prof_r06/trace.log") if l.startswith("{")][-1])
sd=d["dbscan"]["sort_dbscan"]; print("cluster_columns", json.dumps(sd.get("cluster_columns"))[:600])
print("dbscan", d["dbscan"]["value"], d["dbscan"]["ms_per_step"])
P
export TIDDIT_BENCH_TMP=/dev/shm
python tools/time_sv_modes.py 240 TIDDIT_INGEST_RAMP=0,32,64,128 4 2>&1 | grep -v "amdgpu.ids\|reader thread" > gpurun_out/r06_sv_ramp_240mb.txt; cat gpurun_out/r06_sv_ramp_240mb.txt | cut -c1-150
for r in 0 32 64 0 32 64; do TIDDIT_INGEST_RAMP=$r python bench.py --steps 3 --warmup 1 --no-cov-sv --no-dbscan --no-gc --no-next --no-cpu-baseline --no-sv-e2e --contigs 1 --full-line 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); i=d['ingest']; print('ramp $r ingest', round(i['value']/1e6,1), 'M rec/s', round(i['ms_per_step'],2), 'ms')"; done
