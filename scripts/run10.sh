cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r10_bench20.json 2> gpurun_out/r10_bench20.err; echo "rc=$?"; tail -3 gpurun_out/r10_bench20.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r10_bench20.json'))
def short(o):
    if isinstance(o,dict): return {k:short(v) for k,v in o.items() if k not in ('how','note','sample','counters_source','untimed_prewarm','workload','peak_is','idm_note','lidar_note','vec_parking_env_note','python_loop_note')}
    return o
print(json.dumps(short(d),indent=1)[:6000])
PY
