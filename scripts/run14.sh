cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r14_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r14_gpu_tests.log
tail -6 gpurun_out/r14_gpu_tests.log
for GE in 32 16 1; do
T2D_FORCE_GATHER=1 timeout 600 python bench.py --steps 128 --warmup 32 --gather-every $GE --no-cpu-baseline --no-configs --no-next-rows > gpurun_out/r14_g.json 2> gpurun_out/r14_g.err; echo rc=$?
python - <<PY
import json
d=json.load(open('gpurun_out/r14_g.json')); print($GE,{k:d[k] for k in ('value','ms_per_step')}, d['gather']['gathers_in_timed_region'])
PY
done
