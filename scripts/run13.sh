cd $GRAFT_REPO_ROOT
for CW in 500 2000 6000; do
for REP in 1 2; do
python bench.py --steps 20 --warmup 5 --clock-warm $CW --no-cpu-baseline --no-configs --no-next-rows --no-profile > gpurun_out/r13.json 2>/dev/null
python - <<PY
import json
d=json.load(open('gpurun_out/r13.json'))
print($CW, 'value %.4g ms/step %.5f step_us %.2f frac %s' % (d['value'], d['ms_per_step'], d['roofline']['step_us'], d['roofline'].get('frac')), {k:round(v['us_per_step'],2) for k,v in d['alternates'].items()})
PY
done; done
