#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04g; mkdir -p $O
export PYTHONUNBUFFERED=1
( time timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | tail -3; echo "bench rc $?"
python - <<PY
import json
d=json.load(open('$O/bench_default.json'))
print('value',d['value'],'ms/step',d['ms_per_step'],'fragment',d['config']['fragment'],'outputs',d['config']['outputs'])
print('closed_loop',json.dumps({k:v for k,v in d['closed_loop'].items() if k in ('us_per_step','groups','launcher','us_per_step_long_run','us_per_step_one_pool_one_stream','candidates','error')}))
print('no_ramp',d['value_without_clock_ramp'])
print('roof frac',d['roofline'].get('frac'),'stale',d['roofline'].get('counters_stale'),'integrator',d['roofline'].get('integrator'))
print('alternates',{k:v['us_per_step'] for k,v in d['alternates'].items()})
print('configs',{k:(v['us_per_step_separate_launches'],v['us_per_step_chained']) for k,v in d['configs'].items() if isinstance(v,dict)})
print('next_rows',{k:v for k,v in d['next_rows'].items() if k.endswith('_us') or 'step_us' in k})
print('cpu',d['cpu_baseline']['value'],d['cpu_baseline']['cores'])
PY
