#!/bin/bash
# All BASELINE.json configs on one GPU, both step modes (the default bench line is the metric workload, chained).
for c in metric cfg2 cfg3 cfg4 cfg5; do for m in ${MODES:-chain step}; do
  timeout 300 python bench.py --config $c --mode $m --steps 512 --warmup 64 --no-cpu-baseline --no-configs --no-next-rows --no-alternates 2>&1 | grep '^{' | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$c', d['config']['mode'], d['config']['envs_per_gpu'], 'x', d['config']['participants_per_env'], '| %.3e part-steps/s | %.1f us/step |' % (d['value'], 1e3*d['ms_per_step']), {k: round(v['avg_us_per_step'],1) for k,v in (d['roofline']['kernels'] or {}).items()}, '| frac', d['roofline'].get('frac'))"
done; done
