#!/bin/bash
# All BASELINE.json configs on one GPU (the default bench line is the metric workload).
for c in metric cfg2 cfg3 cfg4 cfg5; do for g in ${GROUPS_LIST:-0}; do
  timeout 300 python bench.py --config $c --steps 300 --warmup 30 --groups $g --no-cpu-baseline --no-configs 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read())
print('$c', 'G', d['config']['env_groups'], d['config']['envs_per_gpu'], 'x', d['config']['participants_per_env'], '| %.3e part-steps/s | %.1f us/step |' % (d['value'], 1e3*d['ms_per_step']), {k: round(v['avg_us'],1) for k,v in d['roofline']['kernels'].items()}, '| frac %.4f' % d['roofline']['frac'])"
done; done
