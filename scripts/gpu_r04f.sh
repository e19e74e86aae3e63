#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04f; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc $?"; tail -8 $O/pytest_all.log
T2D_COUNT_STEPS=100 bash scripts/sq_variants.sh libt2d_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/valu.txt
for K in hw rb ix; do T2D_COUNT_STEPS=100 T2D_COUNT_CONFIG=$K bash scripts/sq_variants.sh libt2d_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $O/valu.txt; done
timeout 300 python bench.py --steps 1024 --warmup 128 --no-cpu-baseline --no-configs --no-next-rows > $O/bench_1024.json 2> $O/bench_1024.err; python -c "
import json; d=json.load(open('$O/bench_1024.json')); print('chain us/step', d['ms_per_step']*1e3, 'alt', {k:v['us_per_step'] for k,v in (d.get('alternates') or {}).items()})"
