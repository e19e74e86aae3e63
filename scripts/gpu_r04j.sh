#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc $?"; tail -6 $O/pytest_all.log
T2D_COUNT_STEPS=100 bash scripts/sq_variants.sh libt2d_hip.so 2>&1 | grep -v amdgpu.ids | tee $O/valu.txt
for K in hw rb ix cfg3 cfg4 cfg5; do T2D_COUNT_STEPS=100 T2D_COUNT_CONFIG=$K bash scripts/sq_variants.sh libt2d_hip.so 2>&1 | grep -v amdgpu.ids | tee -a $O/valu.txt; done
timeout 600 python scripts/ab_step.py libt2d_hip.so 2>&1 | grep AB_RESULT | tee $O/ab.txt
