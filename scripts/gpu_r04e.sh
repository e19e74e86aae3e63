#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04e; mkdir -p $O
export PYTHONUNBUFFERED=1 T2D_COUNT_STEPS=100
bash scripts/sq_variants.sh libt2d_hip.so libt2d_p1.so libt2d_p4.so libt2d_p8.so libt2d_p64.so libt2d_p13.so libt2d_p77.so 2>&1 | grep -v amdgpu.ids | tee $O/variants_metric.txt
for K in hw rb ix; do T2D_COUNT_CONFIG=$K bash scripts/sq_variants.sh libt2d_hip.so libt2d_p13.so libt2d_p77.so 2>&1 | grep -v amdgpu.ids | tee -a $O/variants_kinds.txt; done
