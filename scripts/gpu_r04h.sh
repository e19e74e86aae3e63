#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04h; mkdir -p $O
export PYTHONUNBUFFERED=1 T2D_COUNT_STEPS=100
bash scripts/sq_variants.sh libt2d_hip.so libt2d_p128.so libt2d_p384.so 2>&1 | grep -v amdgpu.ids | tee $O/variants.txt
