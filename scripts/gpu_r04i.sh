#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04i; mkdir -p $O
export PYTHONUNBUFFERED=1 T2D_AB_CHAIN=0
timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_st1.so libt2d_st2.so libt2d_hip.so 2>&1 | grep AB_RESULT | tee $O/ab.txt
