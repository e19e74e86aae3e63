#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04l; mkdir -p $O
export PYTHONUNBUFFERED=1 T2D_AB_SMALL=1
(cd _r03 && timeout 600 python scripts/ab_step.py libt2d_hip.so 2>&1 | grep AB_RESULT | sed 's/^/r03 /') | tee -a $O/ab.txt
timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_nse.so libt2d_oq.so libt2d_both.so 2>&1 | grep AB_RESULT | tee -a $O/ab.txt
