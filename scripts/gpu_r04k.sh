#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04k; mkdir -p $O
export PYTHONUNBUFFERED=1
for i in 1 2; do
(cd _r03 && timeout 600 python scripts/ab_step.py libt2d_hip.so 2>&1 | grep AB_RESULT | sed 's/^/r03 /') | tee -a $O/ab.txt
timeout 600 python scripts/ab_step.py libt2d_hip.so 2>&1 | grep AB_RESULT | sed 's/^/r04 /' | tee -a $O/ab.txt
done
