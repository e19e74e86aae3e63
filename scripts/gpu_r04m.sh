#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04m; mkdir -p $O
export PYTHONUNBUFFERED=1
(cd _r03 && timeout 600 python scripts/ab_step.py libt2d_hip.so 2>&1 | grep AB_RESULT | sed 's/^/r03 /') | tee -a $O/ab.txt
timeout 900 python scripts/ab_step.py $LIBS 2>&1 | grep AB_RESULT | tee -a $O/ab.txt
