#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04d; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 150 scripts/_build/valu_roof > $O/valu_roof.json 2> $O/valu_roof.err; echo "valu_roof rc $?"
GPU_MAX_HW_QUEUES=8 timeout 300 python scripts/closed_loop_sweep.py 400 $O/sweep_q8.json > $O/sweep_q8.log 2>&1; echo "sweep q8 rc $?"; tail -16 $O/sweep_q8.log
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc $?"; tail -8 $O/pytest_all.log
