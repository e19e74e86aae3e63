#!/bin/bash
# round 4, first GPU call: new tests, VALU issue microbenchmark, closed-loop sweep, full GPU suite, class counters
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04a; mkdir -p $O
export PYTHONUNBUFFERED=1
timeout 120 scripts/_build/valu_roof > $O/valu_roof.json 2> $O/valu_roof.err; echo "valu_roof rc $?"
timeout 600 python -m pytest tests/test_gpu_chain_oracle.py tests/test_gpu_closed_loop.py tests/test_drift.py -m gpu -x -q > $O/pytest_new.log 2>&1; echo "new tests rc $?"; tail -15 $O/pytest_new.log
GPU_MAX_HW_QUEUES=8 timeout 420 python scripts/closed_loop_sweep.py 400 $O/sweep_q8.json > $O/sweep_q8.log 2>&1; echo "sweep q8 rc $?"; tail -20 $O/sweep_q8.log
GPU_MAX_HW_QUEUES=16 timeout 420 python scripts/closed_loop_sweep.py 400 $O/sweep_q16.json > $O/sweep_q16.log 2>&1; echo "sweep q16 rc $?"; tail -16 $O/sweep_q16.log
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest_all.log 2>&1; echo "all gpu tests rc $?"; tail -5 $O/pytest_all.log
MODE=chain timeout 600 bash scripts/sq_classes.sh r04a_chain > $O/sq_classes_chain.log 2>&1; echo "sq classes rc $?"; tail -60 $O/sq_classes_chain.log | head -80
