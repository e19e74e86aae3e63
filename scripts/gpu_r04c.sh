#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c; mkdir -p $O
export PYTHONUNBUFFERED=1
rocminfo 2>/dev/null | grep -i -E "Compute Unit|Max Queue|Queue Max|Name:|Partition|Workgroup Max|Max Waves|Queue" | head -40 > $O/rocminfo.txt
rocm-smi --showcomputepartition --showmemorypartition 2>&1 | head -20 >> $O/rocminfo.txt
for Q in 2 4 8; do GPU_MAX_HW_QUEUES=$Q timeout 200 scripts/_build/stream_overlap2 > $O/overlap2_q$Q.json 2> $O/overlap2_q$Q.err; echo "overlap2 q$Q rc $?"; done
timeout 200 scripts/_build/stream_overlap2 > $O/overlap2_default.json 2>&1
timeout 200 scripts/_build/valu_roof > $O/valu_roof.json 2> $O/valu_roof.err; echo "valu_roof rc $?"
timeout 300 python -m pytest tests/test_gpu_chain_oracle.py -m gpu -x -q -k "broken or changes or ring" > $O/pytest_fault.log 2>&1; echo "fault tests rc $?"; tail -15 $O/pytest_fault.log
timeout 300 python -m pytest tests/test_gpu_closed_loop.py tests/test_drift.py -m gpu -x -q > $O/pytest_loop.log 2>&1; echo "loop tests rc $?"; tail -5 $O/pytest_loop.log
GPU_MAX_HW_QUEUES=4 timeout 420 python scripts/closed_loop_sweep.py 400 $O/sweep_q4.json > $O/sweep_q4.log 2>&1; echo "sweep q4 rc $?"; tail -16 $O/sweep_q4.log
