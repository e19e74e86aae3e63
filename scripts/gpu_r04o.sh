#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04o; mkdir -p $O
export PYTHONUNBUFFERED=1
for Q in 4 8; do for RAW in raw torch; do for G in 1 2 3 4; do
  GPU_MAX_HW_QUEUES=$Q timeout 120 python scripts/closed_loop_fresh.py $G threads $RAW 2>/dev/null | grep us_per_step >> $O/fresh.jsonl
done; done; done
GPU_MAX_HW_QUEUES=4 timeout 120 python scripts/closed_loop_fresh.py 4 thread raw 2>/dev/null | grep us_per_step >> $O/fresh.jsonl
GPU_MAX_HW_QUEUES=4 timeout 120 python scripts/closed_loop_fresh.py 4 graph raw 2>/dev/null | grep us_per_step >> $O/fresh.jsonl
cat $O/fresh.jsonl
