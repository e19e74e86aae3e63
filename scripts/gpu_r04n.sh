#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04n; mkdir -p $O
for Q in 4 8; do for M in 0 1 2; do for S in 2 3 4 6; do
  GPU_MAX_HW_QUEUES=$Q timeout 20 scripts/_build/stream_overlap2 $M $S >> $O/overlap2.jsonl 2>> $O/overlap2.err || echo "{\"q\": $Q, \"mode\": $M, \"streams\": $S, \"failed\": true}" >> $O/overlap2.jsonl
done; done; done
cat $O/overlap2.jsonl
