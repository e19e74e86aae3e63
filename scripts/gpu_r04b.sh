#!/bin/bash
# round 4, second GPU call: why do env groups serialise?  stream concurrency probe, kernel traces of the closed loop, the
# round-2 groups sweep (no policy kernel), per-SIMD VALU issue rates, the fault tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04b; mkdir -p $O
export PYTHONUNBUFFERED=1
for Q in 4 8 16; do GPU_MAX_HW_QUEUES=$Q timeout 120 scripts/_build/stream_overlap > $O/stream_overlap_q$Q.json 2> $O/stream_overlap_q$Q.err; echo "overlap q$Q rc $?"; done
timeout 120 scripts/_build/stream_overlap > $O/stream_overlap_default.json 2>&1
timeout 120 scripts/_build/valu_roof > $O/valu_roof.json 2> $O/valu_roof.err; echo "valu_roof rc $?"
timeout 300 python -m pytest tests/test_gpu_chain_oracle.py -m gpu -x -q -k "broken or changes or ring" > $O/pytest_fault.log 2>&1; echo "fault tests rc $?"; tail -15 $O/pytest_fault.log
for CFG in "4 thread" "4 thread raw" "2 thread" "4 graph"; do
  T=$(echo $CFG | tr ' ' '_')
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/tr_$T -o run -- python scripts/closed_loop_trace.py $CFG > $O/tr_$T.log 2>&1
  python scripts/trace_overlap.py $O/tr_$T/run_kernel_trace.csv > $O/tr_$T.json 2>> $O/tr_$T.log; echo "trace $CFG: $(cat $O/tr_$T.json | cut -c1-600)"
done
timeout 400 python scripts/env_groups_sweep.py > $O/env_groups_sweep.log 2>&1; head -8 $O/env_groups_sweep.log
