#!/bin/bash
# Round 6, first GPU call: what bounds step_kernel_chained?  Phase stamps of the chained form, the wait-side SQ counters,
# power + clock sampled through the run.  -> gpurun_out/r06_*.json (copied to profiles/ by hand)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest.log 2>&1; echo "gpu tests rc $?" | tee -a gpurun_out/r06_gputest.log; tail -3 gpurun_out/r06_gputest.log
timeout 300 python scripts/power_clock.py gpurun_out/r06_power_clock.json > gpurun_out/r06_power_clock.log 2>&1; tail -c 1500 gpurun_out/r06_power_clock.log
T2D_LIB_NAME=libt2d_hip_timing.so timeout 300 python scripts/chain_timing.py 20 gpurun_out/r06_chain_timing_frag20.json > gpurun_out/r06_chain_timing_frag20.log 2>&1; tail -c 600 gpurun_out/r06_chain_timing_frag20.log
T2D_LIB_NAME=libt2d_hip_timing.so timeout 300 python scripts/chain_timing.py 32 gpurun_out/r06_chain_timing_frag32.json > gpurun_out/r06_chain_timing_frag32.log 2>&1
MODE=chain bash scripts/sq_wait_chain.sh r06 > gpurun_out/r06_sq_wait_chain.log 2>&1; tail -c 1500 gpurun_out/r06_sq_wait_chain.log
MODE=step bash scripts/sq_wait_chain.sh r06 > gpurun_out/r06_sq_wait_step.log 2>&1
timeout 300 python scripts/time_integrate.py fast > gpurun_out/r06_time_integrate.log 2>&1; tail -20 gpurun_out/r06_time_integrate.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r06_bench_driver_like_0.json 2> gpurun_out/r06_bench_driver_like_0.err; tail -c 400 gpurun_out/r06_bench_driver_like_0.json
amd-smi metric -g 0 > gpurun_out/r06_amdsmi_metric.txt 2>&1; amd-smi static -g 0 > gpurun_out/r06_amdsmi_static.txt 2>&1
