cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_nopf.so libt2d_hip.so libt2d_nopf.so > gpurun_out/r06_ab_prefetch.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_prefetch.txt
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest3.log 2>&1; tail -5 gpurun_out/r06_gputest3.log
T2D_LIB_NAME=libt2d_hip_timing.so timeout 300 python scripts/chain_timing.py 20 gpurun_out/r06b_chain_timing_frag20.json > gpurun_out/r06b_chain_timing_frag20.log 2>&1; tail -c 300 gpurun_out/r06b_chain_timing_frag20.log
