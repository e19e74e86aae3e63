cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_loop4.so libt2d_loop4nr.so > gpurun_out/r06_ab_loop4.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_loop4.txt
T2D_LIB_NAME=libt2d_loop4.so timeout 600 python -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/r06_loop4_chain_tests.log 2>&1; tail -5 gpurun_out/r06_loop4_chain_tests.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest2.log 2>&1; tail -5 gpurun_out/r06_gputest2.log
