cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
O=gpurun_out/r06_ab_chain_depth.txt; : > $O
timeout 600 python scripts/ab_step.py libt2d_hip.so libt2d_ck.so >> $O 2>&1
for D in 2 4 5 10; do echo "== T2D_CHAIN_DEPTH=$D" >> $O; T2D_AB_ONLY=metric T2D_CHAIN_DEPTH=$D timeout 600 python scripts/ab_step.py libt2d_ck.so >> $O 2>&1; done
grep "AB_RESULT\|==" $O
T2D_LIB_NAME=libt2d_ck.so T2D_CHAIN_DEPTH=4 timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_configs.py tests/test_gpu_chain_oracle.py -x -q > gpurun_out/r06_ck_tests.log 2>&1; tail -5 gpurun_out/r06_ck_tests.log
