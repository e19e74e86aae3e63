cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r7_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r7_gpu_tests.log
tail -15 gpurun_out/r7_gpu_tests.log
T2D_AB_CHAIN=1 timeout 600 python scripts/ab_step.py libt2d_hip.so > gpurun_out/r7_ab.log 2>&1; grep AB_RESULT gpurun_out/r7_ab.log
bash scripts/sq_variants.sh libt2d_hip.so 2>&1 | grep "^libt2d"
for K in hw rb ix; do T2D_COUNT_CONFIG=$K T2D_COUNT_STEPS=100 bash scripts/sq_variants.sh libt2d_hip.so; done 2>&1 | grep "^libt2d"
