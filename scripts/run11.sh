cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r11_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r11_gpu_tests.log
tail -12 gpurun_out/r11_gpu_tests.log
T2D_AB_CHAIN=1 timeout 600 python scripts/ab_step.py libt2d_hip.so > gpurun_out/r11_ab.log 2>&1; grep AB_RESULT gpurun_out/r11_ab.log
