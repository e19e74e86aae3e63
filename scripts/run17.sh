cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/r17_gpu_tests.log 2>&1; echo "gpu tests rc=$?" >> gpurun_out/r17_gpu_tests.log
tail -4 gpurun_out/r17_gpu_tests.log | head -3
