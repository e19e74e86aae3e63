cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_mapgrid.py tests/test_gpu_forms.py -x -q > gpurun_out/r06_gputest6a.log 2>&1; tail -30 gpurun_out/r06_gputest6a.log
timeout 1800 python -m pytest tests -m gpu -q > gpurun_out/r06_gputest6.log 2>&1; tail -12 gpurun_out/r06_gputest6.log
timeout 300 python scripts/time_integrate.py fast > gpurun_out/r06c_time_integrate.log 2>&1; tail -14 gpurun_out/r06c_time_integrate.log
