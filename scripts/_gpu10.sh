cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest5.log 2>&1; tail -8 gpurun_out/r06_gputest5.log
timeout 300 python scripts/time_integrate.py fast > gpurun_out/r06c_time_integrate.log 2>&1; tail -14 gpurun_out/r06c_time_integrate.log
