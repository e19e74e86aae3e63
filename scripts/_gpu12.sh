cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_mapgrid.py -x -q > gpurun_out/r06_gputest7.log 2>&1; tail -40 gpurun_out/r06_gputest7.log
