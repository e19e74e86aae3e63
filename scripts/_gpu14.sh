cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_physics.py -x -q 2>&1 | tail -3
T2D_TI_ENVS=65536 timeout 300 python scripts/time_integrate.py fast 2>&1 | tail -6 | tee gpurun_out/r06d_time_integrate.log
