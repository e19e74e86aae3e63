cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r06_gputest4.log 2>&1; tail -5 gpurun_out/r06_gputest4.log
timeout 300 python scripts/time_integrate.py fast > gpurun_out/r06b_time_integrate.log 2>&1; tail -14 gpurun_out/r06b_time_integrate.log
T2D_AB_ONLY=metric timeout 900 python scripts/ab_step.py libt2d_late0.so libt2d_hip.so libt2d_late12.so libt2d_late20.so libt2d_hip.so > gpurun_out/r06_ab_late3.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_late3.txt
