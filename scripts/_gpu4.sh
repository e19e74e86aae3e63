cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T2D_AB_ONLY=metric timeout 1200 python scripts/ab_step.py libt2d_hip.so libt2d_loop4p1.so libt2d_loop4p2.so libt2d_loop4p1s.so > gpurun_out/r06_ab_loop4_prio.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_loop4_prio.txt
