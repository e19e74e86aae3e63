cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T2D_AB_ONLY=metric timeout 1500 python scripts/ab_step.py libt2d_late0.so libt2d_hip.so libt2d_late2.so libt2d_late3.so libt2d_late5.so libt2d_late8.so libt2d_late0.so libt2d_late3.so libt2d_late5.so > gpurun_out/r06_ab_late2.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_late2.txt
