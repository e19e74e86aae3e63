cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T2D_LIB_NAME=libt2d_loop4_timing.so timeout 300 python scripts/loop_timing.py 20 10 > gpurun_out/r06_loop4_timing.json 2> gpurun_out/r06_loop4_timing.err; tail -c 1500 gpurun_out/r06_loop4_timing.json
T2D_AB_ONLY=metric timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_loop4s.so libt2d_loop4s2.so > gpurun_out/r06_ab_loop4_stagger.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_loop4_stagger.txt
T2D_LIB_NAME=libt2d_loop4.so T2D_SQ_FILTER='collide_kernel<true, 1, false, false, true' T2D_SQ_STEPS_PER_WAVE=32 PASSES=3 MODE=chain bash scripts/sq_wait_chain.sh r06loop4 > gpurun_out/r06_sq_wait_loop4.log 2>&1; tail -c 1200 gpurun_out/r06_sq_wait_loop4.log
