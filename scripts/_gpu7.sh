cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
T2D_AB_ONLY=metric timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_late0.so libt2d_late3.so libt2d_hip.so libt2d_late0.so > gpurun_out/r06_ab_late.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_late.txt
T2D_LIB_NAME=libt2d_hip_timing.so timeout 300 python scripts/chain_timing.py 20 gpurun_out/r06c_chain_timing_frag20.json > gpurun_out/r06c_chain_timing_frag20.log 2>&1; tail -c 300 gpurun_out/r06c_chain_timing_frag20.log
for L in libt2d_hip.so libt2d_late0.so libt2d_hip.so libt2d_late0.so; do T2D_LIB_NAME=$L timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-configs --no-next-rows --no-alternates --no-closed-loop --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L', d['value'], d['ms_per_step'])"; done | tee gpurun_out/r06_driver_like_late.txt
