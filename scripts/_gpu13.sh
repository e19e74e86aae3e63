cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python scripts/ab_ego.py libt2d_hip.so libt2d_lidar2.so libt2d_hip.so libt2d_lidar2.so > gpurun_out/r06_ab_lidar_one_wave.txt 2>&1; grep AB_RESULT gpurun_out/r06_ab_lidar_one_wave.txt
for L in libt2d_hip.so libt2d_lidar2.so; do T2D_LIB_NAME=$L timeout 200 python scripts/time_lidar.py 2>&1 | grep "cfg2"; done | tee gpurun_out/r06_time_lidar_one_wave.txt
timeout 600 python -m pytest tests/test_lidar.py tests/test_gpu_envs.py tests/test_gpu_configs.py -m gpu -x -q 2>&1 | tail -4
timeout 300 python scripts/_gpu13.py 2>&1 | grep "us per step" | tee gpurun_out/r06_host_path_beams.txt
T2D_LIB_NAME=libt2d_hip_timing.so timeout 200 python scripts/lidar_phases.py 2>&1 | tail -8 | tee gpurun_out/r06_lidar_phases.txt
