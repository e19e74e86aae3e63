#!/bin/bash
# end-of-round evidence run: GPU test suite, the profile round (rocprofv3 kernel traces + PMC passes -> summaries), the
# driver-like default bench line and a long-run one.  Usage on the GPU box: bash scripts/gpu_final.sh TAG
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/${TAG}_logs
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_logs/pytest_gpu.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/${TAG}_logs/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_logs/smoke.log 2>&1; echo "smoke rc $?"; tail -1 gpurun_out/${TAG}_logs/smoke.log
bash scripts/profile_round.sh $TAG > gpurun_out/${TAG}_logs/profile_round.log 2>&1; echo "profile round rc $?"
cp gpurun_out/prof_$TAG/*.log gpurun_out/${TAG}_logs/ 2>/dev/null; rm -rf gpurun_out/prof_$TAG
# (the bench reads profiles/traffic_latest.json: the fresh one is put in place for the two lines below)
cp gpurun_out/${TAG}_traffic_latest.json profiles/traffic_latest.json
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver_like.json 2> gpurun_out/${TAG}_logs/bench_driver_like.err; echo "bench (20 steps) rc $?"
timeout 900 python bench.py --steps 1000 --warmup 100 --no-cpu-baseline > gpurun_out/${TAG}_bench_1000.json 2> gpurun_out/${TAG}_logs/bench_1000.err; echo "bench (1000 steps) rc $?"
# round 6: the chained form's attribution on the final binary (timing build), the wait-side counters, power + clock
T2D_LIB_NAME=libt2d_hip_timing.so timeout 300 python scripts/chain_timing.py 20 gpurun_out/${TAG}_chain_timing_frag20.json > gpurun_out/${TAG}_logs/chain_timing.log 2>&1; echo "chain timing rc $?"
MODE=chain PASSES=3 bash scripts/sq_wait_chain.sh $TAG > gpurun_out/${TAG}_logs/sq_wait_chain.log 2>&1; echo "sq wait rc $?"
T2D_PC_SECONDS=2.5 timeout 300 python scripts/power_clock.py gpurun_out/${TAG}_power_clock.json > gpurun_out/${TAG}_logs/power_clock.log 2>&1; echo "power clock rc $?"
timeout 600 python tests/soak/soak.py 60 > gpurun_out/${TAG}_logs/soak.log 2>&1; echo "soak rc $?"; tail -2 gpurun_out/${TAG}_logs/soak.log
timeout 600 python tests/soak/soak_lidar.py 150 > gpurun_out/${TAG}_logs/soak_lidar.log 2>&1; echo "soak lidar rc $?"; tail -2 gpurun_out/${TAG}_logs/soak_lidar.log
timeout 600 python tests/soak/soak_mapgrid.py 24 > gpurun_out/${TAG}_logs/soak_mapgrid.log 2>&1; echo "soak mapgrid rc $?"; tail -1 gpurun_out/${TAG}_logs/soak_mapgrid.log
timeout 300 python scripts/mapgrid_timing.py 1024 > gpurun_out/${TAG}_logs/mapgrid_timing.log 2>&1; echo "mapgrid timing rc $?"; grep '^grid_tier\|^lds_record' gpurun_out/${TAG}_logs/mapgrid_timing.log > gpurun_out/${TAG}_mapgrid_timing.txt
du -sh gpurun_out
