#!/bin/bash
# the randomised soaks at ten times gpu_final.sh's length (events, stepping, lidar, generator, grid tier): bash scripts/long_soak.sh TAG
TAG=${1:-soak}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
{
timeout 1500 python tests/soak/soak.py 400 2>&1 | tail -2
timeout 1200 python tests/soak/soak_lidar.py 1200 2>&1 | tail -1
[ -f tests/soak/soak_generate.py ] && timeout 1200 python tests/soak/soak_generate.py 2>&1 | tail -1
timeout 1200 python tests/soak/soak_mapgrid.py 200 2>&1 | tail -1
} | grep -v amdgpu.ids | tee gpurun_out/${TAG}_long_soak.txt
