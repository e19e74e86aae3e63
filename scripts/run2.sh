#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/r2_chain_tests.log 2>&1
echo "chain tests rc=$?" >> gpurun_out/r2_chain_tests.log
timeout 900 python scripts/ab_step.py libt2d_hip.so > gpurun_out/r2_ab.log 2>&1
tail -5 gpurun_out/r2_chain_tests.log; grep AB_RESULT gpurun_out/r2_ab.log
