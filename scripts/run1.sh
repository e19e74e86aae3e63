#!/bin/bash
# round-3 GPU run 1: dispatch probe, chain tests, full GPU suite, A/B of the step-kernel builds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 60 scripts/_build/probe_dispatch > gpurun_out/probe_dispatch.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_chain.py -x -q > gpurun_out/r1_chain_tests.log 2>&1
echo "chain tests rc=$?" >> gpurun_out/r1_chain_tests.log
timeout 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_chain.py > gpurun_out/r1_gpu_tests.log 2>&1
echo "gpu tests rc=$?" >> gpurun_out/r1_gpu_tests.log
timeout 900 python scripts/ab_step.py libt2d_hip.so libt2d_base.so libt2d_safe.so libt2d_filt.so > gpurun_out/r1_ab.log 2>&1
tail -3 gpurun_out/probe_dispatch.txt; tail -5 gpurun_out/r1_chain_tests.log; tail -5 gpurun_out/r1_gpu_tests.log; grep AB_RESULT gpurun_out/r1_ab.log
