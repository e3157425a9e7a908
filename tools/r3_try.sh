#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/soak_optimizer.py > gpurun_out/soak_optimizer.log 2>&1; grep -v amdgpu gpurun_out/soak_optimizer.log | tail -3
