#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_optimizer_gpu.py tests/test_facade_gpu.py -x -q -m gpu > gpurun_out/oct_tests.log 2>&1; rc=$?
tail -3 gpurun_out/oct_tests.log
