#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/with_lib.py tools/bin/librumi_hip_stamp.so tools/oct_stamp.py diagonal 2>&1 | grep "^oct" | tail -8 | cut -c1-150
timeout -k 10 900 python -m pytest tests/test_extractor_gpu.py -x -q -m gpu -k "clustered" > gpurun_out/oct_tests.log 2>&1; rc=$?
tail -15 gpurun_out/oct_tests.log
