#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python tools/soak_extractor.py 60 > gpurun_out/soak_extractor.log 2>&1; rc=$?
grep -v amdgpu gpurun_out/soak_extractor.log | tail -2
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python tools/soak_matcher.py > gpurun_out/soak_matcher.log 2>&1; rc=$?
grep -v amdgpu gpurun_out/soak_matcher.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
