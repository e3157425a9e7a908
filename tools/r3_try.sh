#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_track_frame_gpu.py tests/test_track_steps_gpu.py tests/test_facade_gpu.py tests/test_tracking_loop_gpu.py -x -q -m gpu > gpurun_out/oct_tests.log 2>&1; rc=$?
tail -3 gpurun_out/oct_tests.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do
python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c230-330 | head -1
done
RUMI_TRACK_SPECULATE=0 python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c230-330 | head -1 | sed 's/$/ (stage by stage)/'
