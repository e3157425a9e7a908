#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 900 python -m pytest tests/test_extractor_gpu.py -x -q -m gpu > gpurun_out/oct_tests.log 2>&1; rc=$?
tail -3 gpurun_out/oct_tests.log
[ $rc -eq 0 ] || exit $rc
RUMI_SERIAL=1 python tools/stage_probe.py 1000 2000 5000 2>&1 | grep -v amdgpu | cut -c1-6,90-140
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', d['value'], d['ms_per_step'])"; done
