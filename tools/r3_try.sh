#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
python bench.py 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', d['value'], d['ms_per_step'], d.get('batch_sweep_fps'), d.get('records_path_fps'))"
