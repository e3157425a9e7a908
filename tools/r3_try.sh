#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
RUMI_SERIAL=1 python tools/stage_probe.py 1000 2>&1 | grep fast
for i in 1 2; do python bench.py --steps 30 --warmup 5 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fps', d['value'], d['ms_per_step'])"; done
