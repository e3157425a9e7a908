#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2; do
python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300
RUMI_TRACK_SPECULATE=0 python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-300 | sed 's/^/stage by stage: /'
done
