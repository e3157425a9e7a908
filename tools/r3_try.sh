#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
for i in 1 2 3; do
python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c230-330,400-520 | tr '\n' ' '; echo
RUMI_TRACK_PINNED=0 python tools/track_probe.py 2>&1 | grep -v amdgpu | tail -2 | cut -c230-330,400-520 | tr '\n' ' '; echo " (2D from pageable)"
done
