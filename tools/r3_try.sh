#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python tools/single_frame_probe.py 2>&1 | grep -v amdgpu | tail -2
python tools/single_frame_probe.py 2>&1 | grep -v amdgpu | tail -2
