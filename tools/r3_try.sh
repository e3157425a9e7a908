#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_track
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_track -- python3 $R/tools/track_probe.py > $O/prof_track.log 2>&1
f=$(ls $O/prof_track/*/*kernel_stats.csv | head -1); head -40 $f | cut -c1-200
