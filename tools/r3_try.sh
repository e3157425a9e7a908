#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_single
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof_single -- python3 $R/tools/prof_single.py > $O/prof_single.log 2>&1
tail -1 $O/prof_single.log
