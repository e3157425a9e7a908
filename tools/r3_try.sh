#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
python bench.py --steps 20 --warmup 5 --cpu-seconds 2 > gpurun_out/bench_r3c.json 2> gpurun_out/bench_r3c.err; tail -c 300 gpurun_out/bench_r3c.err
