#!/bin/bash
# scratch runner for round-3 experiments on the GPU box: bench.py under several RUMI_FAST_WPG values
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; cd $R
for w in "$@"; do
  RUMI_FAST_WPG=$w python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_wpg$w.json 2> $O/bench_wpg$w.err
  python - <<PY
import json
t=open("$O/bench_wpg$w.json").read().strip().splitlines()
if t:
    d=json.loads(t[-1]); print("wpg=$w", d["value"], d["ms_per_step"], d.get("batch_sweep_fps"), d.get("extract_only_fps"), d.get("single_frame_host_api_fps"))
else:
    print("wpg=$w FAILED", open("$O/bench_wpg$w.err").read()[-800:])
PY
done
