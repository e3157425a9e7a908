#!/bin/bash
# Second measurement pass (latency paths): outputs land in gpurun_out/, collected by tools/collect_profiles.py.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_track $O/prof_lba $O/prof_single
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_single -- python3 $R/tools/prof_single.py > $O/prof_single.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_track -- python3 $R/tools/track_loop.py > $O/track_loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lba -- python3 $R/tools/prof_lba.py > $O/prof_lba.log 2>&1
cd $R
python tools/track_probe.py > $O/track_probe.log 2>&1
python tools/single_frame_probe.py > $O/single_frame.log 2>&1
RUMI_BENCH_LOGICAL_SHARDS=1 python bench.py --one-process --gpus 2 --steps 10 --warmup 3 --no-cpu > $O/bench_one_process.json 2> $O/bench_one_process.err
python tools/lba_probe.py 20 12 28 > $O/lba_probe.log 2>&1
python tools/queue_probe.py 512 2>&1 | grep shard > $O/queue_probe.log
python tools/bow_batch_probe.py > $O/bow_batch.log 2>&1
python tools/pose_probe.py > $O/pose_probe.log 2>&1
tools/bin/rsq_probe > $O/rsq_probe.log 2>&1
python tools/bench_matrix.py > $O/bench_matrix.log 2>&1
cat $O/track_probe.log $O/lba_probe.log $O/bow_batch.log $O/pose_probe.log $O/rsq_probe.log | grep -v amdgpu
