#!/bin/bash
# One full measurement pass on the GPU box (run through gpurun); outputs land in gpurun_out/, then `python tools/collect_profiles.py` (round tag $TAG, default r04).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; TAG=${TAG:-r04}
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_$TAG.json 2> $O/bench_$TAG.err
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_default $O/prof_serial $O/prof_serial128 $O/pmc_fetch $O/pmc_write $O/prof_lba $O/pmc_a $O/pmc_b
# the bench command as timed (sub-chunks of 64 frames pipelined over 4 streams), and with every kernel alone on one stream in 256-frame launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $O/prof_default.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --steps 10 --warmup 3 --batch 256 --no-cpu > $O/prof_serial.log 2>&1
# the same serial pass in 128-frame launches: a rank's share of BASELINE configs[4]
RUMI_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial128 -- python3 $R/bench.py --steps 10 --warmup 3 --batch 128 --no-cpu > $O/prof_serial128.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_fetch.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_write.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_a.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lba -- python3 $R/tools/prof_lba.py > $O/prof_lba.log 2>&1
cd $R
# the N > 1 code path (per-frame records + the exchange step) on this one GPU
RUMI_BENCH_FORCE_RECORDS=1 python bench.py --steps 20 --warmup 5 --no-cpu > $O/bench_${TAG}_records.json 2> $O/bench_${TAG}_records.err
RUMI_SERIAL=1 python tools/stage_probe.py 1000 2000 5000 > $O/stage_serial.log 2>&1
python tools/host_batch_probe.py > $O/host_batch.log 2>&1
timeout -k 10 60 tools/bin/valu_rate > $O/${TAG}_valu_issue_rates.txt 2>&1
cat $O/stage_serial.log $O/host_batch.log
cut -c1-250 $O/bench_$TAG.json
