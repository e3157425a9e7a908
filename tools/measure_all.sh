#!/bin/bash
# One full measurement pass on the GPU box (run through gpurun); outputs land in gpurun_out/, then `python tools/collect_profiles.py` (round tag r02).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd $R
python bench.py --steps 20 --warmup 5 > $O/bench_r02.json 2> $O/bench_r02.err
cd /tmp; export TMPDIR=/tmp
rm -rf $O/prof_default $O/prof_serial $O/pmc_fetch $O/pmc_write $O/prof_lba $O/pmc_a $O/pmc_b
# the bench command as timed (sub-chunks of 64 frames pipelined over 4 streams), and with every kernel alone on one stream in 256-frame launches
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu > $O/prof_default.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_serial -- python3 $R/bench.py --steps 10 --warmup 3 --batch 256 --no-cpu > $O/prof_serial.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_fetch.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_write.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_a.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_b.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_lba -- python3 $R/tools/prof_lba.py > $O/prof_lba.log 2>&1
cd $R
RUMI_SERIAL=1 python tools/stage_probe.py 1000 2000 5000 > $O/stage_serial.log 2>&1
python tools/host_batch_probe.py > $O/host_batch.log 2>&1
timeout -k 10 60 tools/bin/valu_rate > $O/r02_valu_issue_rates.txt 2>&1
cat $O/stage_serial.log $O/host_batch.log
cut -c1-250 $O/bench_r02.json
