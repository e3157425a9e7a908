set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_a $O/pmc_b
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_fetch.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_write.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $O/pmc_a -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_a.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pmc_b -- python3 $R/bench.py --steps 2 --warmup 1 --batch 256 --no-cpu > $O/pmc_b.log 2>&1
