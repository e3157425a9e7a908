#!/bin/bash
# SQ counter passes of the extraction kernels, every kernel alone on its stream, 256-frame launches (run through gpurun).
# usage: pmc_fast.sh <tag> [kernel-substring]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; T=${1:-x}; K=${2:-rumi::}
cd /tmp; export TMPDIR=/tmp
rm -rf $O/pq_a_$T $O/pq_b_$T
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $O/pq_a_$T -- python3 $R/tools/stage_probe.py 1000 > $O/pq_a_$T.log 2>&1
RUMI_SERIAL=1 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $O/pq_b_$T -- python3 $R/tools/stage_probe.py 1000 > $O/pq_b_$T.log 2>&1
cd $R
python3 tools/pmc_kernel.py $O/pq_a_$T $O/pq_b_$T --like "$K"
