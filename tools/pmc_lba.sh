# PMC passes (each its own run, --kernel-trace only beside --pmc) over the window-batched local BA: 16 windows of BASELINE configs[3] in ONE launch
# group (RUMI_BAW_GROUPS=1: kernels do not overlap, so per-kernel counters and durations belong together), then one window.
# -> gpurun_out/pmc_lba_{mfma,sq,mem}{16,1}/ ; tools/pmc_lba_summary.py turns them into profiles/r04_lba_pmc_mfma.json
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp; export TMPDIR=/tmp
for W in 16 0; do
  T=$([ $W = 0 ] && echo 1 || echo 16)
  rm -rf $O/pmc_lba_mfma$T $O/pmc_lba_sq$T $O/pmc_lba_time$T
  RUMI_BAW_GROUPS=1 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_VALU_MFMA_F64 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_lba_mfma$T -- python3 $R/tools/prof_lba_batch.py $W 2 > $O/pmc_lba_mfma$T.log 2>&1
  RUMI_BAW_GROUPS=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/pmc_lba_sq$T -- python3 $R/tools/prof_lba_batch.py $W 2 > $O/pmc_lba_sq$T.log 2>&1
  RUMI_BAW_GROUPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pmc_lba_time$T -- python3 $R/tools/prof_lba_batch.py $W 3 > $O/pmc_lba_time$T.log 2>&1
done
