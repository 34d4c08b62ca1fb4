# PMC passes of the round: one counter group per rocprofv3 run (FETCH_SIZE and WRITE_SIZE cannot share a pass),
# each under its own timeout.  Run on the GPU box:  bash tools/pmc_passes.sh ; then python tools/pmc_collect.py gpurun_out/pmc_${TAG:-r1f}_?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { n=$1; shift; echo "pass $n: $*" >> $R/gpurun_out/pmc_${TAG:-r1f}_progress.log; timeout -k 10 150 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${TAG:-r1f}_$n -- python3 $R/tools/pmc_probe.py > $R/gpurun_out/pmc_${TAG:-r1f}_$n.log 2>&1; echo "pass $n rc=$?" >> $R/gpurun_out/pmc_${TAG:-r1f}_progress.log; }
run a FETCH_SIZE && run b WRITE_SIZE && run c GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES && run d SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_WAVE_CYCLES && run e TCC_HIT_sum TCC_MISS_sum
cat $R/gpurun_out/pmc_${TAG:-r1f}_progress.log
