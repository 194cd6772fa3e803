# What the waves of the GEMM kernels spend their cycles on (diagnostic): SQ wait / issue / FIFO-full counters per kernel of a
# training step.  Four rocprofv3 --pmc passes (each with --kernel-trace only) over one profiled step; run through gpurun.
#   BATCH=1024 bash tools/pmc_stalls.sh   -> gpurun_out/pmc_stalls/summary.md
BATCH=${BATCH:-1024}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_stalls
rm -rf $O; mkdir -p $O
export KMB_GEMM_TUNE_FILE=$GRAFT_REPO_ROOT/$O/gemm_tune.txt
python3 bench.py --batch $BATCH --steps 3 --warmup 3 --no-cpu-baseline --no-roofline --no-pcie --no-extras --serial > $O/warm.log 2>&1
P="python3 bench.py --batch $BATCH --steps 2 --warmup 2 --no-cpu-baseline --no-roofline --no-pcie --no-extras --serial"
timeout 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/wait -o w -- $P > $O/wait.log 2>&1
timeout 500 rocprofv3 --pmc SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $O/vmem -o v -- $P > $O/vmem.log 2>&1
timeout 500 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/lds -o l -- $P > $O/lds.log 2>&1
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/valu -o a -- $P > $O/valu.log 2>&1
python3 tools/pmc_stalls_summary.py $O > $O/summary.md 2>&1
rm -rf $O/*/*trace*
