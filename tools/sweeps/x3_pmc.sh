# PMC passes over the 256 x 256-tile contraction layers (tools/sweeps/x3_big.py); tables under gpurun_out/x3pmc/
O=gpurun_out/x3pmc; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/p1 -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_INSTS_VMEM_RD SQ_INSTS_VALU -d $O/p2 -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/p2.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC -d $O/p3 -o p --output-format csv -- python tools/sweeps/x3_big.py > $O/p3.log 2>&1
echo rc=$?
