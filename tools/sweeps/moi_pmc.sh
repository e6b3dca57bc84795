cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/moipmc
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA -d $O/p1 -o p --output-format csv -- python tools/sweeps/moi_only.py > $O/p1.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE -d $O/p2 -o p --output-format csv -- python tools/sweeps/moi_only.py > $O/p2.log 2>&1
echo done $?
