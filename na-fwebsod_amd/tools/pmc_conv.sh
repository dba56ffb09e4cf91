# PMC passes over tools/ab_conv.py for one pipeline (NAWS ring id $1, default 11), two images.
export TMPDIR=/tmp
R=${1:-11}
O=gpurun_out/pmc_conv_$R
cd na-fwebsod_amd
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d ../$O/p1 -o p -- python tools/ab_conv.py --rings $R --images 2 --rounds 2 > ../$O.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d ../$O/p2 -o p -- python tools/ab_conv.py --rings $R --images 2 --rounds 2 > ../$O.p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d ../$O/p3 -o p -- python tools/ab_conv.py --rings $R --images 2 --rounds 2 > ../$O.p3.log 2>&1
tail -3 ../$O.p3.log
ls ../$O/*
