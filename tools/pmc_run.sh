#!/bin/bash
# developer helper: PMC passes over tools/conv_bench (run on the GPU box)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmc1 gpurun_out/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d gpurun_out/pmc1 -o p --output-format csv -- ./tools/conv_bench 32 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_VMEM_TA_ADDR_FIFO_FULL -d gpurun_out/pmc2 -o p --output-format csv -- ./tools/conv_bench 32 2 > /dev/null 2>&1
ls gpurun_out/pmc1 gpurun_out/pmc2
