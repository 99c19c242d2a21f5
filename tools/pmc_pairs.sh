#!/bin/bash
# developer helper (run on the GPU box): hardware counters of the ResBlock-pair kernels via tools/conv_bench
# (pairs only, 2 repetitions), one rocprofv3 pass per counter group, summary -> gpurun_out/r2_pmc_pairs.json
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
BIN=${1:-./tools/conv_bench}
OUT=gpurun_out/pmc_pairs
rm -rf $OUT
export QVC_BENCH_PAIRS=1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/a -o p --output-format csv -- $BIN 32 2 > /dev/null 2>&1 || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD -d $OUT/b -o p --output-format csv -- $BIN 32 2 > /dev/null 2>&1 || exit 2
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES -d $OUT/c -o p --output-format csv -- $BIN 32 2 > /dev/null 2>&1 || exit 3
rocprofv3 --kernel-trace --pmc TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY -d $OUT/d -o p --output-format csv -- $BIN 32 2 > /dev/null 2>&1 || exit 4
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCC_HIT TCC_MISS -d $OUT/e -o p --output-format csv -- $BIN 32 2 > /dev/null 2>&1 || exit 5
python3 tools/pmc_summary.py gpurun_out/r2_pmc_pairs.json $OUT/a $OUT/b $OUT/c $OUT/d $OUT/e
