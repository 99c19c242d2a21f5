#!/bin/bash
# developer helper: SQ counters for every kernel of the real step (eager launches)
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/pmc_sq
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc_sq -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2> gpurun_out/pmc_sq.err
ls gpurun_out/pmc_sq
