#!/bin/bash
# Round profile (run on the GPU box): HBM-traffic PMC passes (FETCH_SIZE / WRITE_SIZE, separate), then the bench line
# (which picks that traffic file up), rocprofv3 kernel stats of the same command, and SQ / TCP counter passes of the
# real step (eager launches).
# Outputs land under gpurun_out/; copy the summaries into profiles/ afterwards:
#   r03_bench.json, r03_bench_prof.json (one lane), r03_bench_batch1.json, prof_r03/r03_kernel_stats.csv,
#   prof_r03_inflight2/r03_kernel_stats.csv, r03_traffic.json, r03_sq_counters.json
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
R=${ROUND:-r03}
rm -rf gpurun_out/prof_$R gpurun_out/prof_${R}_inflight2 gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c gpurun_out/pmc_sq_e
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2> gpurun_out/pmc_fetch.err || exit 3
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2> gpurun_out/pmc_write.err || exit 4
echo "write done"
python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/${R}_traffic.json > gpurun_out/${R}_traffic.txt || exit 5
# the bench line reads profiles/${R}_traffic.json (and checks the kernel-source hash in it): put the fresh file there first
cp gpurun_out/${R}_traffic.json profiles/${R}_traffic.json
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err || exit 1
echo "bench done"
# per-kernel durations: one batch at a time (what the roofline leg's per-launch events measure), then the default
# command as it is (two batches in flight: launches of the two lanes overlap, so their durations are not per-kernel costs)
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 > gpurun_out/${R}_bench_prof.json 2> gpurun_out/prof.err || exit 2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}_inflight2 -o $R --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_bench_prof_inflight2.json 2> gpurun_out/prof2.err || exit 2
echo "kernel stats done"
python bench.py --steps 20 --warmup 5 --batch 1 --no-cpu-baseline > gpurun_out/${R}_bench_batch1.json 2>> gpurun_out/${R}_bench.err || exit 2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc_sq_a -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2> gpurun_out/pmc_sq.err || exit 6
echo "sq a done"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_VMEM_RD -d gpurun_out/pmc_sq_b -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2>> gpurun_out/pmc_sq.err || exit 7
echo "sq b done"
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES TCP_TCP_TA_DATA_STALL_CYCLES -d gpurun_out/pmc_sq_c -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2>> gpurun_out/pmc_sq.err || exit 8
echo "tcp c done"
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_LATENCY TCC_HIT TCC_MISS -d gpurun_out/pmc_sq_e -o s --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graph --in-flight 1 > /dev/null 2>> gpurun_out/pmc_sq.err || exit 9
echo "tcc e done"
python3 tools/pmc_summary.py gpurun_out/${R}_sq_counters.json gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b gpurun_out/pmc_sq_c gpurun_out/pmc_sq_e > gpurun_out/${R}_sq_counters.txt || exit 10
echo "summaries done"
