#!/bin/bash
# Socket power, clock and temperature (rocm-smi, every 0.5 s) while (a) the bench runs 4000 steps, (b) tools/conv_bench
# runs only the WaveNet stack launches, (c) only the ResBlock pairs -- each for a few seconds.  Run on the GPU box.
cd "$(dirname "$0")/.."
sample() {   # $1 = output file, $2 = number of samples
  for i in $(seq 1 $2); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done > $1
}
rocm-smi --showpower --showclocks --showtemp > gpurun_out/smi_idle.txt 2>&1
sample gpurun_out/smi_bench.txt 24 & S=$!
timeout -k 10 200 python bench.py --steps 4000 --warmup 5 --no-cpu-baseline > gpurun_out/bench_power.json 2> gpurun_out/bench_power.err
wait $S
sleep 2
sample gpurun_out/smi_wn.txt 14 & S=$!
QVC_BENCH_WN=1 timeout -k 10 100 ./tools/conv_bench 32 8000 > gpurun_out/power_wn.txt 2>&1
wait $S
sleep 2
sample gpurun_out/smi_pairs.txt 30 & S=$!
QVC_BENCH_PAIRS=1 timeout -k 10 100 ./tools/conv_bench 32 1500 > gpurun_out/power_pairs.txt 2>&1
wait $S
echo done
