#!/bin/bash
# Round profile: the bench line, rocprofv3 kernel stats of the same command, and HBM-traffic PMC passes.
export TMPDIR=/tmp
cd "$(dirname "$0")/.."
rm -rf gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
python bench.py --steps 20 --warmup 3 > gpurun_out/bench_final.json 2> gpurun_out/bench_final.err || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o r01 --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_prof.json 2> gpurun_out/prof.err || exit 2
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -o f --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2> gpurun_out/pmc_fetch.err || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -o w --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > /dev/null 2> gpurun_out/pmc_write.err || exit 4
ls gpurun_out/prof gpurun_out/pmc_fetch gpurun_out/pmc_write
# speaker-encoder launches: kernel trace of tools/spk_bench.py (also runs torch.nn.LSTM for comparison)
rm -rf gpurun_out/prof_spk
python tools/spk_bench.py 1 32 128 > gpurun_out/spk_bench.txt 2>&1 || exit 5
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_spk -o spk --output-format csv -- python3 tools/spk_bench.py 32 > gpurun_out/spk_prof.log 2>&1 || exit 6
