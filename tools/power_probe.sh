cd $GRAFT_REPO_ROOT
rocm-smi --showpower --showclocks --showtemp > gpurun_out/smi_idle.txt 2>&1
( for i in $(seq 1 24); do rocm-smi --showpower --showclocks --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.5; done > gpurun_out/smi_samples.txt ) &
SMI=$!
timeout -k 10 200 python bench.py --steps 4000 --warmup 5 --no-cpu-baseline > gpurun_out/bench_power.json 2> gpurun_out/bench_power.err
wait $SMI
echo done
