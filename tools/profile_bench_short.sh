export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
R=r03
rm -rf gpurun_out/prof_$R gpurun_out/prof_${R}_inflight2
python bench.py --steps 20 --warmup 5 > gpurun_out/${R}_bench.json 2> gpurun_out/${R}_bench.err || exit 1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$R -o $R --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --in-flight 1 > gpurun_out/${R}_bench_prof.json 2> gpurun_out/prof.err || exit 2
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${R}_inflight2 -o $R --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${R}_bench_prof_inflight2.json 2> gpurun_out/prof2.err || exit 3
python bench.py --steps 20 --warmup 5 --batch 1 --no-cpu-baseline > gpurun_out/${R}_bench_batch1.json 2>> gpurun_out/${R}_bench.err || exit 4
echo done
