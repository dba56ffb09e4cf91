# Round profile of bench.py on the GPU box (run from the repo root): kernel stats + the two PMC
# passes (FETCH_SIZE, WRITE_SIZE); summarise with tools/summarize_profile.py into profiles/.
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_h2_stats -o st -- python bench.py --no-cpu-baseline --no-alt-plan > gpurun_out/bench_h2_prof.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_h2_fetch -o pf -- python bench.py --no-cpu-baseline --no-alt-plan --steps 2 --warmup 1 > gpurun_out/pmc_h2_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_h2_write -o pw -- python bench.py --no-cpu-baseline --no-alt-plan --steps 2 --warmup 1 > gpurun_out/pmc_h2_write.log 2>&1
tail -1 gpurun_out/bench_h2_prof.log | cut -c1-300
ls gpurun_out/prof_h2_stats gpurun_out/prof_h2_fetch gpurun_out/prof_h2_write
