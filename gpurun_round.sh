export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x 2>&1 | tail -5 > gpurun_out/gpu_tests_h2.log
python bench.py > gpurun_out/bench_h2.log 2>&1
bash na-fwebsod_amd/tools/profile_bench.sh > gpurun_out/profile_h2.log 2>&1
python na-fwebsod_amd/tools/summarize_profile.py gpurun_out/prof_h2_stats gpurun_out/prof_h2_fetch gpurun_out/prof_h2_write gpurun_out/r01_bench_fp16x2 fp16x2 > /dev/null 2>&1
cat gpurun_out/gpu_tests_h2.log; tail -1 gpurun_out/bench_h2.log | cut -c1-2500
