# Everything under profiles/rNN_* in one GPU-box call (run from the repo root):
#   bash na-fwebsod_amd/tools/round_profile.sh r05
# raw rocprofv3 output -> gpurun_out/prof_rNN/, summaries -> gpurun_out/rNN_profiles/ (copy those
# into profiles/ afterwards: only gpurun_out/ travels back from the box).
R=${1:-r05}
T=na-fwebsod_amd/tools
O=gpurun_out/prof_$R
P=gpurun_out/${R}_profiles
mkdir -p $P
bash $T/profile_bench.sh $R
for M in fp16x2 fp32x3 fp32 bf16; do
  F=$O/${M}_stats; W=$O/${M}_stats
  if [ $M = fp16x2 ] || [ $M = bf16 ]; then F=$O/${M}_fetch; W=$O/${M}_write; fi
  python $T/summarize_profile.py $O/${M}_stats $F $W $P/${R}_bench_$M $M > /dev/null
done
python $T/summarize_profile.py $O/infer_stats $O/infer_stats $O/infer_stats $P/${R}_infer fp16x2 --allow-missing > /dev/null
mv $P/${R}_infer.md $P/${R}_infer_kernel_stats.md
python $T/pmc_default_plan.py $O/fp16x2_sq $P/${R}_default_plan_pmc.md > /dev/null
python $T/pmc_default_plan.py $O/infer_sq $P/${R}_infer_pmc.md > /dev/null
python $T/step_timeline.py $O/fp16x2_stats full > $P/${R}_step_timeline.txt 2>&1
python bench.py > $P/${R}_bench_line.json 2> $P/${R}_bench_line.err
python bench.py --infer > $P/${R}_infer_line.json 2>/dev/null
# configs[2]'s schedule with four ranks on this one GPU over gloo (a diagnostic, not a rate; the
# box's process guard allows six processes on the GPU: rounds 4-5 ran eight ranks here)
python bench.py --gpus 4 --share-gpu --steps 6 --warmup 2 2>/dev/null | tail -1 > $P/${R}_share_gpu4_line.json
# kernel timelines of one step of the 2-rank projection, update piece by piece / in one launch
rocprofv3 --kernel-trace --output-format csv -d $O/n2_trace -o tr -- python bench.py --no-cpu-baseline --no-alt-plan --no-extra-configs --no-parity-check --emulate-exchange 2 --steps 20 --warmup 5 > $O.n2.log 2>&1
python $T/exchange_timeline.py $O/n2_trace pipelined 100 > $P/${R}_n2_projection_timeline_pipelined.txt 2>&1
python $T/exchange_timeline.py $O/n2_trace unpipelined 100 > $P/${R}_n2_projection_timeline_unpipelined.txt 2>&1
python -m pytest tests/test_gpu_fullsize_oracle.py -q -s 2>&1 | grep -E "^\[|^\.\[|passed|failed" > $P/${R}_fullsize_parity_raw.txt
( cd na-fwebsod_amd
  python tools/soak_crossplan.py --steps 200 --lr 1e-5 --out ../$P/${R}_soak_lr1e-5.md > /dev/null 2>&1
  python tools/soak_crossplan.py --steps 200 --lr 1e-4 --out ../$P/${R}_soak_lr1e-4.md > /dev/null 2>&1
  python tools/soak_crossplan.py --steps 200 --lr 1e-8 --skewed --out ../$P/${R}_soak_lr1e-8_skewed.md > /dev/null 2>&1 )
ls -la $P
tail -c 300 $P/${R}_bench_line.err
