# Round profile of bench.py on the GPU box (run from the repo root): kernel stats of the three fp32
# plans, of the bf16 / C = 80 plan (configs[3]) and of the TTA inference (configs[4]) + the two PMC
# passes (FETCH_SIZE, WRITE_SIZE) and an SQ pass of the default plan + an SQ pass of the inference;
# summarise with tools/summarize_profile.py / pmc_default_plan.py into profiles/.
export TMPDIR=/tmp
R=${1:-r05}
O=gpurun_out/prof_$R
B="python bench.py --no-cpu-baseline --no-alt-plan --no-extra-configs --no-parity-check --no-projection"
for P in fp16x2 fp32x3 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_stats -o st -- $B --mfma-dtype $P --steps 20 --warmup 5 > $O.$P.log 2>&1
  tail -1 $O.$P.log | cut -c1-300
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bf16_stats -o st -- $B --mfma-dtype bf16 --classes 80 --steps 20 --warmup 5 > $O.bf16.log 2>&1
tail -1 $O.bf16.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/infer_stats -o st -- python bench.py --infer > $O.infer.log 2>&1
tail -1 $O.infer.log | cut -c1-300
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fp16x2_fetch -o pf -- $B --steps 2 --warmup 1 > $O.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fp16x2_write -o pw -- $B --steps 2 --warmup 1 > $O.write.log 2>&1
# the bf16 / C = 80 plan's fc6 forward (its roofline had traffic: null)
BB="$B --mfma-dtype bf16 --classes 80"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bf16_fetch -o pf -- $BB --steps 2 --warmup 1 > $O.bf16fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bf16_write -o pw -- $BB --steps 2 --warmup 1 > $O.bf16write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/fp16x2_sq -o ps -- $B --steps 3 --warmup 1 > $O.sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/infer_sq -o ps -- python bench.py --infer --steps 8 --warmup 4 > $O.infersq.log 2>&1
ls $O
