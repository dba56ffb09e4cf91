# Round profile of bench.py on the GPU box (run from the repo root): kernel stats of the three fp32
# plans + the two PMC passes (FETCH_SIZE, WRITE_SIZE) and an SQ pass of the default plan;
# summarise with tools/summarize_profile.py into profiles/.
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/prof_$R
for P in fp16x2 fp32x3 fp32; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${P}_stats -o st -- python bench.py --no-cpu-baseline --no-alt-plan --mfma-dtype $P --steps 20 --warmup 5 > $O.$P.log 2>&1
  tail -1 $O.$P.log | cut -c1-400
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fp16x2_fetch -o pf -- python bench.py --no-cpu-baseline --no-alt-plan --steps 2 --warmup 1 > $O.fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/fp16x2_write -o pw -- python bench.py --no-cpu-baseline --no-alt-plan --steps 2 --warmup 1 > $O.write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/fp16x2_sq -o ps -- python bench.py --no-cpu-baseline --no-alt-plan --steps 3 --warmup 1 > $O.sq.log 2>&1
ls $O
