cd na-fwebsod_amd
O=../gpurun_out/r06_soak_update_raw.txt
: > $O
for ARGS in "--train-step 1" "--train-step 0" "--train-step 0 --exchange 2 --pipeline 0" "--train-step 1 --exchange 2 --pipeline 1" "--train-step 1 --exchange 8 --pipeline 1" "--train-step 1 --exchange 2 --pipeline 1 --switch-route"; do
  echo "== soak_update.py --steps 1000 --check 250 --lr 1e-5 --digest $ARGS" >> $O
  python tools/soak_update.py --steps 1000 --check 250 --lr 1e-5 --digest $ARGS >> $O 2>&1
done
cd ..
python bench.py > gpurun_out/r06_bench_line2.json 2> gpurun_out/r06_bench_line2.err
tail -3 gpurun_out/r06_soak_update_raw.txt
