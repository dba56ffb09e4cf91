export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bf16prof -o p -- python bench.py --mfma-dtype bf16 --classes 80 --no-cpu-baseline --no-alt-plan --no-extra-configs --no-parity-check --no-projection --steps 20 --warmup 3 > gpurun_out/bf16prof.log 2>&1
tail -1 gpurun_out/bf16prof.log | cut -c1-200
python - <<PY
import csv,glob
f=glob.glob("gpurun_out/bf16prof/**/*kernel_stats.csv", recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:24]:
    print(r["Name"][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"])
PY
