export TMPDIR=/tmp
python -m pytest tests/test_gpu_h2.py tests/test_gpu_engine.py -q -x 2>&1 | tail -3
python na-fwebsod_amd/tools/x3_accuracy.py > gpurun_out/h2_accuracy.md 2>&1
cd na-fwebsod_amd
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d ../gpurun_out/h2_pmc1 -o p1 -- python tools/kernel_bench.py --what h2 --iters 1 > ../gpurun_out/h2_pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d ../gpurun_out/h2_pmc2 -o p2 -- python tools/kernel_bench.py --what h2 --iters 1 > ../gpurun_out/h2_pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS --output-format csv -d ../gpurun_out/h2_pmc3 -o p3 -- python tools/kernel_bench.py --what h2 --iters 1 > ../gpurun_out/h2_pmc3.log 2>&1
cd ..
python - <<'PY'
import csv, glob, collections
for d in ('h2_pmc1', 'h2_pmc2', 'h2_pmc3'):
    for f in glob.glob('gpurun_out/%s/**/*counter_collection.csv' % d, recursive=True):
        rows = list(csv.DictReader(open(f)))
        agg = collections.OrderedDict()
        for r in rows:
            if 'gemm_x3_kernel' not in r['Kernel_Name']: continue
            key = (r['Dispatch_Id'], r['Grid_Size'], r['Kernel_Name'][:60])
            agg.setdefault(key, {})[r['Counter_Name']] = float(r['Counter_Value'])
        for k, v in list(agg.items())[-8:]:
            print(d, k[1], k[2][-30:], {a: '%.4g' % b for a, b in v.items()})
    for f in glob.glob('gpurun_out/%s/**/*kernel_trace.csv' % d, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if 'gemm_x3_kernel' in r['Kernel_Name']]
        for r in rows[-8:]:
            print(d, 'dur', r['Grid_Size'], (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6, 'ms')
PY
