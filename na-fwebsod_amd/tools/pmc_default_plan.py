#!/usr/bin/env python3
"""profiles/rNN_default_plan_pmc.md from the SQ counter pass of tools/profile_bench.sh
(`<dir>/fp16x2_sq`): per hot kernel and problem shape - clock, MFMA busy, wait shares.

    python tools/pmc_default_plan.py gpurun_out/prof_r02b/fp16x2_sq profiles/r02_default_plan_pmc.md"""
import collections
import csv
import glob
import sys

HOT = ('gemm_x3', 'gemm_h2_btr', 'conv_h2', 'conv_x3', 'roi_pool', 'acm_sgd', 'split2h_dual', 'gemm_smallk', 'wino_',
       'wino4_', 'conv_c3')
# (template prefix, grid, MFMA-busy cycles / 1e7) -> what the launch is; prefixes, because template
# argument lists grow (round 4 appended one and every exact-name label stopped matching)
LABEL = {('gemm_h2_btr_kernel<256, 256, 4, 2, true', '1572864', None): 'fc6 wgrad + SGD epilogue',
         ('gemm_h2_btr_kernel<256, 256, 4, 2, false', '1572864', None): 'fc6 wgrad, gradient written (deferred route)',
         ('gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, true', '262144', 493): 'fc6 fwd (M=4000 N=8192 K=25088)',
         ('gemm_x3_kernel<128, 128, 2, 2, 3, 2, 1, true', '311296', None): 'Winograd F(2x2) batch GEMM (16 x [2344 tiles x 512 x 512])',
         ('gemm_x3_kernel<128, 128, 2, 2, 3, 2, 1, true', '184320', None): 'Winograd F(4x4) batch GEMM (36 x [608 or 640 tiles x 512 x Cin])',
         ('roi_pool_nhwc_xcd_kernel<true, true', None, None): 'RoIPoolF + boost -> fc6 operand planes'}


def label_of(key):
    for (pre, grid, work), text in LABEL.items():
        if key[0].startswith(pre) and (grid is None or key[1] == grid) and \
                (work is None or abs(key[2] - work) <= 0.06 * work):
            return text
    return None


def main():
    d, out = sys.argv[1:3]
    cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
           for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(cc)):
        e = per[r['Dispatch_Id']]
        e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        e['_k'] = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        e['_g'] = r['Grid_Size']
    groups = collections.defaultdict(list)
    for did, v in per.items():
        if not any(x in v['_k'] for x in HOT):
            continue
        cyc = v['GRBM_GUI_ACTIVE'] / 8
        key = (v['_k'], v['_g'], round(v['SQ_VALU_MFMA_BUSY_CYCLES'] / 1e7))
        groups[key].append((dur[did], cyc / (dur[did] * 1e-3) / 1e9,
                            v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024),
                            v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
                            v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], v['SQ_LDS_BANK_CONFLICT']))
    lines = ["# PMC reading of the default plan's kernels inside `bench.py`", '',
             '`rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES '
             'SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE -- python '
             'bench.py --no-cpu-baseline --no-alt-plan --steps 3 --warmup 1`',
             '(one counter pass, `na-fwebsod_amd/tools/profile_bench.sh`; under counter collection kernels are '
             'serialised, so durations are 5-12 % above the un-profiled ones and concurrent streams do not '
             'overlap).', '',
             'clock = GRBM_GUI_ACTIVE / 8 XCDs / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x '
             '1024 SIMDs); waits are fractions of SQ_WAVE_CYCLES.  Rows are grouped by (kernel, grid, MFMA '
             'work), so one template appears once per distinct problem shape; "clock" above 2.4 GHz on '
             'the 30 us kernels is GRBM_GUI_ACTIVE counting the dispatch ramp around a launch shorter than '
             "the counter's window, not a real clock.", '',
             '| kernel (grid threads) | launches | avg ms | clock GHz | MFMA busy | issue-stalled '
             '(WAIT_INST_ANY) | parked (WAIT_ANY) | LDS bank conflict cycles |',
             '|---|---|---|---|---|---|---|---|']
    for key, v in sorted(groups.items(), key=lambda kv: -sum(x[0] for x in kv[1])):
        n = len(v)
        avg = [sum(x[i] for x in v) / n for i in range(6)]
        name = key[0] + ' (' + key[1] + ')'
        if label_of(key):
            name += ' = ' + label_of(key)
        lines.append('| %s | %d | %.3f | %.2f | %.0f %% | %.0f %% | %.0f %% | %d |' % (
            name, n, avg[0], avg[1], 100 * avg[2], 100 * avg[3], 100 * avg[4], avg[5]))
    tail = sys.argv[3] if len(sys.argv) > 3 else None
    if tail:
        lines += ['', open(tail).read().rstrip()]
    open(out, 'w').write('\n'.join(lines) + '\n')
    print('\n'.join(lines[8:40]))


if __name__ == '__main__':
    main()
