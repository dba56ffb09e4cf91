#!/usr/bin/env python3
"""Kernel timeline of ONE step of the N-rank projection (bench.py --emulate-exchange N under
`rocprofv3 --kernel-trace`): every launch between two consecutive RoIPool launches of the
piece-by-piece phase (fc6 forward pieces have half the full launch's grid), with its queue.

    python tools/exchange_timeline.py <rocprof dir> [pipelined|unpipelined] [min_us]"""
import csv
import glob
import sys


def nm(r):
    return r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:46]


def main():
    d = sys.argv[1]
    mode = sys.argv[2] if len(sys.argv) > 2 else 'pipelined'
    min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 30.0
    f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r['s'], r['e'] = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    rows.sort(key=lambda r: r['s'])
    ex = [r for r in rows if 'exchange_proxy' in r['Kernel_Name']]
    if not ex:
        sys.exit('no exchange_proxy_kernel launches in the trace')
    roi = [r for r in rows if 'roi_pool_nhwc_xcd' in r['Kernel_Name'] and ex[0]['s'] < r['s'] < ex[-1]['e']]
    # steps whose fc6 forward ran as two half-grid pieces = the pipelined projection
    steps = []
    for a, b in zip(roi[:-1], roi[1:]):
        win = [r for r in rows if a['s'] <= r['s'] < b['s']]
        halves = [r for r in win if 'gemm_x3_m16' in r['Kernel_Name'] and r['Grid_Size_X'] == '131072'
                  and r['e'] - r['s'] > 1.0e6]
        exch = [r for r in win if 'exchange_proxy' in r['Kernel_Name']]
        if exch and (len(halves) >= 2) == (mode == 'pipelined') and (b['s'] - a['s']) < 40e6:
            steps.append((a, b, win))
    if not steps:
        sys.exit('no %s step found' % mode)
    a, b, win = steps[len(steps) // 2]
    t0 = a['s']
    print('%s step: %.3f ms between RoIPool launches; %d launches (showing >= %.0f us)' % (
        mode, (b['s'] - t0) / 1e6, len(win), min_us))
    for r in win:
        if (r['e'] - r['s']) / 1e3 >= min_us:
            print('q%-2s %8.3f -> %8.3f  %7.1f us  %-46s grid %s' % (
                r['Queue_Id'], (r['s'] - t0) / 1e6, (r['e'] - t0) / 1e6, (r['e'] - r['s']) / 1e3, nm(r),
                r['Grid_Size_X']))


if __name__ == '__main__':
    main()
