#!/usr/bin/env python3
"""Per-launch table of rocprofv3 --pmc passes (directories given on the command line, each
holding *_counter_collection.csv + *_kernel_trace.csv): counters joined per (kernel, grid) in
launch order, averaged over launches.   python tools/read_pmc.py DIR [DIR ...] [--match conv]"""
import collections
import csv
import glob
import sys


def main():
    dirs = [a for a in sys.argv[1:] if not a.startswith('--')]
    match = ''
    if '--match' in sys.argv:
        match = sys.argv[sys.argv.index('--match') + 1]
        dirs.remove(match)
    table = collections.OrderedDict()
    for d in dirs:
        cc = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
        kt = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
        dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
               for r in csv.DictReader(open(kt))}
        per = collections.OrderedDict()
        for r in csv.DictReader(open(cc)):
            k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            if match and match not in k:
                continue
            e = per.setdefault(r['Dispatch_Id'], {'_k': k, '_g': r['Grid_Size'], '_lds': r.get('LDS_Block_Size', '')})
            e[r['Counter_Name']] = e.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        seq = collections.defaultdict(int)
        for did, e in per.items():
            key0 = (e['_k'], e['_g'])
            key = key0 + (seq[key0] % 1000,)
            seq[key0] += 1
            key = key0
            t = table.setdefault(key, collections.defaultdict(list))
            t['us'].append(dur[did])
            for c, v in e.items():
                if not c.startswith('_'):
                    t[c].append(v)
    for key, t in table.items():
        avg = {c: sum(v) / len(v) for c, v in t.items()}
        print('%s grid %s  launches %d  %.1f us' % (key[0], key[1], len(t['us']), avg['us']))
        cyc = avg.get('GRBM_GUI_ACTIVE', 0) / 8
        if cyc:
            print('    clock %.2f GHz  MFMA busy %.1f %%' % (
                cyc / avg['us'] / 1e3, 100 * avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024)))
        wc = avg.get('SQ_WAVE_CYCLES')
        for c, v in sorted(avg.items()):
            if c == 'us':
                continue
            extra = ''
            if wc and c.startswith('SQ_') and c != 'SQ_WAVE_CYCLES':
                extra = '  (%.1f %% of wave cycles)' % (100 * v / wc)
            print('    %-28s %.4g%s' % (c, v, extra))


if __name__ == '__main__':
    main()
