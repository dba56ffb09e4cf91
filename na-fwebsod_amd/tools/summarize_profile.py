#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC
passes) into the small summaries kept under profiles/.

  summarize_profile.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix> [mfma_dtype]

mfma_dtype (fp16x2 default | fp32x3 | fp32 | bf16) selects which kernel is bench.py's dominant one.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so
read bytes = 2 * FETCH_SIZE * 1024 (upper estimate for narrow accesses)."""
import collections
import csv
import glob
import os
import sys


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:70]


def _find(d, pat):
    return glob.glob(os.path.join(d, pat)) + glob.glob(os.path.join(d, '*', pat))


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    mode = sys.argv[5] if len(sys.argv) > 5 else 'fp16x2'
    # (kernel-name substring, label, algorithmic GB per launch, Grid_Size of the fc6-fwd launch:
    #  other launches of the same template - fc6 wgrad - have another grid)
    dom_sub, dom_name, alg_gb, dom_grid = {
        'fp32': ('gemm_f32_kernel<256, 256, 16, true, true, false, 4, 4>',
                 'gemm_f32_kernel<256,256,16,KC,KC,4x4> fc6 fwd', 1.354, None),
        'fp16x2': ('gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, true>',
                   'gemm_x3_m16_kernel<256,256,4x2,2 stages,2 planes x 2 slabs,f16> fc6 fwd', 1.354, '262144'),
        'fp32x3': ('gemm_x3_m16_kernel<256, 128, 4, 2, 2, 3, 2, false>',
                   'gemm_x3_m16_kernel<256,128,4x2,2 stages,3 planes x 2 slabs> fc6 fwd', 1.966, '524288'),
        'bf16': ('gemm_x3_m16_kernel<256, 256, 4, 2, 2, 1, 4, false>',
                 'gemm_x3_m16_kernel<256,256,4x2,2 stages,1 plane x 4 slabs,bf16> fc6 fwd', 0.677, '262144'),
    }[mode]
    rows = list(csv.DictReader(open(_find(stats_dir, '*_kernel_stats.csv')[0])))
    lines = ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in rows[:30]:
        lines.append('| %s | %s | %.3f | %.1f | %.2f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for d, cname in ((fetch_dir, 'FETCH_SIZE'), (write_dir, 'WRITE_SIZE')):
        f = _find(d, '*_counter_collection.csv')
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            if r['Counter_Name'] == cname:
                pmc[short(r['Kernel_Name'])][cname].append(float(r['Counter_Value']))
    lines += ['', '| kernel | launches | FETCH_SIZE KiB/launch | WRITE_SIZE KiB/launch | '
              'HBM traffic MB/launch (2*FETCH+WRITE) |', '|---|---|---|---|---|']
    for k, v in sorted(pmc.items(), key=lambda kv: -sum(kv[1].get('FETCH_SIZE', [0]))):
        f = v.get('FETCH_SIZE', [])
        w = v.get('WRITE_SIZE', [])
        fa = sum(f) / len(f) if f else 0.0
        wa = sum(w) / len(w) if w else 0.0
        lines.append('| %s | %d | %.0f | %.0f | %.1f |' % (k, max(len(f), len(w)), fa, wa,
                                                          (2 * fa + wa) * 1024 / 1e6))
    # the dominant kernel of bench.py: fc6 forward = the K-contiguous x K-contiguous GEMM with the
    # largest grid (M=4000 N=8192 -> 2048 workgroups)
    dom = {}
    for d, cname in ((fetch_dir, 'FETCH_SIZE'), (write_dir, 'WRITE_SIZE')):
        f = _find(d, '*_counter_collection.csv')
        vals = []
        if f:
            rws = [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == cname and
                   dom_sub in r['Kernel_Name'] and (dom_grid is None or r['Grid_Size'] == dom_grid)]
            if rws:
                # fc7 fwd (batch 2) has the same thread count; fc6 launches are the ones that
                # move the most bytes
                allv = [float(r['Counter_Value']) for r in rws]
                vals = [v for v in allv if v >= 0.5 * max(allv)]
        dom[cname] = sum(vals) / len(vals) if vals else None
    if dom.get('FETCH_SIZE') is not None and dom.get('WRITE_SIZE') is not None:
        import json
        tb = (2 * dom['FETCH_SIZE'] + dom['WRITE_SIZE']) * 1024
        json.dump({'kernel': dom_name, 'mfma_dtype': mode, 'algorithmic_bytes_per_launch': alg_gb * 1e9,
                   'FETCH_SIZE_KiB_per_launch': dom['FETCH_SIZE'],
                   'WRITE_SIZE_KiB_per_launch': dom['WRITE_SIZE'],
                   'hbm_bytes_per_launch': tb,
                   'note': 'separate --pmc passes of `bench.py --steps 2 --warmup 1`; read bytes = '
                           '2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request; '
                           'Infinity-Cache hits are included in the memory-side counters)'},
                  open(out + '_traffic.json', 'w'), indent=1)
        lines += ['', 'fc6 fwd GEMM: FETCH %.0f KiB, WRITE %.0f KiB per launch -> %.2f GB fabric '
                  'traffic per launch (algorithmic %.3f GB)' % (dom['FETCH_SIZE'], dom['WRITE_SIZE'],
                                                               tb / 1e9, alg_gb)]
    # per-shape durations of the dominant kernel's template from the kernel trace (the stats table
    # above averages every launch of the template: fc6 fwd, fc6 wgrad and the three fc7 GEMMs)
    tr = _find(stats_dir, '*_kernel_trace.csv')
    if tr:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if dom_sub in r['Kernel_Name']:
                key = (r['Grid_Size_X'], r['Grid_Size_Z'])
                by[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        lines += ['', '| %s launches by grid (x, z) and duration cluster | launches | avg ms | min ms | max ms |'
                  % short(dom_sub), '|---|---|---|---|---|']
        # launches of one template with one grid can still be different problems (fc6 forward
        # K = 25088 and the batch-2 fc7 GEMMs K = 4096 share grid 262144 x 1): the trace has no K
        # column, so a grid's launches are split where two neighbouring sorted durations differ by
        # more than 1.8x - each cluster is one problem shape
        clusters = []
        for key, v in by.items():
            v = sorted(v)
            cur = [v[0]]
            for a, b in zip(v, v[1:]):
                if b > 1.8 * a:
                    clusters.append((key, cur))
                    cur = []
                cur.append(b)
            clusters.append((key, cur))
        for key, v in sorted(clusters, key=lambda kv: -sum(kv[1]) / len(kv[1])):
            lines.append('| grid %s x %s | %d | %.3f | %.3f | %.3f |' % (key[0], key[1], len(v),
                                                                       sum(v) / len(v), min(v), max(v)))
        if dom_grid is not None:
            dom_c = [v for key, v in clusters if key[0] == dom_grid]
            if dom_c:
                v = max(dom_c, key=lambda c: sum(c) / len(c))
                lines += ['', 'dominant launch (fc6 forward, M=4000 N=8192 K=25088): grid %s, %d launches, '
                          'avg %.3f ms (min %.3f, max %.3f) - the launch `bench.py` times live with HIP '
                          'events (`roofline.kernel_ms`)' % (dom_grid, len(v), sum(v) / len(v), min(v), max(v))]
    open(out + '.md', 'w').write('\n'.join(lines) + '\n')
    import shutil
    shutil.copy(_find(stats_dir, '*_kernel_stats.csv')[0], out + '_kernel_stats.csv')
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
