#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel-trace stats + separate FETCH_SIZE / WRITE_SIZE PMC
passes) into the small summaries kept under profiles/.

  summarize_profile.py <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <out_prefix> [mfma_dtype]

mfma_dtype (fp16x2 default | fp32x3 | fp32 | bf16) selects which kernel is bench.py's dominant one.

HBM traffic per launch follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so
read bytes = 2 * FETCH_SIZE * 1024 (upper estimate for narrow accesses).

Kernels are matched on a TEMPLATE PREFIX (name + leading template arguments, followed by `,` or
`>`): round 4 appended a ninth template argument to gemm_x3_m16_kernel and the exact-string match
of the time silently stopped firing (empty cluster table, stale traffic file).  `summarize()`
now raises when the dominant kernel is absent from the trace; tests/test_profile_tools.py runs it
over a committed excerpt."""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

# mode -> (template prefix, label, algorithmic GB per launch (operand planes of x [4000, 25088] and
# of fc6_w [8192, 25088] read + h6 [4000, 8192] fp32 written: 401 + 822 + 131 MB at 4 B of planes per
# element, 201 + 411 + 131 at the bf16 plan's 2 B), Grid_Size of the fc6-fwd launch:
# other launches of the same template - fc7, fc6 wgrad - have another grid or another duration)
DOMINANT = {
    'fp32': ('gemm_f32_kernel<256, 256, 16, true, true, false, 4, 4',
             'gemm_f32_kernel<256,256,16,KC,KC,4x4> fc6 fwd', 1.354, None),
    'fp16x2': ('gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, true',
               'gemm_x3_m16_kernel<256,256,4x2,2 stages,2 planes x 2 slabs,f16> fc6 fwd', 1.354, '262144'),
    'fp32x3': ('gemm_x3_m16_kernel<256, 128, 4, 2, 2, 3, 2, false',
               'gemm_x3_m16_kernel<256,128,4x2,2 stages,3 planes x 2 slabs> fc6 fwd', 1.966, '524288'),
    'bf16': ('gemm_x3_m16_kernel<256, 256, 4, 2, 2, 1, 4, false',
             'gemm_x3_m16_kernel<256,256,4x2,2 stages,1 plane x 4 slabs,bf16> fc6 fwd', 0.743, '262144'),
}

# Other kernels of the default plan whose measured-to-algorithmic traffic ratio goes into the
# bench line.  (template prefix, Grid_Size or None, label, algorithmic MB per
# launch at BASELINE configs[1]: 2 images 600 x 1000, R = 2000 each, C = 20).
#   roi_pool: 2 x 2000 x 25088 x 4 B of operand planes written + 2 x 18.8 MB of conv5_3 read (half of
#     it per image: since round 5 the engine pools each image on its own conv stream)
#   fc6 wgrad + SGD: dZ6^T planes 131 MB + x planes 401 MB read; w, momentum read and written
#     (4 x 822 MB) + the 4-byte planes (822 MB)
#   conv_h2_wp (one image per launch): fp32 NHWC input + (pooled) output; a grid is shared by the
#     layers of one resolution, the figure is their mean: conv1_2 153.6 + 38.4; conv2_1 38.4 +
#     76.8, conv2_2 76.8 + 19.2; conv3_1 19.2 + 38.4, conv3_2 38.4 + 38.4, conv3_3 38.4 + 9.6
EXTRA = [
    ('roi_pool_nhwc_xcd_kernel<true, true', '4096000', 'roi_pool_nhwc_xcd (RoIPoolF + boost -> fc6 planes), all 4000 proposals in one launch', 439.0),
    ('roi_pool_nhwc_xcd_kernel<true, true', '2048000', 'roi_pool_nhwc_xcd, one image (2000 proposals) per launch on its conv stream', 219.5),
    ('gemm_h2_btr_kernel<256, 256, 4, 2, true', None, 'gemm_h2_btr<256,256,SGD> (fc6 wgrad + update)', 4642.0),
    ('gemm_h2_btr_kernel<256, 256, 4, 2, false', None, 'gemm_h2_btr<256,256> (fc6 wgrad, gradient written)', 1354.0),
    ('conv_h2_wp_kernel<2, 2, 1, true, 2', '614400', 'conv_h2_wp conv1_2 + pool1 (one image)', 192.0),
    ('conv_h2_wp_kernel<2, 2, 1, true, 2', '311296', 'conv_h2_wp conv2_1 / conv2_2 + pool2 (mean)', 105.6),
    ('conv_h2_wp_kernel<2, 2, 1, true, 2', '155648', 'conv_h2_wp conv3_1 / 3_2 / 3_3 + pool3 (mean)', 60.8),
    # Winograd F(4x4,3x3), one image per launch (round 6): input transform = fp32 NHWC input read + the
    # two f16 V planes written (36 x tiles x Cin x 4 B); batch GEMM = V + U planes read + M written;
    # output transform = M read + NHWC output written.  608 tiles at 75 x 125, 640 at 74 x 124 dilated.
    ('wino4_input_h2_kernel', '233472', 'wino4_input conv4_2 conv4_3 (19.2 MB in, 44.8 MB of V planes out)', 64.0),
    ('wino4_input_h2_kernel', '245760', 'wino4_input conv5_x (18.8 MB in, 47.2 MB out)', 66.0),
    ('gemm_x3_kernel<128, 128, 2, 2, 3, 2, 1, true', '184320',
     'gemm_x3 winograd4 batch GEMM 36 frequencies (gemm_x3_kernel<128,128,3 stages>; mean of conv4_1 86.1, conv4_2/3 127.3, conv5_x 132.1)', 122.8),
    ('wino4_output_kernel', '77824', 'wino4_output conv4_x (44.8 MB of M in, 19.2 MB out)', 64.0),
    ('wino4_output_kernel', '81920', 'wino4_output conv5_x (47.2 MB in, 18.8 MB out)', 66.0),
]


def short(name):
    name = name.replace('(anonymous namespace)::', '').replace('void ', '')
    return name.split('(')[0][:70]


def template_match(prefix, kernel_name):
    """True when `kernel_name` is an instance of the template prefix: the prefix must be followed
    by `,` (more template arguments), `>` (none), `(` (the kernel is not a template) or `<` (the prefix
    is a bare template name)."""
    return re.search(re.escape(prefix) + r'\s*[,>(<]', kernel_name) is not None


def _find(d, pat):
    return glob.glob(os.path.join(d, pat)) + glob.glob(os.path.join(d, '*', pat))


def _counter_rows(d, cname):
    f = _find(d, '*_counter_collection.csv')
    if not f:
        return []
    return [r for r in csv.DictReader(open(f[0])) if r['Counter_Name'] == cname]


def _avg(v):
    return sum(v) / len(v) if v else None


def cluster_durations(v, ratio=1.8):
    """Sorted durations of one (template, grid) split where two neighbours differ by more than
    `ratio`: launches of one template with one grid can still be different problems (fc6 forward
    K = 25088 and the batch-2 fc7 GEMMs K = 4096 share grid 262144 x 1; the trace has no K column)."""
    v = sorted(v)
    out, cur = [], [v[0]]
    for a, b in zip(v, v[1:]):
        if b > ratio * a:
            out.append(cur)
            cur = []
        cur.append(b)
    out.append(cur)
    return out


def summarize(stats_dir, fetch_dir, write_dir, out, mode='fp16x2', require_dominant=True):
    """Writes <out>.md, <out>_kernel_stats.csv and (when both PMC passes hold the dominant kernel)
    <out>_traffic.json; returns {'dominant': {...}, 'traffic': {...} | None, 'extra': [...]}."""
    dom_pre, dom_name, alg_gb, dom_grid = DOMINANT[mode]
    rows = list(csv.DictReader(open(_find(stats_dir, '*_kernel_stats.csv')[0])))
    lines = ['| kernel | calls | total ms | avg us | % |', '|---|---|---|---|---|']
    for r in rows[:30]:
        lines.append('| %s | %s | %.3f | %.1f | %.2f |' % (
            short(r['Name']), r['Calls'], float(r['TotalDurationNs']) / 1e6,
            float(r['AverageNs']) / 1e3, float(r['Percentage'])))
    fetch_rows, write_rows = _counter_rows(fetch_dir, 'FETCH_SIZE'), _counter_rows(write_dir, 'WRITE_SIZE')
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for rws, cname in ((fetch_rows, 'FETCH_SIZE'), (write_rows, 'WRITE_SIZE')):
        for r in rws:
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
    result = {'dominant': None, 'traffic': None, 'extra': []}
    # the dominant kernel of bench.py: fc6 forward = the K-contiguous x K-contiguous GEMM with the
    # largest grid (M=4000 N=8192 -> 2048 workgroups)
    dom = {}
    for rws, cname in ((fetch_rows, 'FETCH_SIZE'), (write_rows, 'WRITE_SIZE')):
        sel = [float(r['Counter_Value']) for r in rws if template_match(dom_pre, r['Kernel_Name'])
               and (dom_grid is None or r['Grid_Size'] == dom_grid)]
        # fc7 fwd (batch 2) has the same thread count; fc6 launches are the ones that move the
        # most bytes
        sel = [v for v in sel if v >= 0.5 * max(sel)] if sel else []
        dom[cname] = _avg(sel)
    if dom.get('FETCH_SIZE') is not None and dom.get('WRITE_SIZE') is not None:
        tb = (2 * dom['FETCH_SIZE'] + dom['WRITE_SIZE']) * 1024
        result['traffic'] = {
            'kernel': dom_name, 'mfma_dtype': mode, 'algorithmic_bytes_per_launch': alg_gb * 1e9,
            'FETCH_SIZE_KiB_per_launch': dom['FETCH_SIZE'],
            'WRITE_SIZE_KiB_per_launch': dom['WRITE_SIZE'],
            'hbm_bytes_per_launch': tb,
            'ratio_vs_algorithmic': round(tb / (alg_gb * 1e9), 3),
            'note': 'separate --pmc passes of `bench.py --steps 2 --warmup 1`; read bytes = '
                    '2*FETCH_SIZE*1024 (gfx950 FETCH_SIZE counts 64 B per 128-B request; '
                    'Infinity-Cache hits are included in the memory-side counters)'}
        lines += ['', 'fc6 fwd GEMM: FETCH %.0f KiB, WRITE %.0f KiB per launch -> %.2f GB fabric '
                  'traffic per launch (algorithmic %.3f GB)' % (dom['FETCH_SIZE'], dom['WRITE_SIZE'],
                                                               tb / 1e9, alg_gb)]
    # measured / algorithmic traffic of the other hot kernels of the default plan
    if mode == 'fp16x2' and (fetch_rows or write_rows):
        lines += ['', '| kernel (grid) | launches | HBM traffic MB/launch | algorithmic MB/launch | ratio |',
                  '|---|---|---|---|---|']
        for pre, grid, label, alg_mb in EXTRA:
            f = [float(r['Counter_Value']) for r in fetch_rows if template_match(pre, r['Kernel_Name'])
                 and (grid is None or r['Grid_Size'] == grid)]
            w = [float(r['Counter_Value']) for r in write_rows if template_match(pre, r['Kernel_Name'])
                 and (grid is None or r['Grid_Size'] == grid)]
            if not f or not w:
                continue
            mb = (2 * _avg(f) + _avg(w)) * 1024 / 1e6
            result['extra'].append({'kernel': label, 'grid': grid, 'launches': len(f),
                                    'hbm_mb_per_launch': round(mb, 1), 'algorithmic_mb_per_launch': alg_mb,
                                    'ratio': round(mb / alg_mb, 3)})
            lines.append('| %s (%s) | %d | %.1f | %.1f | %.2f |' % (label, grid or 'any', len(f), mb,
                                                                    alg_mb, mb / alg_mb))
        if result['traffic'] is not None:
            result['traffic']['other_kernels'] = result['extra']
    if result['traffic'] is not None:
        json.dump(result['traffic'], open(out + '_traffic.json', 'w'), indent=1)
    # per-shape durations of the dominant kernel's template from the kernel trace (the stats table
    # above averages every launch of the template: fc6 fwd, fc6 wgrad and the three fc7 GEMMs)
    tr = _find(stats_dir, '*_kernel_trace.csv')
    if tr:
        by = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if template_match(dom_pre, r['Kernel_Name']):
                key = (r['Grid_Size_X'], r['Grid_Size_Z'])
                by[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        if not by and require_dominant:
            raise RuntimeError('summarize_profile: no launch of the dominant kernel `%s` in %s - '
                               'the template changed? (update DOMINANT)' % (dom_pre, tr[0]))
        lines += ['', '| %s launches by grid (x, z) and duration cluster | launches | avg ms | min ms | max ms |'
                  % short(dom_pre), '|---|---|---|---|---|']
        clusters = [(key, c) for key, v in by.items() for c in cluster_durations(v)]
        for key, v in sorted(clusters, key=lambda kv: -sum(kv[1]) / len(kv[1])):
            lines.append('| grid %s x %s | %d | %.3f | %.3f | %.3f |' % (key[0], key[1], len(v),
                                                                       sum(v) / len(v), min(v), max(v)))
        dom_c = [v for key, v in clusters if dom_grid is None or key[0] == dom_grid]
        if dom_c:
            v = max(dom_c, key=lambda c: sum(c) / len(c))
            result['dominant'] = {'grid': dom_grid, 'launches': len(v), 'avg_ms': sum(v) / len(v),
                                  'min_ms': min(v), 'max_ms': max(v)}
            lines += ['', 'dominant launch (fc6 forward, M=4000 N=8192 K=25088): grid %s, %d launches, '
                      'avg %.3f ms (min %.3f, max %.3f) - the launch `bench.py` times live with HIP '
                      'events (`roofline.kernel_ms`)' % (dom_grid, len(v), sum(v) / len(v), min(v), max(v))]
    open(out + '.md', 'w').write('\n'.join(lines) + '\n')
    shutil.copy(_find(stats_dir, '*_kernel_stats.csv')[0], out + '_kernel_stats.csv')
    result['text'] = '\n'.join(lines)
    return result


def main():
    stats_dir, fetch_dir, write_dir, out = sys.argv[1:5]
    mode = sys.argv[5] if len(sys.argv) > 5 else 'fp16x2'
    # (the infer summary reuses the fp16x2 tables; its fc6 launches have other grids)
    print(summarize(stats_dir, fetch_dir, write_dir, out, mode,
                    require_dominant='--allow-missing' not in sys.argv)['text'])


if __name__ == '__main__':
    main()
