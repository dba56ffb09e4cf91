"""tools/summarize_profile.py over a committed excerpt of a real rocprofv3 run (tests/golden/
profile_excerpt: rows of the hot kernels cut from round 4's `--kernel-trace --stats` and the two
`--pmc` passes of bench.py).  Round 4's summaries went stale because the dominant kernel gained a
template argument and an exact-string match stopped firing without anybody noticing: this test
fails when the fc6-forward row, the traffic file or the per-kernel traffic ratios are missing."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, 'tests', 'golden', 'profile_excerpt')


def _tool():
    spec = importlib.util.spec_from_file_location(
        'summarize_profile', os.path.join(ROOT, 'na-fwebsod_amd', 'tools', 'summarize_profile.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_template_prefix_match():
    sp = _tool()
    pre = sp.DOMINANT['fp16x2'][0]
    old = 'void (anonymous namespace)::gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, true>(XArgs)'
    new = 'void (anonymous namespace)::gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, true, false>(XArgs)'
    other = 'void (anonymous namespace)::gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 2, false, false>(XArgs)'
    longer = 'void (anonymous namespace)::gemm_x3_m16_kernel<256, 256, 4, 2, 2, 2, 22, true>(XArgs)'
    assert sp.template_match(pre, old) and sp.template_match(pre, new)
    assert not sp.template_match(pre, other) and not sp.template_match(pre, longer)


def test_summary_has_the_fc6_row_and_the_traffic_file(tmp_path):
    sp = _tool()
    out = str(tmp_path / 'rXX_bench_fp16x2')
    res = sp.summarize(os.path.join(EX, 'stats'), os.path.join(EX, 'fetch'), os.path.join(EX, 'write'),
                       out, 'fp16x2')
    dom = res['dominant']
    assert dom is not None and dom['grid'] == '262144' and dom['launches'] >= 3
    assert 2.5 < dom['avg_ms'] < 5.0           # fc6 forward, not the 0.6 ms fc7 launches of the same grid
    text = open(out + '.md').read()
    assert 'dominant launch (fc6 forward' in text and '| grid 262144 x 1 |' in text
    tj = json.load(open(out + '_traffic.json'))
    assert tj['mfma_dtype'] == 'fp16x2'
    assert 3e9 < tj['hbm_bytes_per_launch'] < 8e9
    assert tj['ratio_vs_algorithmic'] == pytest.approx(tj['hbm_bytes_per_launch'] / 1.354e9, rel=1e-3)
    labels = [e['kernel'] for e in tj['other_kernels']]
    for want in ('roi_pool_nhwc_xcd', 'gemm_h2_btr<256,256,SGD>', 'conv_h2_wp conv1_2'):      # (r04 excerpt: one RoIPool launch per step)
        assert any(want in l for l in labels), (want, labels)
    assert all(e['ratio'] > 0.9 for e in tj['other_kernels'])
    assert os.path.exists(out + '_kernel_stats.csv')


def test_missing_dominant_kernel_is_an_error(tmp_path):
    """A trace without the dominant kernel (a renamed template) must not produce an empty table."""
    sp = _tool()
    d = tmp_path / 'stats'
    d.mkdir()
    for f in ('st_kernel_stats.csv', 'st_kernel_trace.csv'):
        s = open(os.path.join(EX, 'stats', f)).read().replace('gemm_x3_m16_kernel', 'gemm_renamed_kernel')
        (d / f).write_text(s)
    with pytest.raises(RuntimeError):
        sp.summarize(str(d), os.path.join(EX, 'fetch'), os.path.join(EX, 'write'),
                     str(tmp_path / 'o'), 'fp16x2')


def test_committed_traffic_file_is_the_one_bench_reads():
    """bench.py copies roofline.traffic from the newest profiles/r*_bench*traffic*.json of its plan:
    that file must carry the per-kernel ratios the bench line republishes."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_fp16x2_traffic.json')))
    assert files
    tj = json.load(open(files[-1]))
    assert tj['mfma_dtype'] == 'fp16x2' and tj['hbm_bytes_per_launch'] > 0
    assert tj.get('other_kernels'), 'regenerate with tools/summarize_profile.py'
