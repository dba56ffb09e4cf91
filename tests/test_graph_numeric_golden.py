"""The oracle's WSDDN outputs / cls_pred / entropy gate / loss seeds against vectors produced by the
IMPORTED reference's own builders executed in numpy (tests/golden/make_golden_graph_numeric.py:
add_webly_outputs, add_cls_pred, add_spatial_entropy_weight, add_webly_losses; 56 ops per case;
the assumed arithmetic of each Caffe2 built-in is listed in reference_graph_numeric.json).

This pins the COMPOSITION numerically: which blobs are
multiplied, transposed, reduced, clipped, in which order, with which constants.  The custom C++
operators stay unpinned: RoIIoU's matrix is supplied by the oracle on both sides, and blobs
downstream of WeightedCrossEntropyWithLogits are stored as `unpinned_*`."""
import json
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden')


def load_cases():
    z = np.load(os.path.join(GOLD, 'reference_graph_numeric.npz'))
    meta = json.load(open(os.path.join(GOLD, 'reference_graph_numeric.json')))
    out = []
    for i, info in enumerate(meta['cases']):
        pre = 'c%d_' % i
        out.append((info, {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}))
    return out, meta


def logits_of(case):
    """The four fc8 logit matrices: stored at R = 64, recomputed (x W^T + b in fp32, as the
    generator did) at R = 300."""
    if 'out_fc8c' in case:
        return [case['out_' + n] for n in ('fc8c', 'fc8d', 'noisy_fc8c', 'noisy_fc8d')]
    x, xn = case['in_drop7'], case['in__[noisy]_drop7']
    return [(src @ case['in_%s_w' % n].T + case['in_%s_b' % n]).astype(np.float32)
            for n, src in (('fc8c', x), ('fc8d', x), ('noisy_fc8c', xn), ('noisy_fc8d', xn))]


def close(a, b, rtol, atol=0.0):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(b)
    assert np.all(np.abs(a[ok] - b[ok]) <= atol + rtol * np.abs(b[ok])), \
        float(np.max(np.abs(a[ok] - b[ok]) / (atol + rtol * np.abs(b[ok]) + 1e-300)))


CASES, META = load_cases()


def test_fixture_is_what_the_docstring_says():
    ops = [t[0] for t in META['executed_ops']]
    assert len(ops) == 56 and ops.count('Softmax') == 4 and ops.count('Stat') == 6
    assert ops.index('RoIIoU') < ops.index('MatMul') < ops.index('LeakyRelu') < ops.index('Clip')
    assert set(ops) - {'Stat', 'StopGradient'} <= set(META['assumptions']) | {
        'Mul', 'Add', 'Sub', 'Div', 'Shape', 'Cast'}
    assert [(c['R'], c['C']) for c, _ in CASES] == [(64, 20), (64, 80), (300, 20), (300, 80), (64, 20)]
    assert [c['nan_case'] for c, _ in CASES] == [False, False, False, False, True]
    for info, case in CASES:
        assert (case['out_rois_pred'][info['p0_rows'], info['p0_class']] == 0).all()     # p = 0 entries
        assert np.isnan(case['out_rois_pred_hatE_sum']).any() == info['nan_case']        # D = 0 column
        assert float(case['out_loss_cls_grad']) == 1.0 and float(case['out_loss_cls_noise_grad']) == 1.0


@pytest.mark.parametrize('idx', range(len(CASES)))
def test_oracle_wsddn_outputs_match_the_reference_graph(idx):
    from oracle import oracle
    info, case = CASES[idx]
    fc8c, fc8d, nc, nd = logits_of(case)
    ac, ad, rp, cp = oracle.wsddn_outputs(fc8c, fc8d)
    acn, adn, rpn, cpn = oracle.wsddn_outputs(fc8c, fc8d, nc, nd)
    if 'out_alpha_cls' in case:
        close(ac, case['out_alpha_cls'], 2e-6); close(ad, case['out_alpha_det'], 2e-6, 1e-44)
        close(acn, case['out_alpha_cls_noise'], 2e-6); close(adn, case['out_alpha_det_noise'], 2e-6, 1e-44)
    close(rp, case['out_rois_pred'], 1e-5, 1e-44)
    close(rpn, case['out_rois_pred_noise'], 1e-5, 1e-44)
    close(cp, case['out_cls_prob'], 1e-5)                       # add_cls_pred: ReduceSum over proposals
    close(cpn, case['out_cls_prob_noise'], 1e-5)
    assert (rp[info['p0_rows'], info['p0_class']] == 0).all()


@pytest.mark.parametrize('idx', range(len(CASES)))
def test_oracle_entropy_gate_matches_the_reference_graph(idx):
    from oracle import oracle
    info, case = CASES[idx]
    rois = case['in_rois']
    if 'out_rois_J' in case:
        assert np.array_equal(oracle.roi_iou(rois), case['out_rois_J'])          # the supplied matrix
    cw, cwn, hs, hsn = oracle.entropy_gate(rois, case['out_rois_pred'], case['out_cls_prob'],
                                           case['in_labels_oh'])
    # J E sums R terms per entry and hatE_sum R more: fp32 summation order is the only freedom
    close(hs, case['out_rois_pred_hatE_sum'], 2e-5)
    close(hsn, case['out_rois_pred_hatE_sum_norm'], 2e-5)
    close(cwn, case['out_rois_class_weight_noise'], 2e-5)
    close(cw, case['out_rois_class_weight'], 2e-5)
    lab = case['in_labels_oh']
    fg = lab[0] == 1
    assert (np.asarray(cwn)[0, fg] == 0).all() and (np.asarray(cw)[0, fg] == 1).all()


@pytest.mark.parametrize('idx', range(len(CASES)))
def test_oracle_loss_tail_composes_as_the_reference_graph(idx):
    """loss_tail = outputs -> gate -> WCE -> AveragedLoss with seed 1.0; the WCE value itself is the
    oracle's on both sides (unpinned), so this checks wiring: which probability meets which weight."""
    from oracle import oracle
    info, case = CASES[idx]
    fc8c, fc8d, nc, nd = logits_of(case)
    t = oracle.loss_tail(dict(fc8c=fc8c, fc8d=fc8d, noisy_fc8c=nc, noisy_fc8d=nd), case['in_rois'],
                         case['in_labels_oh'])
    close(np.reshape(t['loss_cls'], ()), case['unpinned_loss_cls'], 2e-5)
    close(np.reshape(t['loss_cls_noise'], ()), case['unpinned_loss_cls_noise'], 2e-5)
    close(t['class_weight'], case['out_rois_class_weight'], 2e-5)
