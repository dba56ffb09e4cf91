"""Known-answer tests that pin the CPU oracle (oracle/README.md): hand-derived values, the one
recorded reference output, and fixtures generated from the imported reference Python."""
import numpy as np
import pytest

from oracle import oracle


def test_wce_recorded_reference_output():
    # reference cross_entropy_wsl_op.cc compiled by the survey -> 0.206650317
    x = np.array([[.9, .05, .3, 0.]], np.float32)
    l = np.array([[1, 0, .4, 0]], np.float32)
    w = np.array([[1, .5, 1, .25]], np.float32)
    assert abs(oracle.weighted_ce(x, l, w, True) - 0.206650317) < 5e-8
    # hand calculation in float64
    p, q = np.maximum(x, 1e-20).astype(np.float64), np.maximum(1 - x, 1e-20).astype(np.float64)
    hand = -((l * np.log(p) + (1 - l) * np.log(q)) * w).sum() / 4
    assert abs(oracle.weighted_ce(x, l, w, True) - hand) < 1e-7
    assert abs(oracle.weighted_ce(x, l, None, False) -
               (-(l * np.log(p) + (1 - l) * np.log(q)).sum())) < 1e-6


def test_wce_grad_clamp_and_weight_order():
    x = np.array([[1e-30, 0.5]], np.float32)
    l = np.array([[0.0, 1.0]], np.float32)
    w = np.array([[3.0, 2.0]], np.float32)
    g = oracle.weighted_ce_grad(x, l, w, np.float32(1.0), False)
    np.testing.assert_allclose(g, [[1.0 * 3.0, -2.0 * 2.0]], rtol=1e-6)
    xl = np.array([[1.0]], np.float32)          # 1/(1-x) -> 1e20, clamped to 1e4 BEFORE the weight
    g = oracle.weighted_ce_grad(xl, np.array([[0.0]], np.float32), np.array([[5.0]], np.float32),
                                np.float32(1.0), False)
    assert g[0, 0] == np.float32(5e4)
    g = oracle.weighted_ce_grad(x, l, w, np.float32(2.0), True)      # is_mean -> /C
    np.testing.assert_allclose(g, [[3.0, -4.0]], rtol=1e-6)


def test_roi_iou_known():
    rois = np.array([[0, 0, 0, 9, 9], [0, 5, 5, 14, 14], [0, 148.5, 3.2, 200.7, 90.9],
                     [0, 148, 3, 200, 90], [0, 50, 50, 40, 40]], np.float32)
    j = oracle.roi_iou(rois)
    assert j[0, 1] == np.float32(25.0) / np.float32(175.0) and j[1, 0] == j[0, 1]
    assert j[2, 3] == 1.0           # 148.5 -> 148 etc.: identical after int truncation
    assert np.all(np.diag(j) == 1.0)
    assert j[0, 2] == 0.0
    # degenerate box (x2<x1): w = max(40-50+1,0) = 0 -> inters 0
    assert j[4, 0] == 0.0


def test_roi_pool_known():
    h, w = 6, 8
    x = np.arange(h * w, dtype=np.float32).reshape(1, 1, h, w)
    # roi (x1,y1,x2,y2) = (8,8,39,31) * 0.125 -> start (1,1), end (round(4.875)=5, round(3.875)=4)
    rois = np.array([[0, 8, 8, 39, 31]], np.float32)
    y, am = oracle.roi_pool_f(x, rois, 2, 2, 0.125)
    # roi_h = 4, roi_w = 5; bins h: [1,3),[3,5); w: [floor(0),ceil(2.5))+1=[1,4), [3,6)
    assert y.reshape(-1).tolist() == [2 * 8 + 3, 2 * 8 + 5, 4 * 8 + 3, 4 * 8 + 5]
    assert am.reshape(-1).tolist() == [19, 21, 35, 37]
    # half-away-from-zero: 12*0.125 = 1.5 -> 2 (np.round would give 2 too), 20*.125=2.5 -> 3 (not 2)
    rois = np.array([[0, 20, 12, 20, 12]], np.float32)
    y, am = oracle.roi_pool_f(x, rois, 1, 1, 0.125)
    assert am.reshape(-1).tolist() == [2 * 8 + 3]
    # malformed roi (x2<x1) is forced to 1x1 at its start; outside roi -> empty bins -> 0, -1
    rois = np.array([[0, 24, 16, 8, 8], [0, 800, 800, 900, 900]], np.float32)
    y, am = oracle.roi_pool_f(x, rois, 2, 2, 0.125)
    assert am[0].reshape(-1).tolist() == [2 * 8 + 3] * 4
    assert np.all(y[1] == 0) and np.all(am[1] == -1)
    # ties: first maximum in raster order wins; negative features keep their max (not 0)
    xt = -np.ones((1, 1, 4, 4), np.float32)
    y, am = oracle.roi_pool_f(xt, np.array([[0, 0, 0, 24, 24]], np.float32), 1, 1, 0.125)
    assert y.reshape(-1).tolist() == [-1.0] and am.reshape(-1).tolist() == [0]


def test_feature_boost_and_stat():
    x = np.arange(12, dtype=np.float32).reshape(3, 4)
    s = np.array([[1.5], [2.0], [1.0]], np.float32)
    np.testing.assert_array_equal(oracle.roi_feature_boost(x, s), x * s)
    ai = np.full(3, 9.0, np.float32); al = np.full(3, 9.0, np.float32)
    oracle.stat(np.array([1, 2, 3], np.float32), np.array([1, 0, 1], np.float32), ai, al, True)
    oracle.stat(np.array([1, 1, 1], np.float32), np.array([1, 1, 0], np.float32), ai, al, False)
    assert ai.tolist() == [2, 1, 3] and al.tolist() == [2, 1, 1]


def test_acm_sgd_known():
    g = np.array([1.0, -2.0], np.float32)
    m = np.array([5.0, 5.0], np.float32)      # discarded: first call zeroes momentum
    a = np.array([7.0, 7.0], np.float32)
    p = np.array([10.0, 10.0], np.float32)
    it = oracle.acm_sgd(g, m, np.float32(0.1), p, a, 0.9, 0, 0.5, 1, 2, 2.0, 0)
    # acm = g/2 + 0.5*p = [5.5, 4.0]; m = 0.2*acm; p -= m
    np.testing.assert_allclose(m, [1.1, 0.8], rtol=1e-6)
    np.testing.assert_allclose(p, [8.9, 9.2], rtol=1e-6)
    assert it == 1 and np.all(a == 0) and g.tolist() == [1.0, -2.0]
    it = oracle.acm_sgd(g, m, np.float32(0.1), p, a, 0.9, 0, 0.0, 2, 1, 1.0, it)   # iter_size 2:
    assert it == 2                                  # (1+1)%2==0 -> update happens now
    m2 = m.copy(); p2 = p.copy()
    it = oracle.acm_sgd(g, m2, np.float32(0.1), p2, a, 0.9, 0, 0.0, 2, 1, 1.0, it)  # accumulate only
    assert it == 3 and a.tolist() == [1.0, -2.0] and np.array_equal(p2, p)


def test_wsddn_outputs_sums():
    rng = np.random.default_rng(0)
    a, b, c, d = [rng.standard_normal((50, 20)).astype(np.float32) for _ in range(4)]
    ac, ad, rp, cp = oracle.wsddn_outputs(a, b, c, d)
    np.testing.assert_allclose(ac.sum(1), 1, rtol=1e-5)
    np.testing.assert_allclose(ad.sum(0), 1, rtol=1e-5)
    np.testing.assert_allclose(cp[0], (ac * ad).sum(0), rtol=1e-5)
    e = np.exp((a + c) - (a + c).max(1, keepdims=True))
    np.testing.assert_allclose(ac, e / e.sum(1, keepdims=True), rtol=1e-5)
    # backward against a float64 finite-difference-free analytic form
    g = rng.standard_normal(20).astype(np.float32)
    dzc, dzd = oracle.wsddn_outputs_grad(ac, ad, g)
    da = g * ad
    np.testing.assert_allclose(dzc, ac * (da - (da * ac).sum(1, keepdims=True)), rtol=1e-4, atol=1e-7)
    dd = g * ac
    np.testing.assert_allclose(dzd, ad * (dd - (dd * ad).sum(0, keepdims=True)), rtol=1e-4, atol=1e-7)


def test_entropy_gate_known():
    # two identical boxes (J = all ones), C = 2: D[r] = E[0]+E[1], hatE = E^2/D
    rois = np.array([[0, 0, 0, 9, 9], [0, 0, 0, 9, 9]], np.float32)
    p = np.array([[0.5, 0.0], [0.25, 0.0]], np.float32)
    y = p.sum(0, keepdims=True)
    lab = np.array([[1.0, 0.0]], np.float32)
    cw, cwn, hs, hsn = oracle.entropy_gate(rois, p, y, lab)
    e = -(p[:, 0] * np.log(p[:, 0]))
    s = (e * e / e.sum()).sum()
    assert abs(hs[0, 0] - s) < 1e-6
    norm = (np.log(2.0) - np.log(0.75)) * 0.75
    assert abs(hsn[0, 0] - min(max(s / norm, 0), 1)) < 1e-6
    assert cwn[0, 0] == 0.0 and cw[0, 0] == 1.0            # labelled class: bg factor 0
    # class 1: p = 0 everywhere -> E = 0 (ReplaceNaN), D = 0 -> 0/0 = NaN survives the Clip
    assert np.isnan(hs[0, 1]) and np.isnan(cw[0, 1])


def test_golden_python_reference_pairs():
    """_project_im_rois / _sample_rois of the product's loader vs pairs captured from the
    imported reference (tests/golden/make_golden_from_reference.py)."""
    import os
    from detectron.core import config as c
    from detectron.roi_data import wsl
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_rois.npz'))
    c.reset_cfg()
    c.cfg.MODEL.NUM_CLASSES = 21
    c.cfg.TRAIN.BATCH_SIZE_PER_IM = int(z['batch_size_per_im'])
    proj = wsl._project_im_rois(z['boxes'].copy(), float(z['scale']), z['crop'])
    assert np.array_equal(proj, z['projected'])
    entry = dict(boxes=z['boxes'].copy(), obn_scores=z['obn_scores'], gt_classes=z['gt_classes'])
    blob = wsl._sample_rois(entry, float(z['scale']), z['crop'], 2)
    assert np.array_equal(blob['rois'], z['s_rois'])
    assert np.array_equal(blob['obn_scores'], z['s_obn'])
    assert np.array_equal(blob['labels_int32'], z['s_labels_int32'])
    assert np.array_equal(blob['labels_oh'], z['s_labels_oh'])
    c.reset_cfg()


def test_nms_known_answers_and_host_mirror():
    """cython_nms.pyx:36-87: +1 areas, suppress at IoU >= thresh, kept indices ascending."""
    import sys, os
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', 'na-fwebsod_amd'))
    from oracle import oracle
    from detectron.core import test_wsl
    d = np.array([[0, 0, 9, 9, 0.9], [5, 5, 14, 14, 0.8], [0, 0, 9, 9, 0.7], [20, 20, 29, 29, 0.6]],
                 np.float32)
    # inter 25, union 175 -> IoU = 1/7
    assert oracle.nms(d, 0.3).tolist() == [0, 1, 3]
    assert oracle.nms(d, np.float32(25.0) / np.float32(175.0)).tolist() == [0, 3]   # >= is inclusive
    assert oracle.nms(d, 0.15).tolist() == [0, 1, 3]
    assert oracle.nms(d[::-1].copy(), 0.3).tolist() == [0, 2, 3]                     # ascending index
    rng = np.random.default_rng(5)
    for n in (1, 17, 300):
        xy = rng.integers(0, 60, (n, 2)).astype(np.float32)
        wh = rng.integers(1, 40, (n, 2)).astype(np.float32)
        sc = rng.integers(0, 50, (n, 1)).astype(np.float32) / 50.0        # many tied scores
        dets = np.hstack([xy, xy + wh, sc]).astype(np.float32)
        for th in (0.3, 0.5, 0.0):
            assert test_wsl.nms(dets, th) == oracle.nms(dets, th).tolist()
    assert test_wsl.nms(np.zeros((0, 5), np.float32), 0.5) == []


def test_min_entropy_loss_known_answer():
    """min_entropy_loss_op.cc:7-98 by hand: X=[[.5,.25],[.1,.9]], L=[1,0] -> two terms."""
    from oracle import oracle
    x = np.array([[.5, .25], [.1, .9]], np.float32)
    l = np.array([[1, 0]], np.float32)
    want = -(.5 * np.log(.5) + .1 * np.log(.1)) / 2
    assert abs(float(oracle.min_entropy_loss(x, l)) - want) < 1e-6
    g = oracle.min_entropy_loss_grad(x, l, 0.1)
    assert np.allclose(g, [[0.05 * (-1 - np.log(.5)), 0], [0.05 * (-1 - np.log(.1)), 0]], atol=1e-7)
    # p = 0 is clamped to 1e-20; the gradient is capped at 1e4
    g0 = oracle.min_entropy_loss_grad(np.zeros((1, 2), np.float32), l, 1e6)
    assert g0[0, 0] == np.float32(1e4) and g0[0, 1] == 0
    assert oracle.min_entropy_loss(np.zeros((3, 2), np.float32), l) == np.float32(1e-20 * -np.log(np.float32(1e-20)))


def test_roi_label_known_answers():
    """roi_label_op.cc:10-123 by hand.  4 proposals, 3 classes (scores with a background column 0),
    labels {class 0, class 2}: picks = argmax of column 1 and of column 3 (first index wins a tie,
    a proposal is picked once), then every proposal follows the pick it overlaps most."""
    from oracle import oracle
    S = np.array([[.1, .7, .0, .2],
                  [.1, .7, .0, .9],      # ties with row 0 on class 0 -> row 0 wins; best of class 2
                  [.1, .1, .0, .3],
                  [.1, .2, .0, .1]], np.float32)
    L = np.array([[1, 0, 1]], np.float32)
    U = np.array([[1.0, .3, .6, .05],
                  [.3, 1.0, .2, .45],
                  [.6, .2, 1.0, .0],
                  [.05, .45, .0, 1.0]], np.float32)
    st = np.zeros((4,), np.float32)
    rl, rw = oracle.roi_label(S, U, L, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.1, stats=st)
    # picks: (n=0, class 0, p=.7), (n=1, class 2, p=.9)
    # roi 0: IoU 1.0 with pick 0 -> fg, label 1, w .7;  roi 1: IoU 1.0 with pick 1 -> label 3, w .9
    # roi 2: IoU .6 with pick 0 -> fg label 1 w .7;  roi 3: best IoU .45 (pick 1) in [.1,.5) -> bg, w .9
    assert rl.tolist() == [1, 3, 1, 0]
    assert np.allclose(rw, [.7, .9, .7, .9])
    assert np.allclose(st, [3, 1, .7 + .9 + .7, .9])
    # below bg_lo: the class label with weight 0; class weights replace the pick's score
    rl, rw = oracle.roi_label(S, U, L, CW=np.array([.25, .5, .75], np.float32), fg_thresh=0.5,
                              bg_thresh_hi=0.5, bg_thresh_lo=0.46)
    assert rl.tolist() == [1, 3, 1, 3] and np.allclose(rw, [.25, .75, .25, 0])
    # a score matrix without the background column, top_k = 2: the picked list is shared
    rl, rw = oracle.roi_label(S[:, 1:], U, L, top_k=2)
    # class 0 picks rows 0 then 1 (tie .7); class 2 picks the best REMAINING of column 2: row 2
    # (.3), then row 3 (.1): every proposal is a pick and follows itself (IoU 1)
    assert rl.tolist() == [1, 1, 3, 3]
    assert np.allclose(rw, [.7, .7, .3, .1])
    with pytest.raises(ValueError):
        oracle.roi_label(S, U, L, num_pos=2)   # a binding cap needs the reference's random order


def test_softmax_with_loss_n_known_answers():
    """softmax_with_loss_n_op.cc:152-357 by hand, incl. the docstring example of the op itself
    (.cc:60-75: logits [.1,.4,.7,1.5,.2], label 4, scale 5 -> loss 10.667433)."""
    from oracle import oracle
    x = np.array([[.1, .4, .7, 1.5, .2]], np.float32)
    p, loss = oracle.softmax_with_loss_n(x, np.array([4], np.int32), None, scale=5.0)
    assert np.allclose(p, [[0.10715417, 0.144643, 0.19524762, 0.4345316, 0.11842369]], atol=1e-7)
    assert abs(float(loss) - 10.667433) < 2e-6
    # weights: loss = sum(-w log p_t) / sum(w); gradient divides by the COUNT of w > 1e-12
    x = np.log(np.array([[.5, .25, .25], [.1, .2, .7], [.3, .3, .4]], np.float32))
    t = np.array([0, 2, 1], np.int32)
    w = np.array([2.0, 0.0, 0.5], np.float32)
    p, loss = oracle.softmax_with_loss_n(x, t, w)
    assert np.allclose(p, np.exp(x), atol=1e-7)
    assert abs(float(loss) - (-(2 * np.log(.5) + .5 * np.log(.3)) / 2.5)) < 1e-6
    g = oracle.softmax_with_loss_n_grad(t, w, p, 3.0)
    want = (np.exp(x) - np.eye(3, dtype=np.float32)[t]) * w[:, None] * 3.0 / 2.0
    assert np.allclose(g, want, atol=1e-6)
    # all weights zero: loss 0 and the gradient is left unscaled (all zeros here)
    p, loss = oracle.softmax_with_loss_n(x, t, np.zeros((3,), np.float32))
    assert loss == 0 and not oracle.softmax_with_loss_n_grad(t, np.zeros((3,), np.float32), p, 1.0).any()
    with pytest.raises(ValueError):
        oracle.softmax_with_loss_n(x, np.array([0, 3, 1], np.int32))


def test_roi_entropy_and_box_with_nms_limit_known_answers():
    """roi_entropy_op.cu:24-112: E_c = 1 - H(p)/log N_c over the kept detections of class c."""
    from oracle import oracle
    # class 1 (rm_bg -> slot 0): uniform over 4 -> E 0; class 2: one detection -> E 1;
    # class 3: p = (.75, .25) -> 1 - H/log 2
    S = np.array([.2, .2, .2, .2, .9, .3, .1], np.float32)
    C = np.array([1, 1, 1, 1, 2, 3, 3], np.float32)
    E = oracle.roi_entropy(S, C, 4)
    h = -(.75 * np.log(.75) + .25 * np.log(.25)) / np.log(2)
    assert np.allclose(E, [[0, 1, 1 - h, 1]], atol=1e-6)      # class 4 has no detection: stays 1
    # BoxWithNMSLimit, gate form: per class, score filter -> NMS -> descending score
    boxes = np.array([[0, 0, 9, 9], [1, 1, 10, 10], [20, 20, 29, 29]], np.float32)
    scores = np.array([[.1, .9, .0], [.1, .8, .6], [.1, .5, .7]], np.float32)
    s, b, c = oracle.box_with_nms_limit(scores, np.tile(boxes, (1, 3)), score_thresh=1e-11,
                                        nms_thresh=0.5, detections_per_im=999999)
    # class 1: boxes 0 and 1 overlap (IoU 81/119 > .5): keep 0 (.9) and 2 (.5); class 2: .7 then .6
    assert np.allclose(s, [.9, .5, .7, .6]) and c.tolist() == [1, 1, 2, 2]
    assert np.array_equal(b, boxes[[0, 2, 2, 1]])
    # the image-wide cut keeps EXACTLY detections_per_im entries, ties at the cut included
    # (ADVICE r2): two .7s compete for the second slot, the earlier (class, row) stays
    scores2 = np.array([[.1, .9, .0], [.1, .8, .7], [.1, .7, .7]], np.float32)
    s, b, c = oracle.box_with_nms_limit(scores2, np.tile(boxes, (1, 3)), score_thresh=1e-11,
                                        nms_thresh=0.5, detections_per_im=2)
    assert np.allclose(s, [.9, .7]) and c.tolist() == [1, 1]
