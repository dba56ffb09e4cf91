"""GPU parity of the per-class NMS against the oracle's restatement of
cython_nms.pyx: the kept set is bit-identical (index work)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(rng, r, c, integer=True):
    if integer:    # integer coordinates: IoUs land exactly on thresholds, many duplicates / ties
        xy = rng.integers(0, 80, (r, 2)).astype(np.float32)
        wh = rng.integers(1, 60, (r, 2)).astype(np.float32)
        sc = rng.integers(0, 200, (r, c)).astype(np.float32) / 200.0
    else:
        xy = rng.uniform(0, 900, (r, 2)).astype(np.float32)
        wh = np.exp(rng.uniform(np.log(21), np.log(600), (r, 2))).astype(np.float32)
        sc = rng.uniform(0, 1, (r, c)).astype(np.float32) ** 4
    return np.hstack([xy, xy + wh]).astype(np.float32), sc


def _check(dev, boxes, sc, sth, nth):
    from naws_hip import ops
    from oracle import oracle
    keep = ops.nms_per_class(torch.from_numpy(boxes).to(dev), torch.from_numpy(sc).to(dev), sth,
                             nth).cpu().numpy()
    for j in range(sc.shape[1]):
        inds = np.where(sc[:, j] > np.float32(sth))[0]
        dets = np.hstack([boxes[inds], sc[inds, j:j + 1]]).astype(np.float32)
        ref = inds[oracle.nms(dets, nth)]
        assert np.array_equal(np.where(keep[j])[0], ref), j


@pytest.mark.parametrize('r,c', [(1, 1), (63, 3), (64, 2), (65, 2), (300, 5), (1000, 3)])
@pytest.mark.parametrize('integer', [True, False])
def test_nms_matches_oracle(dev, r, c, integer):
    rng = np.random.default_rng(41 + r)
    boxes, sc = _case(rng, r, c, integer)
    for sth, nth in ((0.0, 0.5), (0.3, 0.3), (0.9, 0.5)):
        _check(dev, boxes, sc, sth, nth)
    _check(dev, boxes, sc, 0.0, 25.0 / 175.0)        # a threshold integer boxes hit exactly
    _check(dev, boxes, sc, 2.0, 0.5)                  # no candidates at all


def test_nms_fullsize_config5(dev):
    """BASELINE configs[4]: 4000 proposals, 20 classes, every box a candidate (TEST.SCORE_THRESH
    = 0 in the reference's yaml); also > 4096 boxes (two removed-words per lane)."""
    rng = np.random.default_rng(43)
    boxes, sc = _case(rng, 4000, 20, integer=False)
    _check(dev, boxes, sc, 0.0, 0.5)
    boxes, sc = _case(rng, 4500, 2, integer=True)
    _check(dev, boxes, sc, 0.0, 0.3)


def test_nms_class_tiled_boxes_and_empty(dev):
    from naws_hip import ops
    rng = np.random.default_rng(44)
    boxes, sc = _case(rng, 200, 4)
    tiled = np.tile(boxes, (1, 5))                    # [R, 4*(C+1)] like the reference pred_boxes
    k1 = ops.nms_per_class(torch.from_numpy(boxes).to(dev), torch.from_numpy(sc).to(dev), 0.1, 0.4)
    k2 = ops.nms_per_class(torch.from_numpy(tiled).to(dev), torch.from_numpy(sc).to(dev), 0.1, 0.4)
    assert torch.equal(k1, k2)
    e = ops.nms_per_class(torch.zeros((0, 4), device=dev), torch.zeros((0, 4), device=dev), 0.1, 0.4)
    assert e.shape == (4, 0)


@pytest.mark.parametrize('method', ['linear', 'gaussian', 'hard'])
def test_soft_nms_kernel_matches_the_sequential_restatement(dev, method):
    """naws_soft_nms_fwd (one workgroup per class, the list in LDS, the reference's swap /
    overwrite-by-the-last-box order replayed by a two-pointer compaction) against the oracle's
    statement-for-statement C loop of cython_nms.pyx:98-203: decayed scores bit-identical, kept
    indices in the same ORDER, for lists of 1 ... 4000 boxes incl. ties and heavy discarding."""
    import torch
    from naws_hip import ops
    from oracle import oracle
    rng = np.random.default_rng(70)
    sizes = [1, 2, 3, 17, 64, 65, 333, 1024, 1025, 2500, 4000, 0]
    n_max = max(sizes)
    for thr in (0.0001, 0.05, 0.3):
        packed = np.zeros((len(sizes), n_max, 5), np.float32)
        for k, n in enumerate(sizes):
            b = np.floor(rng.uniform(0, 300, (n, 4))).astype(np.float32)
            b[:, 2:] = b[:, :2] + np.floor(rng.uniform(5, 150, (n, 2))).astype(np.float32)
            sc = (rng.uniform(0, 1, (n, 1)) ** 3).astype(np.float32)
            if n >= 5:
                sc[rng.integers(0, n, n // 5)] = np.float32(0.25)           # ties
            packed[k, :n] = np.hstack([b, sc])
        out, keep, oc = ops.soft_nms_per_class(torch.from_numpy(packed).to(dev),
                                               torch.tensor(sizes, dtype=torch.int32, device=dev),
                                               0.5, 0.3, thr, {'hard': 0, 'linear': 1, 'gaussian': 2}[method])
        out, keep, oc = out.cpu().numpy(), keep.cpu().numpy(), oc.cpu().numpy()
        for k, n in enumerate(sizes):
            want, wk = oracle.soft_nms(packed[k, :n], 0.5, 0.3, thr, method)
            assert oc[k] == len(wk), (method, thr, n, oc[k], len(wk))
            assert keep[k, :oc[k]].tolist() == wk, (method, thr, n)
            assert np.array_equal(out[k, :oc[k]], want), (method, thr, n)


def test_box_results_soft_nms_on_the_device_equals_the_host_form(dev):
    """TEST.SOFT_NMS through box_results_with_nms_and_limit: the device kernel (default) and the
    numpy form (NAWS.HOST_NMS) return identical detections."""
    from detectron.core import config as c
    from detectron.core import test_wsl
    import os
    c.reset_cfg()
    try:
        c.merge_cfg_from_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                           'na-fwebsod_amd', 'configs', 'flickr_voc', 'na_wsddn_V-16-C5_1x.yaml'))
        c.merge_cfg_from_list(['TEST.SOFT_NMS.ENABLED', True, 'TEST.BBOX_VOTE.ENABLED', True,
                               'TEST.DETECTIONS_PER_IM', 60])
        rng = np.random.default_rng(71)
        n, k = 500, 21
        b = np.floor(rng.uniform(0, 300, (n, 4))).astype(np.float32)
        b[:, 2:] = b[:, :2] + np.floor(rng.uniform(10, 120, (n, 2))).astype(np.float32)
        scores = (rng.uniform(0, 1, (n, k)) ** 5).astype(np.float32)
        got = test_wsl.box_results_with_nms_and_limit(scores, np.tile(b, (1, k)))[2]
        c.cfg.NAWS.HOST_NMS = True
        want = test_wsl.box_results_with_nms_and_limit(scores, np.tile(b, (1, k)))[2]
        for j in range(1, k):
            assert np.array_equal(got[j], want[j]), j
    finally:
        c.reset_cfg()
