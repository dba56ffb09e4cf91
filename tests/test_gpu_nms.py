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
