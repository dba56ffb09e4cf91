"""RoIIoU pinned on integer-valued boxes by the one piece of reference code
that compiles in this image untouched: detectron/utils/cython_bbox.pyx (`make -C oracle ref` ->
oracle/_ref/, git-ignored, travels with gpurun).

detectron/ops/roi_iou_op.cu:27-62 truncates the coordinates to int, takes w, h = int(max(d + 1., 0.)),
inters = w * h, uni = areas - inters and divides in float; cython_bbox.pyx:bbox_overlaps does the
same +1 arithmetic on the float32 coordinates themselves.  On integer-valued boxes whose areas stay
below 2^24 (every box of a 600 x 1000 or 1200 x 2000 image: the bench proposals, MCG's uint16 boxes)
every intermediate is an exact integer in both, so the two must agree BIT FOR BIT off the diagonal
(the diagonal is forced to 1 by the operator and is 1 by arithmetic in bbox_overlaps).  On
fractional coordinates (rois after x im_scale) they differ exactly where the truncation changes the
boxes: RoIIoU(frac) == bbox_overlaps(trunc(frac)).

The HIP kernel is held bit-exact to the oracle in tests/test_gpu_ops.py::test_roi_iou_bitexact and,
below (GPU), to the compiled reference module directly."""
import os

import numpy as np
import pytest

from test_datasets import _ref_cython_bbox

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref():
    ref = _ref_cython_bbox()
    if ref is None:
        pytest.skip('oracle/_ref/cython_bbox not built (make -C oracle ref, needs /root/reference)')
    return ref


def _rois(boxes):
    r = np.zeros((boxes.shape[0], 5), np.float32)
    r[:, 1:] = boxes
    return r


def _int_boxes(rng, n, h, w):
    x1 = rng.integers(0, w - 2, n)
    y1 = rng.integers(0, h - 2, n)
    x2 = np.minimum(x1 + rng.integers(0, w, n), w - 1)
    y2 = np.minimum(y1 + rng.integers(0, h, n), h - 1)
    return np.stack([x1, y1, x2, y2], 1).astype(np.float32)


def _bench_boxes(n, h, w, seed):
    from detectron.datasets import synthetic
    e = synthetic.make_roidb(1, n, 20, h, w, seed=seed)[0]
    b = np.asarray(e['boxes'], np.float32)
    assert np.array_equal(b, np.floor(b))           # the bench proposals are integer-valued
    return b


CASES = [('random 2500 in 600x1000', lambda: _int_boxes(np.random.default_rng(5), 2500, 600, 1000)),
         ('random 2100 in 1200x2000', lambda: _int_boxes(np.random.default_rng(6), 2100, 1200, 2000)),
         ('bench proposals 2000', lambda: _bench_boxes(2000, 600, 1000, 11)),
         ('bench proposals 4000 (configs[4])', lambda: _bench_boxes(4000, 375, 500, 11)),
         ('degenerate: 1-px boxes, duplicates, touching edges',
          lambda: np.array([[0, 0, 0, 0], [0, 0, 0, 0], [5, 5, 5, 9], [5, 9, 8, 9], [0, 0, 999, 599],
                            [999, 599, 999, 599], [10, 10, 19, 19], [20, 10, 29, 19],
                            [19, 19, 30, 30]], np.float32))]


@pytest.mark.parametrize('name,make', CASES, ids=[c[0] for c in CASES])
def test_oracle_roi_iou_equals_compiled_reference_bbox_overlaps_on_integer_boxes(name, make):
    from oracle import oracle
    ref = _ref()
    b = make()
    j = oracle.roi_iou(_rois(b))
    ov = ref.bbox_overlaps(b, b)
    # J[j, i] of boxes (i, j): both symmetric; compare whole matrices bit for bit
    assert j.dtype == ov.dtype == np.float32
    assert np.array_equal(j.view(np.int32), ov.view(np.int32)), \
        '%d of %d entries differ' % ((j != ov).sum(), j.size)
    assert (np.diag(j) == 1.0).all()
    # not a vacuous comparison: a fair share of pairs overlap
    assert (j[~np.eye(len(b), dtype=bool)] > 0).mean() > 0.02 or len(b) < 16


def test_oracle_roi_iou_on_fractional_boxes_is_the_reference_on_truncated_boxes():
    """Fractional coordinates: RoIIoU truncates first (roi_iou_op.cu:32-39), bbox_overlaps does
    not - so RoIIoU(b) == bbox_overlaps(trunc(b)) bit for bit, and it differs from
    bbox_overlaps(b) exactly on pairs whose truncated boxes give another quotient."""
    from oracle import oracle
    ref = _ref()
    rng = np.random.default_rng(9)
    b = _int_boxes(rng, 700, 600, 1000)
    scale = np.float32(600.0 / 375.0)               # what _project_im_rois multiplies by
    bf = (b * scale).astype(np.float32)
    assert (bf != np.floor(bf)).any()
    j = oracle.roi_iou(_rois(bf))
    tr = np.trunc(bf).astype(np.float32)
    ov_t = ref.bbox_overlaps(tr, tr)
    off = ~np.eye(len(b), dtype=bool)
    assert np.array_equal(j[off].view(np.int32), ov_t[off].view(np.int32))
    ov_f = ref.bbox_overlaps(bf, bf)
    differs = (j != ov_f) & off
    predicted = (ov_t != ov_f) & off
    assert np.array_equal(differs, predicted)
    assert differs.any()                             # truncation does matter on these boxes
    # known answer of SURVEY.md appendix B: 148.5 -> 148
    k = oracle.roi_iou(_rois(np.array([[148.5, 10.2, 200.9, 60.7], [148, 10, 200, 60]], np.float32)))
    assert k[0, 1] == 1.0 and k[1, 0] == 1.0


@pytest.mark.gpu
@pytest.mark.parametrize('name,make', CASES[:4], ids=[c[0] for c in CASES[:4]])
def test_hip_roi_iou_equals_compiled_reference_bbox_overlaps_on_integer_boxes(name, make, dev):
    """The product kernel (naws_roi_iou_fwd through detectron.ops.RoIIoU) against the compiled
    reference module itself, no restatement in between."""
    import torch
    from naws_hip import ops
    ref = _ref()
    b = make()
    j = ops.roi_iou(torch.from_numpy(_rois(b)).to(dev)).cpu().numpy()
    ov = ref.bbox_overlaps(b, b)
    assert np.array_equal(j.view(np.int32), ov.view(np.int32))
