"""Host side of the test-time augmentations and the post-NMS branches:
`im_detect_bbox_aug` with scale / flip / aspect-ratio passes and the AVG / UNION heuristics,
`box_voting`, `aspect_ratio`, `flip_boxes` against values captured from the IMPORTED reference
(tests/golden/make_golden_tta.py -> reference_tta.npz); soft-NMS against the oracle's C
restatement of cython_nms.pyx (the .pyx does not compile under numpy 2) and hand-derived values."""
import json
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_tta.npz'))
CFGS = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'reference_cfgs.json')))
YAML = os.path.join(ROOT, 'na-fwebsod_amd', 'configs', '%s', 'na_wsddn_V-16-C5_1x.yaml')


class RecordingExecutor(object):
    """The fixture's deterministic stand-in network: scores are a function of the fed roi / obn
    rows only (make_golden_tta.py `fetch`)."""
    engine = None

    def __init__(self, k):
        self.device = torch.device('cpu')
        self.k = k
        self.fed_rois = []

    def feed(self, blobs):
        self.blobs = blobs

    def run(self):
        pass

    def fetch(self, name):
        assert name == 'cls_prob'
        r = self.blobs['rois'].numpy()
        self.fed_rois.append(r.copy())
        b = (r[:, 1:5].sum(1, keepdims=True) * 0.001 + self.blobs['obn_scores'].numpy()).astype(np.float32)
        out = (b * (1.0 + np.arange(self.k, dtype=np.float32)[None, :] * 0.03125)).astype(np.float32)
        return torch.from_numpy(out)


@pytest.mark.parametrize('case', ['avg', 'union', 'ar'])
def test_im_detect_bbox_aug_matches_reference(cfgmod, case):
    from detectron.core import test_wsl
    c = cfgmod
    c.merge_cfg_from_file(YAML % 'flickr_voc')
    over = json.loads(str(GOLD['aug_%s_cfg' % case]))
    c.merge_cfg_from_list(['TEST.BBOX_AUG.ENABLED', True, 'NAWS.DEVICE_PREP', False,
                           'NAWS.TTA_PAIR_FLIPS', False] + over)
    h, w = [int(v) for v in GOLD['aug_im_shape']]
    im = np.zeros((h, w, 3), np.uint8)
    ex = RecordingExecutor(c.cfg.MODEL.NUM_CLASSES)
    scores_c, boxes_c = test_wsl.im_detect_bbox_aug(ex, im, GOLD['aug_boxes'].copy(),
                                                    GOLD['aug_obn'].copy())
    npass = int(GOLD['aug_%s_npass' % case])
    assert len(ex.fed_rois) == npass
    for i in range(npass):                 # every pass feeds what the reference fed, in its order
        assert np.array_equal(ex.fed_rois[i], GOLD['aug_%s_fed%02d' % (case, i)]), (case, i)
    assert scores_c.dtype == GOLD['aug_%s_scores' % case].dtype
    assert np.array_equal(scores_c, GOLD['aug_%s_scores' % case])
    assert np.array_equal(boxes_c, GOLD['aug_%s_boxes' % case])


def test_tta_pass_list_matches_the_reference_order(cfgmod):
    """tta_passes() (what the device path walks) = the order in which the reference fed its passes."""
    from detectron.core import test_wsl
    from detectron.roi_data.minibatch_wsl import get_im_scale
    c = cfgmod
    c.merge_cfg_from_file(YAML % 'flickr_voc')
    c.merge_cfg_from_list(['TEST.BBOX_AUG.ENABLED', True])
    h, w = [int(v) for v in GOLD['aug_im_shape']]
    boxes = GOLD['aug_boxes']
    passes = test_wsl.tta_passes()
    assert len(passes) == int(GOLD['aug_avg_npass'])
    for i, (s, m, flip) in enumerate(passes):
        bx = test_wsl.flip_boxes(boxes, w) if flip else boxes
        rois = test_wsl.project_rois(bx, get_im_scale((h, w), s, m))
        u, _idx, _inv = test_wsl.dedup_rois(rois, c.cfg.DEDUP_BOXES)
        assert np.array_equal(u, GOLD['aug_avg_fed%02d' % i]), i


def test_box_voting_matches_reference():
    from detectron.utils import boxes as B
    top, alld = GOLD['vote_top'], GOLD['vote_all']
    assert np.array_equal(B.bbox_overlaps(top[:, :4], alld[:, :4]), GOLD['vote_overlaps'])
    for method, beta in (('ID', 1.0), ('TEMP_AVG', 0.5), ('AVG', 1.0), ('IOU_AVG', 1.0),
                         ('GENERALIZED_AVG', 2.0), ('QUASI_SUM', 0.5)):
        for th in (0.8, 0.5):
            got = B.box_voting(top, alld, th, scoring_method=method, beta=beta)
            want = GOLD['vote_%s_%d' % (method, int(th * 10))]
            assert got.dtype == want.dtype and np.array_equal(got, want), (method, th)
    with pytest.raises(NotImplementedError):
        B.box_voting(top, alld, 0.5, scoring_method='NOPE')


def test_aspect_ratio_and_flip_boxes_match_reference():
    from detectron.utils import boxes as B
    assert np.array_equal(B.aspect_ratio(GOLD['ar_boxes_in'], 1.5), GOLD['ar_boxes_15'])
    assert np.array_equal(B.aspect_ratio(GOLD['ar_boxes_in'], 0.75), GOLD['ar_boxes_075'])
    assert np.array_equal(B.aspect_ratio(B.aspect_ratio(GOLD['ar_boxes_in'], 1.5), 1.0 / 1.5),
                          GOLD['ar_boxes_inv'])
    assert np.array_equal(B.flip_boxes(GOLD['flip_in'], 500), GOLD['flip_out'])
    assert np.array_equal(B.flip_boxes(B.flip_boxes(GOLD['flip_in'], 500), 500), GOLD['flip_twice'])


def test_soft_nms_known_answers():
    """Hand-derived from cython_nms.pyx:98-203: A = (0,0,9,9) .9, B = (0,0,9,4) .8 -> IoU 50/100."""
    from detectron.utils import boxes as B
    from oracle import oracle
    d = np.array([[0, 0, 9, 9, .9], [0, 0, 9, 4, .8], [50, 50, 59, 59, .7]], np.float32)
    for fn in (B.soft_nms, oracle.soft_nms):
        out, keep = fn(d, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear')
        # round 1 picks A and halves B (.4); round 2 picks C (.7 > .4); B last
        assert keep == [0, 2, 1] and np.allclose(out[:, 4], [.9, .7, .4])
        out, keep = fn(d, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='gaussian')
        assert keep == [0, 2, 1] and np.allclose(out[:, 4], [.9, .7, .8 * np.exp(-.5)], rtol=1e-6)
        out, keep = fn(d, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='hard')
        # B's score becomes 0 < threshold: overwritten by the last box (C), N shrinks to 2
        assert keep == [0, 2] and np.allclose(out[:, 4], [.9, .7])
        out, keep = fn(d, sigma=0.5, overlap_thresh=0.6, score_thresh=0.001, method='linear')
        assert keep == [0, 1, 2] and np.allclose(out[:, 4], [.9, .8, .7])     # IoU .5 <= Nt
    assert B.soft_nms(np.zeros((0, 5), np.float32))[1] == []
    with pytest.raises(AssertionError):
        B.soft_nms(d, method='quadratic')


@pytest.mark.parametrize('n', [1, 2, 9, 120, 700])
def test_soft_nms_vectorised_form_equals_the_sequential_restatement(n):
    """The product's numpy form (one vector expression per round + a replay of the overwrite-by-
    the-last-box compaction) against the oracle's statement-for-statement C loop: decayed scores
    bit-identical, kept indices in the same order, ties and heavy discarding included."""
    from detectron.utils import boxes as B
    from oracle import oracle
    rng = np.random.default_rng(n)
    for method in ('linear', 'gaussian', 'hard'):
        for thr in (0.0001, 0.05, 0.3):
            b = np.floor(rng.uniform(0, 300, (n, 4))).astype(np.float32)
            b[:, 2:] = b[:, :2] + np.floor(rng.uniform(5, 150, (n, 2))).astype(np.float32)
            sc = (rng.uniform(0, 1, (n, 1)) ** 3).astype(np.float32)
            sc[rng.integers(0, n, n // 5)] = np.float32(0.25)
            d = np.hstack([b, sc]).astype(np.float32)
            want, wk = oracle.soft_nms(d, 0.5, 0.3, thr, method)
            got, gk = B.soft_nms(d, 0.5, 0.3, thr, method)
            assert gk == wk and np.array_equal(got, want), (method, thr)


def test_resize_u8_known_answers():
    """cv2.resize 8-bit INTER_LINEAR, from the published fixed-point algorithm (PARITY UNPINNED:
    cv2 is absent): [0, 255] stretched to 4 pixels -> 0, 64, 191, 255 (taps .75/.25: 255 * 512
    >> 4 = 8160, * 2048 >> 16 = 255, (255 + 2) >> 2 = 64)."""
    from detectron.utils import image as I
    im = np.array([[[0], [255]]], np.uint8)
    assert I.resize_linear_u8(im, (4, 1))[0, :, 0].tolist() == [0, 64, 191, 255]
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (9, 12, 3), dtype=np.uint8)
    assert np.array_equal(I.resize_linear_u8(im, (12, 9)), im)
    assert I.aspect_ratio_rel(im, 1.5).shape == (9, 18, 3)
    assert I.aspect_ratio_rel(im, 0.75).shape == (9, 9, 3)
    assert np.unique(I.aspect_ratio_rel(np.full((5, 7, 3), 201, np.uint8), 1.4)).tolist() == [201]


@pytest.mark.parametrize('name', ['flickr_clean', 'flickr_coco'])
def test_the_other_hot_path_yamls_match_the_reference_cfg(cfgmod, name):
    c = cfgmod
    c.merge_cfg_from_file(YAML % name)

    def walk(gold, node, path):
        for k, v in gold.items():
            if k not in node:
                continue                       # a key outside the hot path: not declared here
            if isinstance(v, dict):
                walk(v, node[k], path + [k])
            else:
                mine = node[k]
                mine = mine.tolist() if isinstance(mine, np.ndarray) else mine
                mine = list(mine) if isinstance(mine, tuple) else mine
                assert mine == v, ('.'.join(path + [k]), mine, v)
    walk(CFGS[name], c.cfg, [])
    for sect in ('SOFT_NMS', 'BBOX_VOTE', 'BBOX_AUG'):
        assert set(CFGS[name]['TEST'][sect]) == set(c.cfg.TEST[sect]), sect


def test_dataset_catalog_paths_of_the_hot_path_sets():
    from detectron.datasets import dataset_catalog as dc
    assert dc.get_im_dir('flickr_clean').endswith('/flickr_clean/image')      # sic, upstream
    assert dc.get_ann_fn('flickr_clean').endswith('/flickr_clean/image.json')
    assert dc.get_im_dir('flickr_voc').endswith('/flickr_voc/images')
    assert dc.get_ann_fn('flickr_coco').endswith('/flickr_coco/images.json')


@pytest.mark.parametrize('soft,vote', [(True, False), (False, True), (True, True)])
def test_box_results_soft_nms_and_voting_branches(cfgmod, soft, vote):
    """box_results_with_nms_and_limit with TEST.SOFT_NMS / TEST.BBOX_VOTE against a composition
    of the oracle's soft-NMS / NMS and the (reference-pinned) box_voting, per class."""
    from detectron.core import test_wsl
    from detectron.utils import boxes as B
    from oracle import oracle
    c = cfgmod
    c.merge_cfg_from_file(YAML % 'flickr_voc')
    c.merge_cfg_from_list(['TEST.SOFT_NMS.ENABLED', soft, 'TEST.BBOX_VOTE.ENABLED', vote,
                           'TEST.BBOX_VOTE.SCORING_METHOD', 'AVG', 'TEST.BBOX_VOTE.VOTE_TH', 0.6,
                           'TEST.SOFT_NMS.METHOD', 'gaussian', 'TEST.DETECTIONS_PER_IM', 40,
                           'NAWS.HOST_NMS', True])
    rng = np.random.default_rng(4)
    n, k = 150, 21
    b = np.floor(rng.uniform(0, 200, (n, 4))).astype(np.float32)
    b[:, 2:] = b[:, :2] + np.floor(rng.uniform(10, 120, (n, 2))).astype(np.float32)
    scores = (rng.uniform(0, 1, (n, k)) ** 5).astype(np.float32)
    scores[:, 5] = 0                                   # a class without candidates
    s, bx, cls_boxes = test_wsl.box_results_with_nms_and_limit(scores, np.tile(b, (1, k)))
    want = [np.zeros((0, 5), np.float32)]
    for j in range(1, k):
        inds = np.where(scores[:, j] > c.cfg.TEST.SCORE_THRESH)[0]
        dets = np.hstack([b[inds], scores[inds, j:j + 1]]).astype(np.float32)
        if soft:
            nd = oracle.soft_nms(dets, 0.5, c.cfg.TEST.NMS, 0.0001, 'gaussian')[0]
        else:
            nd = dets[oracle.nms(dets, c.cfg.TEST.NMS)]
        if vote and len(nd):
            nd = B.box_voting(nd, dets, 0.6, scoring_method='AVG')
        want.append(nd)
    allsc = np.hstack([w[:, 4] for w in want[1:]])
    th = np.sort(allsc)[-40]
    total = 0
    for j in range(1, k):
        wj = want[j][want[j][:, 4] >= th]
        assert np.array_equal(cls_boxes[j], wj), j
        total += len(wj)
    assert total == len(s) == len(bx) >= 40
