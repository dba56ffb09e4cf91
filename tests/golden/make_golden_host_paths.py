#!/usr/bin/env python3
"""More golden fixtures from the IMPORTED reference Python (runs only in the
build container, like make_golden_from_reference.py whose stub-import harness it reuses):

  reference_host_paths.npz
    dedup_*     core/test_wsl.py:102-178 `im_detect_bbox` run with a recording workspace: the rois
                it FEEDS after the dedup hash (:125-133), and the scores it returns after the
                scatter-back (:173-176), for proposals built to collide after x DEDUP_BOXES
    imgid_*     roi_data/minibatch_wsl.py:93-108 `_get_image_id_blob` on a list of file names
    mixup_*     roi_data/loader_wsl.py:130-168 `RoIDataLoader.get_next_minibatch` on a seeded
                stream that takes the bagging-mixup branch: the two-image minibatch that went in
                (from a patched get_minibatch) and the blended blobs that came out, plus lambda

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_host_paths.py
"""
import os
import random
import sys
from unittest import mock

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_from_reference as base  # noqa: E402

REF = base.REF


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, base._StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    cfg = rcfg.cfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    out = {}

    # ------------------------------------------------------------------ dedup + scatter-back
    from detectron.core import test_wsl as rt
    rng = np.random.RandomState(11)
    n = 64
    boxes = np.floor(rng.uniform(0, 400, (n, 4))).astype(np.float32)
    boxes[:, 2:] = boxes[:, :2] + np.floor(rng.uniform(21, 200, (n, 2))).astype(np.float32)
    # collisions after round(roi * 0.125): +-1..3 px twins, exact duplicates, and .5 ties
    boxes[10] = boxes[3] + np.array([1, 0, 2, 1], np.float32)
    boxes[11] = boxes[3]
    boxes[20] = boxes[7] + np.array([0, 3, 0, -2], np.float32)
    boxes[33] = np.array([4, 12, 100, 204], np.float32)      # 0.5 / 1.5 / 12.5 / 25.5 after x0.125
    boxes[34] = np.array([3, 11, 99, 203], np.float32)
    obn = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    im_scale = 688.0 / 375.0          # TEST.SCALE / short side: not exactly representable
    k = 21
    fed = {}
    # the reference's own _get_blobs / _get_rois_blob / _project_im_rois run (float64 product,
    # then float32): only the cv2-based image blob is replaced, and numpy's removed aliases
    # np.float / np.int (the reference pins numpy 1.x) are restored for the call
    np.float, np.int = float, int
    rt.blob_utils.get_image_blob = lambda im, ts, tms: (np.zeros((1, 3, 8, 8), np.float32), im_scale,
                                                        np.zeros((1, 3), np.float32))

    def feed(name, v):
        fed[str(name)] = np.array(v)

    def fetch(name):
        assert str(name) == 'cls_prob'
        r = fed['rois']
        # a deterministic "network": scores depend on the fed roi row only
        base_ = (r[:, 1:5].sum(1, keepdims=True) * 0.001 + fed['obn_scores']).astype(np.float32)
        return (base_ + np.arange(k, dtype=np.float32)[None, :] * 0.01).astype(np.float32)

    rt.workspace = mock.MagicMock()
    rt.workspace.FeedBlob = feed
    rt.workspace.FetchBlob = fetch
    rt.core = mock.MagicMock()
    rt.core.ScopedName = lambda s: s
    model = mock.MagicMock()
    scores, pred_boxes, sc = rt.im_detect_bbox(model, np.zeros((8, 8, 3), np.uint8), 688, 4000,
                                               boxes=boxes.copy(), obn_scores=obn.copy())
    del np.float, np.int
    out.update(dedup_boxes=boxes, dedup_obn=obn, dedup_im_scale=np.float64(im_scale),
               dedup_factor=np.float64(cfg.DEDUP_BOXES), dedup_fed_rois=fed['rois'],
               dedup_fed_obn=fed['obn_scores'], dedup_scores=scores, dedup_pred_boxes=pred_boxes)
    assert fed['rois'].shape[0] < n, 'the fixture must contain collisions'

    # ------------------------------------------------------------------ _get_image_id_blob
    from detectron.roi_data import minibatch_wsl as rmb
    names = ['/data/flickr/JPEGImages/2008_000123.jpg', 'images/000042.jpg', 'a/b/flickr_voc_77.png',
             'x/cat.jpg', 'COCO_train2014_000000581921.jpg', 'weird_12a.jpg', '/p/q/9.jpeg']
    blob = rmb._get_image_id_blob([{'image': s} for s in names])
    out.update(imgid_names=np.array(names), imgid_blob=blob)

    # ------------------------------------------------------------------ bagging-mixup blend
    from detectron.roi_data import loader_wsl as rl
    c = cfg.MODEL.NUM_CLASSES - 1
    rng = np.random.RandomState(5)
    two = {
        'data': rng.uniform(-120, 130, (2, 3, 24, 40)).astype(np.float32),
        'data_ids': np.array([[17], [99]], np.int32),
        'rois': np.vstack([np.hstack([np.zeros((5, 1)), rng.uniform(0, 30, (5, 4))]),
                           np.hstack([np.ones((7, 1)), rng.uniform(0, 30, (7, 4))])]).astype(np.float32),
        'obn_scores': rng.uniform(1, 2, (12, 1)).astype(np.float32),
        'labels_int32': np.array([3, 3], np.int32),
        'labels_oh': np.zeros((2, c), np.float32),
    }
    two['labels_oh'][0, 3] = 1
    two['labels_oh'][1, 3] = 1
    two['labels_oh'][1, 8] = 1
    rl.get_minibatch = lambda db: ({k_: v.copy() for k_, v in two.items()}, True)
    loader = object.__new__(rl.RoIDataLoader)
    gt = np.zeros((4,), np.int32)
    gt[0] = 4
    loader._roidb = [{'gt_classes': gt.copy()} for _ in range(6)]
    loader._class2idx = {4: [1, 2, 5]}
    loader._get_next_minibatch_inds = lambda: [0]
    seed = None
    for s in range(100):                      # a stream whose first draw takes the mixup branch
        np.random.seed(s)
        if np.random.random() > 0.8:
            seed = s
            break
    np.random.seed(seed)
    random.seed(seed)
    got = loader.get_next_minibatch()
    np.random.seed(seed)
    np.random.random()
    lam = np.random.beta(cfg.WEBLY.BAGGING_MIXUP_ALPHA, cfg.WEBLY.BAGGING_MIXUP_ALPHA)
    out.update(mixup_seed=np.int64(seed), mixup_lam=np.float64(lam),
               mixup_alpha=np.float64(cfg.WEBLY.BAGGING_MIXUP_ALPHA))
    for k_, v in two.items():
        out['mixup_in_' + k_] = v
    for k_, v in got.items():
        out['mixup_out_' + k_] = np.asarray(v)
    np.savez(os.path.join(HERE, 'reference_host_paths.npz'), **out)
    print('dedup: %d -> %d rois; image ids %s; mixup seed %d lam %.6f' % (
        n, fed['rois'].shape[0], blob.reshape(-1).tolist(), seed, lam))


if __name__ == '__main__':
    main()
