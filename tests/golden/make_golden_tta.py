#!/usr/bin/env python3
"""Golden fixtures of the test-time-augmentation paths, from the IMPORTED reference Python (runs
only in the build container; same stub-import harness as make_golden_from_reference.py):

  reference_tta.npz
    aug_<case>_*   core/test_wsl.py:181-281 `im_detect_bbox_aug` run with a recording workspace and
                   a deterministic stand-in network (scores = a function of the fed rois / obn rows
                   only): every pass's fed rois, and the combined (scores_c, boxes_c).  Cases:
                     avg    the yaml's TTA (H_FLIP, SCALES 480/576/864/1200 + flips, AVG / ID)
                     union  the same passes combined with UNION / UNION
                     ar     UNION / UNION with ASPECT_RATIOS (1.5, 0.75) + ASPECT_RATIO_H_FLIP
                   The reference's own prep_im_for_blob computes every im_scale; only cv2.resize
                   (absent here) is replaced by a function returning zeros of the resized shape.
    vote_*         utils/boxes.py:262-318 `box_voting` (all scoring methods) on top of the
                   reference's own cython_bbox built into oracle/_ref (make -C oracle ref)
    ar_boxes_*     utils/boxes.py:254-259 `aspect_ratio`; flip_* utils/boxes.py `flip_boxes`
  reference_cfgs.json
    cfg trees after merging configs/flickr_clean/na_wsddn_V-16-C5_1x.yaml and
    configs/flickr_coco/na_wsddn_V-16-C5_1x.yaml (the other two hot-path yamls)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_tta.py
"""
import glob
import importlib.util
import json
import os
import subprocess
import sys
from unittest import mock

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden_from_reference as base  # noqa: E402

REF = base.REF


def setup(yaml_rel):
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, base._StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    rcfg.merge_cfg_from_file(os.path.join(REF, yaml_rel))
    return rcfg


def plain(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, dict):
        return {k: plain(x) for k, x in v.items()}
    if isinstance(v, tuple):
        return list(v)
    return v


def job_cfg(yaml_rel):
    rcfg = setup(yaml_rel)
    cfg = rcfg.cfg
    tree = {k: plain(dict(cfg[k])) for k in ('MODEL', 'TRAIN', 'TEST', 'SOLVER', 'FAST_RCNN', 'WSL',
                                              'WEBLY', 'DATA_LOADER')}
    for k in ('NUM_GPUS', 'USE_NCCL', 'DEDUP_BOXES', 'PIXEL_MEANS', 'RNG_SEED', 'MEMONGER', 'VIS',
              'VIS_TH'):
        tree[k] = plain(cfg[k])
    print(json.dumps(tree, sort_keys=True))


def fake_resize(im, dsize=None, dst=None, fx=None, fy=None, interpolation=None):
    """Stand-in for the absent cv2.resize: zeros of the shape cv2 would return (the recorded
    values never depend on pixel data)."""
    h, w = im.shape[:2]
    if dsize is not None:
        return np.zeros((dsize[1], dsize[0]) + im.shape[2:], im.dtype)
    return np.zeros((int(np.round(h * fy)), int(np.round(w * fx))) + im.shape[2:], im.dtype)


def job_aug():
    rcfg = setup('configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml')
    cfg = rcfg.cfg
    from detectron.core import test_wsl as rt
    rt.blob_utils.cv2.resize = fake_resize
    rt.image_utils.cv2.resize = fake_resize
    np.float, np.int = float, int        # the reference pins numpy 1.x
    k = cfg.MODEL.NUM_CLASSES
    rng = np.random.RandomState(23)
    n = 48
    h_im, w_im = 375, 500
    boxes = np.floor(rng.uniform(0, 300, (n, 4))).astype(np.float32)
    boxes[:, 2] = np.minimum(boxes[:, 0] + np.floor(rng.uniform(21, 190, n)), w_im - 1)
    boxes[:, 3] = np.minimum(boxes[:, 1] + np.floor(rng.uniform(21, 70, n)), h_im - 1)
    boxes[9] = boxes[4]                                   # duplicates / near twins: dedup collisions
    boxes[17] = boxes[6] + np.array([1, 0, 1, 0], np.float32)
    obn = rng.uniform(0, 1, (n, 1)).astype(np.float32)
    im = np.zeros((h_im, w_im, 3), np.uint8)
    fed, log = {}, []

    def feed(name, v):
        fed[str(name)] = np.array(v)

    def fetch(name):
        assert str(name) == 'cls_prob'
        r = fed['rois']
        log.append(r.copy())
        b = (r[:, 1:5].sum(1, keepdims=True) * 0.001 + fed['obn_scores']).astype(np.float32)
        return (b * (1.0 + np.arange(k, dtype=np.float32)[None, :] * 0.03125)).astype(np.float32)

    rt.workspace = mock.MagicMock()
    rt.workspace.FeedBlob = feed
    rt.workspace.FetchBlob = fetch
    rt.core = mock.MagicMock()
    rt.core.ScopedName = lambda s: s
    model = mock.MagicMock()
    out = dict(aug_boxes=boxes, aug_obn=obn, aug_im_shape=np.array([h_im, w_im], np.int64))
    cases = {
        'avg': [],
        'union': ['TEST.BBOX_AUG.SCORE_HEUR', 'UNION', 'TEST.BBOX_AUG.COORD_HEUR', 'UNION'],
        'ar': ['TEST.BBOX_AUG.SCORE_HEUR', 'UNION', 'TEST.BBOX_AUG.COORD_HEUR', 'UNION',
               'TEST.BBOX_AUG.SCALES', '(480,)', 'TEST.BBOX_AUG.ASPECT_RATIOS', '(1.5, 0.75)',
               'TEST.BBOX_AUG.ASPECT_RATIO_H_FLIP', True],
    }
    for name, over in cases.items():
        rcfg.merge_cfg_from_list(['TEST.BBOX_AUG.ENABLED', True] + over)
        del log[:]
        scores_c, boxes_c, im_scale_i = rt.im_detect_bbox_aug(model, im, boxes.copy(), obn.copy())
        out['aug_%s_scores' % name] = np.asarray(scores_c)
        out['aug_%s_boxes' % name] = np.asarray(boxes_c)
        out['aug_%s_im_scale_i' % name] = np.float64(im_scale_i)
        out['aug_%s_npass' % name] = np.int64(len(log))
        for i, r in enumerate(log):
            out['aug_%s_fed%02d' % (name, i)] = r
        out['aug_%s_cfg' % name] = np.array(json.dumps(over))
    del np.float, np.int
    np.savez(os.path.join(HERE, '_tta_aug.npz'), **out)
    print('aug cases', {c: int(out['aug_%s_npass' % c]) for c in cases})


def job_vote():
    """box_voting / aspect_ratio / flip_boxes of the reference's utils/boxes.py, with the real
    cython_bbox (oracle/_ref, compiled from the reference's own .pyx) behind bbox_overlaps."""
    so = glob.glob(os.path.join(ROOT, 'oracle', '_ref', 'cython_bbox*.so'))
    assert so, 'run `make -C oracle ref` first'
    spec = importlib.util.spec_from_file_location('detectron.utils.cython_bbox', so[0])
    real = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(real)
    setup('configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml')
    sys.modules['detectron.utils.cython_bbox'] = real
    import detectron.utils.boxes as rb
    rb.cython_bbox = real
    rng = np.random.RandomState(31)
    n = 60
    b = np.floor(rng.uniform(0, 200, (n, 4))).astype(np.float32)
    b[:, 2:] = b[:, :2] + np.floor(rng.uniform(10, 120, (n, 2))).astype(np.float32)
    for i in range(0, 40, 4):                                # clusters of overlapping boxes
        b[i + 1] = b[i] + np.array([2, -1, 3, 1], np.float32)
        b[i + 2] = b[i] + np.array([-3, 2, -1, 4], np.float32)
    sc = rng.uniform(0.05, 1.0, (n, 1)).astype(np.float32)
    all_dets = np.hstack([b, sc]).astype(np.float32)
    top = all_dets[np.argsort(-sc[:, 0], kind='stable')[:12]].copy()
    out = dict(vote_all=all_dets, vote_top=top)
    for method, beta in (('ID', 1.0), ('TEMP_AVG', 0.5), ('AVG', 1.0), ('IOU_AVG', 1.0),
                         ('GENERALIZED_AVG', 2.0), ('QUASI_SUM', 0.5)):
        for th in (0.8, 0.5):
            out['vote_%s_%d' % (method, int(th * 10))] = rb.box_voting(top, all_dets, th,
                                                                       scoring_method=method, beta=beta)
    out['vote_overlaps'] = rb.bbox_overlaps(top[:, :4].astype(np.float32),
                                            all_dets[:, :4].astype(np.float32))
    tiled = np.tile(b[:10], (1, 3))
    out.update(ar_boxes_in=tiled, ar_boxes_15=rb.aspect_ratio(tiled, 1.5),
               ar_boxes_075=rb.aspect_ratio(tiled, 0.75),
               ar_boxes_inv=rb.aspect_ratio(rb.aspect_ratio(tiled, 1.5), 1.0 / 1.5),
               flip_in=tiled, flip_out=rb.flip_boxes(tiled, 500),
               flip_twice=rb.flip_boxes(rb.flip_boxes(tiled, 500), 500))
    np.savez(os.path.join(HERE, '_tta_vote.npz'), **out)
    print('vote ok')


def main():
    if len(sys.argv) > 1:
        return {'cfg': lambda: job_cfg(sys.argv[2]), 'aug': job_aug, 'vote': job_vote}[sys.argv[1]]()
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE='1')
    cfgs = {}
    for name in ('flickr_clean', 'flickr_coco'):        # one process per yaml: cfg is a module global
        rel = 'configs/%s/na_wsddn_V-16-C5_1x.yaml' % name
        txt = subprocess.check_output([sys.executable, __file__, 'cfg', rel], env=env, text=True)
        cfgs[name] = json.loads(txt.strip().splitlines()[-1])
    with open(os.path.join(HERE, 'reference_cfgs.json'), 'w') as f:
        json.dump(cfgs, f, indent=1, sort_keys=True)
    merged = {}
    for job in ('aug', 'vote'):
        print(subprocess.check_output([sys.executable, __file__, job], env=env, text=True).strip())
        part = os.path.join(HERE, '_tta_%s.npz' % job)
        with np.load(part) as z:
            merged.update({k: z[k] for k in z.files})
        os.remove(part)
    np.savez_compressed(os.path.join(HERE, 'reference_tta.npz'), **merged)
    print('wrote reference_tta.npz (%d arrays), reference_cfgs.json' % len(merged))


if __name__ == '__main__':
    main()
