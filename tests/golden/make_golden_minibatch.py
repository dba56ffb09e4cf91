#!/usr/bin/env python3
"""Golden fixture for a WHOLE training minibatch: the IMPORTED reference's
`roi_data.minibatch_wsl.get_minibatch` (`minibatch_wsl.py:53-171`, `wsl.py:62-166`,
`utils/blob.py:67-131`) on a seeded roidb of three images - random scale per image, HSV jitter
draws, random crop, flipped entries, proposal projection into the cropped / scaled frame, labels -
under a seeded `np.random`.  What is captured is everything that does not depend on OpenCV's
pixel arithmetic (absent here): the blobs' shapes, `data_ids`, `rois`, `obn_scores`,
`labels_int32`, `labels_oh`, the per-image scale and crop the call used, and the next draw of the
RNG afterwards (= the number and order of draws consumed).

cv2 is a MagicMock with three functions replaced: `imread` returns a seeded uint8 image of the
entry's size, `cvtColor` is the identity, `resize` returns zeros of the size OpenCV would produce
(cvRound of size * scale).

Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_minibatch.py
"""
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_from_reference import REF, _StubFinder  # noqa: E402

from minibatch_inputs import make_roidb, fake_image  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', 1, 'TRAIN.BATCH_SIZE_PER_IM', 30])
    cfg = rcfg.cfg
    import detectron.roi_data.minibatch_wsl as mbw
    import detectron.utils.blob as blob_utils
    used = {'scales': [], 'resize': []}

    def resize(im, dsize, dst=None, fx=None, fy=None, interpolation=None):
        oh, ow = int(np.round(im.shape[0] * fy)), int(np.round(im.shape[1] * fx))
        used['resize'].append([list(im.shape[:2]), float(fx), [oh, ow]])
        return np.zeros((oh, ow, 3), np.float32)
    for mod in (mbw, blob_utils):
        mod.cv2.imread = fake_image
        mod.cv2.cvtColor = lambda x, code: np.array(x)
        mod.cv2.resize = resize
    out = {}
    for tag, over in (('default', []),
                      ('crop_no_distortion', ['WSL.USE_CROP', True, 'WSL.USE_DISTORTION', False]),
                      ('single_scale', ['TRAIN.SCALES', (600,), 'TRAIN.MAX_SIZE', 1000,
                                        'WSL.USE_CROP', False, 'WSL.USE_DISTORTION', False])):
        rcfg.merge_cfg_from_list(over)
        del used['resize'][:]
        np.random.seed(77)
        blobs, valid = mbw.get_minibatch(make_roidb())
        nxt = np.random.random()
        out[tag + '__valid'] = np.array(bool(valid))
        out[tag + '__next_draw'] = np.float64(nxt)
        out[tag + '__data_shape'] = np.array(blobs['data'].shape, np.int64)
        for k in ('data_ids', 'rois', 'obn_scores', 'labels_int32', 'labels_oh'):
            out[tag + '__' + k] = np.asarray(blobs[k])
        out[tag + '__im_scales'] = np.array([r[1] for r in used['resize']], np.float64)
        out[tag + '__resized_from'] = np.array([r[0] for r in used['resize']], np.int64)
        out[tag + '__resized_to'] = np.array([r[2] for r in used['resize']], np.int64)
        out[tag + '__cfg'] = np.array([str(cfg.TRAIN.SCALES), str(cfg.TRAIN.MAX_SIZE),
                                       str(cfg.WSL.USE_CROP), str(cfg.WSL.USE_DISTORTION),
                                       str(cfg.WSL.CROP), str(cfg.WSL.SATURATION), str(cfg.WSL.EXPOSURE)])
        print(tag, out[tag + '__data_shape'], out[tag + '__im_scales'], out[tag + '__rois'].shape,
              out[tag + '__labels_int32'].ravel(), nxt)
    np.savez(os.path.join(HERE, 'reference_minibatch.npz'), **out)


if __name__ == '__main__':
    main()
