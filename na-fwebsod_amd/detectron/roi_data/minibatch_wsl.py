"""Minibatch construction for the WSL path: image blob + RoI blobs.
Mirrors detectron/roi_data/minibatch_wsl.py:25-50 (blob order contract), :53-90
(get_minibatch), :93-108 (_get_image_id_blob), :111-171 (_get_image_blob) and
detectron/utils/blob.py:67-131 (im_list_to_blob / prep_im_for_blob).

cv2 is not available on the MI355X image: file images are decoded and resized with PIL
(bilinear), synthetic roidb entries (detectron.datasets.synthetic) generate their pixels
from a per-entry seed.  HSV distortion is the one augmentation not restated (it needs the
OpenCV colour transform); it is rejected loudly rather than skipped silently.
"""
import os

import numpy as np
import numpy.random as npr

from detectron.core.config import cfg
import detectron.roi_data.wsl as wsl_roi_data


def get_minibatch_blob_names(is_training=True):
    """Order in which the loader emits blobs (and the net dequeues them)."""
    return ['data', 'data_ids'] + wsl_roi_data.get_wsl_blob_names(is_training=is_training)


def get_minibatch(roidb):
    blobs = {k: [] for k in get_minibatch_blob_names()}
    im_blob, im_scales, im_crops = _get_image_blob(roidb)
    # crops are (y1,x1,y2,x2) -> (x1,y1,x2,y2)
    im_crops = np.array(im_crops, dtype=np.int32)[:, (1, 0, 3, 2)]
    blobs['data'] = im_blob
    blobs['data_ids'] = _get_image_id_blob(roidb)
    valid = wsl_roi_data.add_wsl_blobs(blobs, im_scales, im_crops, roidb)
    return blobs, valid


def _get_image_id_blob(roidb):
    ids = []
    for entry in roidb:
        stem = os.path.splitext(os.path.basename(entry['image']))[0]
        tail = stem.split('_')[-1]
        ids.append([int(tail) if tail.isdigit() else 0])
    return np.array(ids, dtype=np.int32).reshape(-1, 1)


def _read_image(entry):
    """HxWx3 BGR uint8."""
    if 'seed' in entry and not os.path.exists(entry['image']):
        rng = np.random.default_rng(entry['seed'])
        return rng.integers(0, 256, (entry['height'], entry['width'], 3), dtype=np.uint8)
    from PIL import Image
    with Image.open(entry['image']) as im:
        rgb = np.asarray(im.convert('RGB'))
    return rgb[:, :, ::-1].copy()


def prep_im_for_blob(im, pixel_means, target_size, max_size):
    """Mean-subtract, scale the short side to target_size (long side capped at max_size)."""
    im = im.astype(np.float32, copy=False)
    im = im - np.asarray(pixel_means, np.float32).reshape(1, 1, 3)
    size_min, size_max = min(im.shape[:2]), max(im.shape[:2])
    im_scale = float(target_size) / float(size_min)
    if np.round(im_scale * size_max) > max_size:
        im_scale = float(max_size) / float(size_max)
    if im_scale != 1.0:
        from PIL import Image
        h = int(round(im.shape[0] * im_scale))
        w = int(round(im.shape[1] * im_scale))
        chans = [np.asarray(Image.fromarray(im[:, :, c], mode='F').resize((w, h), Image.BILINEAR))
                 for c in range(3)]
        im = np.stack(chans, 2)
    return im, im_scale


def im_list_to_blob(ims):
    """Zero-pad to the largest H,W and emit NCHW float32."""
    if not isinstance(ims, list):
        ims = [ims]
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    blob = np.zeros((len(ims), max_shape[0], max_shape[1], 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, :im.shape[0], :im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2)).copy()


def _get_image_blob(roidb):
    scale_inds = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=len(roidb))
    ims, scales, crops = [], [], []
    for i, entry in enumerate(roidb):
        im = _read_image(entry)
        if entry['flipped']:
            im = im[:, ::-1, :]
        if cfg.WSL.USE_DISTORTION:
            raise NotImplementedError('WSL.USE_DISTORTION needs the OpenCV HSV transform, '
                                      'which this image lacks; set WSL.USE_DISTORTION False')
        if cfg.WSL.USE_CROP:
            shape = np.array(im.shape)
            crop_dims = shape[:2] * cfg.WSL.CROP
            r0, r1 = npr.random(), npr.random()
            s = shape[:2] - crop_dims
            s[0] *= r0
            s[1] *= r1
            crop = np.array([s[0], s[1], s[0] + crop_dims[0] - 1, s[1] + crop_dims[1] - 1],
                            dtype=np.int32)
            im = im[crop[0]:crop[2] + 1, crop[1]:crop[3] + 1, :]
        else:
            crop = np.array([0, 0, im.shape[0] - 1, im.shape[1] - 1], dtype=np.int32)
        im, sc = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_inds[i]],
                                  cfg.TRAIN.MAX_SIZE)
        ims.append(im)
        scales.append(sc)
        crops.append(crop)
    return im_list_to_blob(ims), scales, crops
