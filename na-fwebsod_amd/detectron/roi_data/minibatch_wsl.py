"""Minibatch construction for the WSL path: image blob + RoI blobs.
Mirrors detectron/roi_data/minibatch_wsl.py:25-50 (blob order contract), :53-90
(get_minibatch), :93-108 (_get_image_id_blob), :111-171 (_get_image_blob) and
detectron/utils/blob.py:67-131 (im_list_to_blob / prep_im_for_blob).

cv2 is not available on the MI355X image: file images are decoded with PIL, synthetic roidb
entries (detectron.datasets.synthetic) generate their pixels from a per-entry seed, and
cv2.resize(INTER_LINEAR) is restated (`resize_linear`, same pixel-centre mapping and edge rule).
With NAWS.DEVICE_PREP the float conversion, mean/std, flip, crop, resize and HWC->CHW padding run
on the GPU (naws_prep_image_fwd): the loader threads then only decode, and a minibatch carries
the raw uint8 images + their parameters (`_raw`) instead of a float `data` blob.
The HSV distortion (WSL.USE_DISTORTION: cv2.cvtColor BGR2HSV -> scale S, V -> HSV2BGR on uint8)
is restated from OpenCV's 8-bit algorithm (`distort_hsv`), on the host or inside the device prep.

PARITY UNPINNED for three third-party behaviours the reference inherits from OpenCV (an
un-vendored, un-pinned dependency that is absent here, so no vector of it could be captured):
(1) cv2.resize(INTER_LINEAR)'s fixed-point taps, (2) cv2.cvtColor's 8-bit BGR<->HSV tables, both
restated from the published algorithms and pinned only by hand-derived known answers
(tests/test_oracle_kat.py, tests/test_gpu_prep.py: device == oracle restatement bit for bit);
(3) JPEG decoding: cv2.imread and PIL both sit on libjpeg but may differ in IDCT / upsampling
options by +-1 in some pixels.  None of this touches the synthetic benchmark or the parity tests
(their pixels are generated); it matters only for bit-level reproduction of a reference run on
real JPEGs.
"""
import os

import numpy as np
import numpy.random as npr

from detectron.core.config import cfg
import detectron.roi_data.wsl as wsl_roi_data


def get_minibatch_blob_names(is_training=True):
    """Order in which the loader emits blobs (and the net dequeues them)."""
    return ['data', 'data_ids'] + wsl_roi_data.get_wsl_blob_names(is_training=is_training)


def get_minibatch(roidb, raw=None):
    """raw=True (default: cfg.NAWS.DEVICE_PREP): `data` is a placeholder and blobs['_raw'] holds
    one dict per image (uint8 pixels, flip, crop, scale) for the device-side preparation."""
    raw = cfg.NAWS.DEVICE_PREP if raw is None else raw
    blobs = {k: [] for k in get_minibatch_blob_names()}
    im_blob, im_scales, im_crops = _get_image_blob(roidb, raw=raw)
    raws = im_blob if raw else None
    if raw:
        im_blob = np.zeros((len(roidb), 3, 1, 1), np.float32)
    # crops are (y1,x1,y2,x2) -> (x1,y1,x2,y2)
    im_crops = np.array(im_crops, dtype=np.int32)[:, (1, 0, 3, 2)]
    blobs['data'] = im_blob
    blobs['data_ids'] = _get_image_id_blob(roidb)
    valid = wsl_roi_data.add_wsl_blobs(blobs, im_scales, im_crops, roidb)
    if raw:
        blobs['_raw'] = raws
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


def _hsv_tables():
    i = np.arange(256, dtype=np.float64)
    i[0] = 1.0
    sdiv = np.rint((255 << 12) / i).astype(np.int64)
    hdiv = np.rint((180 << 12) / (6.0 * i)).astype(np.int64)
    sdiv[0] = hdiv[0] = 0
    return sdiv, hdiv


def distort_hsv(im, s0, s1):
    """The saturation / exposure jitter of minibatch_wsl.py:127-138 on a uint8 BGR image:
    cv2 8-bit BGR->HSV (fixed-point, H in [0,180)), S = min(s0*S, 255), V = min(s1*V, 255) in
    float32, truncated to uint8, cv2 8-bit HSV->BGR (float sector formula, rounded)."""
    sdiv, hdiv = _hsv_tables()
    x = im.astype(np.int64)
    b, g, r = x[..., 0], x[..., 1], x[..., 2]
    v = np.maximum(np.maximum(b, g), r)
    diff = v - np.minimum(np.minimum(b, g), r)
    s = (diff * sdiv[v] + 2048) >> 12
    h = np.where(v == r, g - b, np.where(v == g, b - r + 2 * diff, r - g + 4 * diff))
    h = (h * hdiv[diff] + 2048) >> 12
    h = np.where(h < 0, h + 180, h)
    f32 = np.float32
    s = np.minimum(f32(s0) * s.astype(f32), f32(255)).astype(np.uint8).astype(f32) * f32(1.0 / 255.0)
    v = np.minimum(f32(s1) * v.astype(f32), f32(255)).astype(np.uint8).astype(f32) * f32(1.0 / 255.0)
    hf = h.astype(f32) * f32(6.0 / 180.0)
    hf = np.where(hf >= 6, hf - f32(6), hf).astype(f32)
    sec = np.floor(hf).astype(np.int64)
    fr = hf - sec.astype(f32)
    bad = (sec < 0) | (sec >= 6)
    sec, fr = np.where(bad, 0, sec), np.where(bad, f32(0), fr).astype(f32)
    one = f32(1)
    tab = np.stack([v, v * (one - s), v * (one - s * fr), v * (one - s * (one - fr))], -1)
    slot = np.array([[1, 3, 0], [1, 0, 2], [3, 0, 1], [0, 2, 1], [0, 1, 3], [2, 1, 0]])[sec]
    bgr = np.take_along_axis(tab, slot, -1)
    bgr = np.where((s == 0)[..., None], v[..., None], bgr).astype(f32)
    return np.clip(np.rint(bgr * f32(255)), 0, 255).astype(np.uint8)


def resize_linear(im, im_scale):
    """cv2.resize(im, None, None, fx=im_scale, fy=im_scale, interpolation=cv2.INTER_LINEAR) on a
    float32 HxWxC image (blob.py:123-130): destination size cvRound(size*scale); source position
    float((d+0.5)/scale - 0.5); the two taps floor(pos), +1 with weights (1-frac, frac), clamped
    to the edge pixel with weight 1; rows first, then columns of rows."""
    im = np.asarray(im, np.float32)
    h, w = im.shape[:2]
    oh, ow = int(np.round(h * im_scale)), int(np.round(w * im_scale))

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) / float(im_scale) - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = f - s0.astype(np.float32)
        f[s0 < 0] = 0.0
        s0[s0 < 0] = 0
        f[s0 >= n_src - 1] = 0.0
        s0[s0 >= n_src - 1] = n_src - 1
        return s0, np.minimum(s0 + 1, n_src - 1), np.float32(1) - f, f
    sx, sx1, a0, a1 = taps(ow, w)
    sy, sy1, b0, b1 = taps(oh, h)
    rows = im[:, sx, :] * a0[None, :, None] + im[:, sx1, :] * a1[None, :, None]
    return (rows[sy] * b0[:, None, None] + rows[sy1] * b1[:, None, None]).astype(np.float32)


def resize_linear_xy(im, ow, oh):
    """cv2.resize(im, dsize=(ow, oh)) on a float32 image: per-axis scale = source / destination
    size (a double), otherwise the taps of resize_linear."""
    im = np.asarray(im, np.float32)
    h, w = im.shape[:2]

    def taps(n_dst, n_src):
        f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * (float(n_src) / float(n_dst)) - 0.5).astype(np.float32)
        s0 = np.floor(f).astype(np.int64)
        f = f - s0.astype(np.float32)
        f[s0 < 0] = 0.0
        s0[s0 < 0] = 0
        f[s0 >= n_src - 1] = 0.0
        s0[s0 >= n_src - 1] = n_src - 1
        return s0, np.minimum(s0 + 1, n_src - 1), np.float32(1) - f, f
    sx, sx1, a0, a1 = taps(ow, w)
    sy, sy1, b0, b1 = taps(oh, h)
    rows = im[:, sx, :] * a0[None, :, None] + im[:, sx1, :] * a1[None, :, None]
    return (rows[sy] * b0[:, None, None] + rows[sy1] * b1[:, None, None]).astype(np.float32)


def get_im_scale(shape_hw, target_size, max_size):
    size_min, size_max = min(shape_hw), max(shape_hw)
    im_scale = float(target_size) / float(size_min)
    if np.round(im_scale * size_max) > max_size:      # cap the long side (blob.py:119-122)
        im_scale = float(max_size) / float(size_max)
    return im_scale


def prep_im_for_blob(im, pixel_means, target_size, max_size):
    """float32, - PIXEL_MEANS, / PIXEL_STDS, scale the short side to target_size (long side
    capped at max_size) (blob.py:100-131)."""
    im = im.astype(np.float32, copy=False)
    im = im - np.asarray(pixel_means, np.float32).reshape(1, 1, 3)
    im = im / np.asarray(cfg.PIXEL_STDS, np.float32).reshape(1, 1, 3)
    im_scale = get_im_scale(im.shape[:2], target_size, max_size)
    return resize_linear(im, im_scale), im_scale


def im_list_to_blob(ims):
    """Zero-pad to the largest H,W and emit NCHW float32."""
    if not isinstance(ims, list):
        ims = [ims]
    max_shape = np.array([im.shape for im in ims]).max(axis=0)
    blob = np.zeros((len(ims), max_shape[0], max_shape[1], 3), dtype=np.float32)
    for i, im in enumerate(ims):
        blob[i, :im.shape[0], :im.shape[1], :] = im
    return blob.transpose((0, 3, 1, 2)).copy()


def _get_image_blob(roidb, raw=False):
    scale_inds = npr.randint(0, high=len(cfg.TRAIN.SCALES), size=len(roidb))
    ims, scales, crops = [], [], []
    for i, entry in enumerate(roidb):
        im0 = _read_image(entry)
        im = im0[:, ::-1, :] if entry['flipped'] else im0
        distort = None
        if cfg.WSL.USE_DISTORTION:
            s0 = npr.random() * (cfg.WSL.SATURATION - 1) + 1
            s1 = npr.random() * (cfg.WSL.EXPOSURE - 1) + 1
            s0 = s0 if npr.random() > 0.5 else 1.0 / s0
            s1 = s1 if npr.random() > 0.5 else 1.0 / s1
            distort = (float(s0), float(s1))
            if not raw:
                im = distort_hsv(im, s0, s1)
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
        if raw:
            sc = get_im_scale(im.shape[:2], cfg.TRAIN.SCALES[scale_inds[i]], cfg.TRAIN.MAX_SIZE)
            ims.append(dict(im=np.ascontiguousarray(im0), flip=bool(entry['flipped']),
                            crop=tuple(int(v) for v in crop), scale=sc, distort=distort,
                            out_hw=(int(np.round(im.shape[0] * sc)), int(np.round(im.shape[1] * sc)))))
        else:
            im, sc = prep_im_for_blob(im, cfg.PIXEL_MEANS, cfg.TRAIN.SCALES[scale_inds[i]],
                                      cfg.TRAIN.MAX_SIZE)
            ims.append(im)
        scales.append(sc)
        crops.append(crop)
    return (ims if raw else im_list_to_blob(ims)), scales, crops
