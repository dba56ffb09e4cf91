"""Image transforms of the test-time augmentations (reference: detectron/utils/image.py:27-32
`aspect_ratio_rel`, used by core/test_wsl.py:330-352).

cv2 is absent from the MI355X image (and from the build container): `cv2.resize` with its
default INTER_LINEAR is restated from OpenCV's published 8-bit algorithm (imgproc resize.cpp:
11-bit fixed-point tap weights, `cvRound(w * 2048)`, the row pass in int, the column pass
`((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2`).  OpenCV is an un-vendored,
un-pinned dependency of the reference: PARITY UNPINNED for the pixel values (the shapes and the
box arithmetic around this function are pinned by tests/golden/reference_tta.npz)."""
import numpy as np

_COEF_BITS = 11
_ONE = 1 << _COEF_BITS


def _taps(n_dst, n_src):
    """Source index pairs and fixed-point weights of one axis (resize.cpp: the position is the
    float (d + 0.5) * scale - 0.5 with scale = n_src / n_dst as a double; floor / fraction;
    clamped to the edge pixel with the whole weight)."""
    scale = float(n_src) / float(n_dst)
    f = ((np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    s0 = np.floor(f).astype(np.int64)
    f = f - s0.astype(np.float32)
    f[s0 < 0] = 0.0
    s0[s0 < 0] = 0
    f[s0 >= n_src - 1] = 0.0
    s0[s0 >= n_src - 1] = n_src - 1
    a1 = np.rint(f * np.float32(_ONE)).astype(np.int64)               # saturate_cast<short>
    a0 = np.rint((np.float32(1) - f) * np.float32(_ONE)).astype(np.int64)
    return s0, np.minimum(s0 + 1, n_src - 1), a0, a1


def resize_linear_u8(im, dsize):
    """cv2.resize(im, dsize=(width, height)) for a uint8 HxWxC image, INTER_LINEAR."""
    im = np.asarray(im)
    assert im.dtype == np.uint8 and im.ndim == 3
    ow, oh = int(dsize[0]), int(dsize[1])
    h, w = im.shape[:2]
    sx, sx1, a0, a1 = _taps(ow, w)
    sy, sy1, b0, b1 = _taps(oh, h)
    x = im.astype(np.int64)
    rows = x[:, sx, :] * a0[None, :, None] + x[:, sx1, :] * a1[None, :, None]     # x 2^11
    out = (((b0[:, None, None] * (rows[sy] >> 4)) >> 16) +
           ((b1[:, None, None] * (rows[sy1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def aspect_ratio_rel(im, aspect_ratio):
    """Width-relative aspect-ratio transformation (image.py:27-32): the width becomes
    int(round(aspect_ratio * width)), the height stays."""
    im_h, im_w = im.shape[:2]
    im_ar_w = int(round(aspect_ratio * im_w))
    if im.dtype == np.uint8:
        return resize_linear_u8(im, (im_ar_w, im_h))
    # float images (not what the test loop passes): the float tap arithmetic of the loader
    from detectron.roi_data.minibatch_wsl import resize_linear_xy
    return resize_linear_xy(np.asarray(im, np.float32), im_ar_w, im_h)
