"""Seeded synthetic inputs shared by the parity tests (SURVEY.md §8d shapes, scaled down)."""
import numpy as np


def make_rois(rng, n_img, r_per_img, height, width, fractional=True, degenerate=True):
    """[R,5] (batch, x1, y1, x2, y2) in input-image pixels, grouped by image."""
    out = []
    for b in range(n_img):
        w = np.exp(rng.uniform(np.log(21), np.log(width), r_per_img))
        h = np.exp(rng.uniform(np.log(21), np.log(height), r_per_img))
        x1 = np.floor(rng.uniform(0, width - 1, r_per_img))
        y1 = np.floor(rng.uniform(0, height - 1, r_per_img))
        x2 = np.minimum(x1 + w, width - 1)
        y2 = np.minimum(y1 + h, height - 1)
        r = np.stack([np.full(r_per_img, b), x1, y1, x2, y2], 1).astype(np.float32)
        if fractional:  # rois carry fractional coords after x im_scale (e.g. 148.5)
            r[:, 1:] = r[:, 1:] * np.float32(1.171875)
            r[::7, 1:] = np.floor(r[::7, 1:]) + 0.5
        if degenerate and r_per_img >= 8:
            r[1, 1:] = [5, 5, 5, 5]                       # 1-px
            r[2, 1:] = [30, 40, 10, 20]                   # malformed (x2<x1)
            r[3, 1:] = [width * 2, height * 2, width * 2 + 50, height * 2 + 50]  # outside
            r[4, 1:] = [-40, -30, -5, -2]                 # negative
            r[5, 1:] = [0, 0, width * 1.171875 * 2, height * 1.171875 * 2]       # covers all
        out.append(r)
    return np.concatenate(out, 0)


def seg_offsets(rois):
    b = rois[:, 0].astype(np.int64)
    n = int(b.max()) + 1
    counts = np.bincount(b, minlength=n)
    return np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
