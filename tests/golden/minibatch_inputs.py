"""Seeded inputs of the whole-minibatch capture (make_golden_minibatch.py) and of the test that
replays it: a three-image roidb and the pixels `imread` returns.  Needs nothing of the reference."""
import os

import numpy as np

SIZES = {'img_000017.jpg': (333, 500), 'img_000204.jpg': (480, 360), 'web_cat_12.jpg': (375, 499)}


def make_roidb(seed=21):
    rng = np.random.RandomState(seed)
    roidb = []
    for name, flipped, cls in (('img_000017.jpg', False, [7]), ('img_000204.jpg', True, [3, 15]),
                               ('web_cat_12.jpg', False, [20])):
        h, w = SIZES[name]
        n = 37
        x1 = np.floor(rng.uniform(0, w - 30, n))
        y1 = np.floor(rng.uniform(0, h - 30, n))
        x2 = np.minimum(x1 + np.floor(rng.uniform(21, w, n)), w - 1)
        y2 = np.minimum(y1 + np.floor(rng.uniform(21, h, n)), h - 1)
        boxes = np.stack([x1, y1, x2, y2], 1).astype(np.float32)
        gt = np.zeros((n,), np.int32)
        gt[:len(cls)] = cls
        roidb.append(dict(image='/data/' + name, flipped=flipped, height=h, width=w, boxes=boxes,
                          obn_scores=np.sort(rng.uniform(0, 1, (n, 1)).astype(np.float32), 0)[::-1].copy(),
                          gt_classes=gt))
    return roidb


def fake_image(path):
    name = os.path.basename(path)
    h, w = SIZES[name]
    return np.random.RandomState(sum(map(ord, name))).randint(0, 256, (h, w, 3)).astype(np.uint8)
