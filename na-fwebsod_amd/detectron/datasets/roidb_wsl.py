"""Training roidb assembly.

Behaviour follows the reference's detectron/datasets/roidb_wsl.py: one roidb per
(dataset, proposal file) pair with ground truth and crowd filtering (:26-37), horizontally
mirrored duplicates when TRAIN.USE_FLIPPED (:61-93: x1' = W - x2 - 1, x2' = W - x1 - 1),
concatenation, and removal of images without at least one foreground (overlap >= FG_THRESH)
and one background (BG_THRESH_LO <= overlap < BG_THRESH_HI) box (:96-121).  Box-regression
targets (:124-161) are not produced: the WSL heads have no bbox_pred.
Checked against pairs captured from the reference functions (tests/test_datasets.py).
"""
import logging

import numpy as np

from detectron.core.config import cfg
from detectron.datasets.json_dataset_wsl import JsonDataset

logger = logging.getLogger(__name__)

_NOT_SHARED_WITH_MIRROR = frozenset(('boxes', 'segms', 'gt_keypoints', 'flipped'))


def _as_tuple(v):
    return (v,) if isinstance(v, str) else tuple(v)


def _mirrored(entry):
    """A shallow copy of `entry` describing the horizontally flipped image."""
    w = entry['width']
    src = entry['boxes']
    flipped = src.copy()
    flipped[:, 0], flipped[:, 2] = w - src[:, 2] - 1, w - src[:, 0] - 1
    assert (flipped[:, 2] >= flipped[:, 0]).all()
    twin = {k: v for k, v in entry.items() if k not in _NOT_SHARED_WITH_MIRROR}
    twin.update(boxes=flipped, segms=[], flipped=True)
    return twin


def extend_with_flipped_entries(roidb, dataset=None):
    """Append the mirrored twin of every entry, in order, after the originals."""
    roidb.extend([_mirrored(e) for e in list(roidb)])


def _usable_for_training(entry):
    ov = entry['max_overlaps']
    has_fg = bool(np.any(ov >= cfg.TRAIN.FG_THRESH))
    has_bg = bool(np.any((ov >= cfg.TRAIN.BG_THRESH_LO) & (ov < cfg.TRAIN.BG_THRESH_HI)))
    return has_fg and has_bg


def filter_for_training(roidb):
    kept = list(filter(_usable_for_training, roidb))
    logger.info('Filtered %d roidb entries: %d -> %d', len(roidb) - len(kept), len(roidb), len(kept))
    return kept


def combined_roidb_for_training(dataset_names, proposal_files):
    names, files = _as_tuple(dataset_names), _as_tuple(proposal_files)
    if not files:
        files = (None,) * len(names)
    assert len(names) == len(files)
    merged = []
    for name, pfile in zip(names, files):
        ds = JsonDataset(name)
        part = ds.get_roidb(gt=True, proposal_file=pfile,
                            crowd_filter_thresh=cfg.TRAIN.CROWD_FILTER_THRESH)
        if cfg.TRAIN.USE_FLIPPED:
            logger.info('Appending horizontally-flipped training examples...')
            extend_with_flipped_entries(part, ds)
        logger.info('Loaded dataset: %s', ds.name)
        merged += part
    return filter_for_training(merged)
