"""roidb assembly for training (reference: detectron/datasets/roidb_wsl.py:21-58
combined_roidb_for_training, :61-93 extend_with_flipped_entries, :96-121 filter_for_training).
Bounding-box regression targets (:124-161) belong to the Fast R-CNN box head, which the WSL
path does not have (no bbox_pred blob): not restated."""
import logging

import numpy as np

from detectron.core.config import cfg
from detectron.datasets.json_dataset_wsl import JsonDataset

logger = logging.getLogger(__name__)


def combined_roidb_for_training(dataset_names, proposal_files):
    def get_roidb(dataset_name, proposal_file):
        ds = JsonDataset(dataset_name)
        roidb = ds.get_roidb(gt=True, proposal_file=proposal_file,
                             crowd_filter_thresh=cfg.TRAIN.CROWD_FILTER_THRESH)
        if cfg.TRAIN.USE_FLIPPED:
            logger.info('Appending horizontally-flipped training examples...')
            extend_with_flipped_entries(roidb, ds)
        logger.info('Loaded dataset: {:s}'.format(ds.name))
        return roidb

    if isinstance(dataset_names, str):
        dataset_names = (dataset_names, )
    if isinstance(proposal_files, str):
        proposal_files = (proposal_files, )
    if len(proposal_files) == 0:
        proposal_files = (None, ) * len(dataset_names)
    assert len(dataset_names) == len(proposal_files)
    roidbs = [get_roidb(*args) for args in zip(dataset_names, proposal_files)]
    roidb = roidbs[0]
    for r in roidbs[1:]:
        roidb.extend(r)
    return filter_for_training(roidb)


def extend_with_flipped_entries(roidb, dataset):
    flipped_roidb = []
    for entry in roidb:
        width = entry['width']
        boxes = entry['boxes'].copy()
        oldx1 = boxes[:, 0].copy()
        oldx2 = boxes[:, 2].copy()
        boxes[:, 0] = width - oldx2 - 1
        boxes[:, 2] = width - oldx1 - 1
        assert (boxes[:, 2] >= boxes[:, 0]).all()
        flipped_entry = {k: v for k, v in entry.items()
                         if k not in ('boxes', 'segms', 'gt_keypoints', 'flipped')}
        flipped_entry['boxes'] = boxes
        flipped_entry['segms'] = []
        flipped_entry['flipped'] = True
        flipped_roidb.append(flipped_entry)
    roidb.extend(flipped_roidb)


def filter_for_training(roidb):
    def is_valid(entry):
        overlaps = entry['max_overlaps']
        fg_inds = np.where(overlaps >= cfg.TRAIN.FG_THRESH)[0]
        bg_inds = np.where((overlaps < cfg.TRAIN.BG_THRESH_HI) &
                           (overlaps >= cfg.TRAIN.BG_THRESH_LO))[0]
        return len(fg_inds) > 0 and len(bg_inds) > 0

    num = len(roidb)
    filtered = [entry for entry in roidb if is_valid(entry)]
    logger.info('Filtered {} roidb entries: {} -> {}'.format(num - len(filtered), num,
                                                             len(filtered)))
    return filtered
