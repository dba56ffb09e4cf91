"""Inference.  Mirrors detectron/core/test_wsl.py `im_detect_bbox` (:102-178): scale the
image, project the proposals, de-duplicate them on the DEDUP_BOXES grid (hash :125-133),
one forward pass, scatter the scores back to the original proposal order."""
import numpy as np
import torch

from detectron.core.config import cfg
from detectron.roi_data.minibatch_wsl import im_list_to_blob, prep_im_for_blob


def dedup_rois(rois, dedup_boxes):
    """-> (unique rois, index, inv_index) with the reference hash
    round(rois * DEDUP_BOXES) . [1, 1e3, 1e6, 1e9, 1e12]."""
    v = np.array([1, 1e3, 1e6, 1e9, 1e12])
    hashes = np.round(rois * dedup_boxes).dot(v)
    _, index, inv_index = np.unique(hashes, return_index=True, return_inverse=True)
    return rois[index, :], index, inv_index


def im_detect_bbox(executor, im, target_scale, target_max_size, boxes, obn_scores):
    """im: HxWx3 BGR float/uint8; boxes [n,4] in image pixels -> (scores [n, C+1], boxes)."""
    blob_im, im_scale = prep_im_for_blob(im, cfg.PIXEL_MEANS, target_scale, target_max_size)
    data = im_list_to_blob([blob_im])
    rois = np.hstack((np.zeros((boxes.shape[0], 1), np.float32), boxes * im_scale)).astype(np.float32)
    obn = (obn_scores + 1.0).astype(np.float32)
    inv_index = None
    if cfg.DEDUP_BOXES > 0:
        rois, index, inv_index = dedup_rois(rois, cfg.DEDUP_BOXES)
        obn = obn[index, :]
    dev = executor.device
    executor.feed(dict(data=torch.from_numpy(data).to(dev), rois=torch.from_numpy(rois).to(dev),
                       obn_scores=torch.from_numpy(obn).to(dev)))
    executor.run()
    scores = executor.fetch('cls_prob').cpu().numpy()
    scores = scores.reshape([-1, scores.shape[-1]])
    if inv_index is not None:
        scores = scores[inv_index, :]
    return scores, boxes
