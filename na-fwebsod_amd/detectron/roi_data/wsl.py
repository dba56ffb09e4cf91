"""Per-image RoI blobs of the WSL minibatch (rois, obn_scores, labels_int32, labels_oh).
Mirrors detectron/roi_data/wsl.py:20-58 (blob names), :61-85 (add_wsl_blobs),
:87-166 (_sample_rois), :212-225 (_project_im_rois)."""
import numpy as np

from detectron.core.config import cfg


def get_wsl_blob_names(is_training=True):
    names = ['rois', 'obn_scores']
    if is_training:
        names += ['labels_int32', 'labels_oh']
    return names


def add_wsl_blobs(blobs, im_scales, im_crops, roidb):
    for im_i, entry in enumerate(roidb):
        for k, v in _sample_rois(entry, im_scales[im_i], im_crops[im_i], im_i).items():
            blobs[k].append(v)
    for k, v in blobs.items():
        if isinstance(v, list) and len(v) > 0:
            blobs[k] = np.concatenate(v)
    return True


def _sample_rois(roidb, im_scale, im_crop, batch_idx):
    """First min(BATCH_SIZE_PER_IM, n) proposals in file order (the reference's np.delete
    calls discard their result, wsl.py:114-115, so ground-truth rows are kept), projected into
    the crop/scale frame; obn_scores + 1; image-level labels from gt_classes."""
    n = int(min(int(cfg.TRAIN.BATCH_SIZE_PER_IM), roidb['boxes'].shape[0]))
    boxes = roidb['boxes'][:n].copy()
    scores = np.add(roidb['obn_scores'][:n].copy(), 1.0)
    rois = _project_im_rois(boxes, im_scale, im_crop)
    rois = np.hstack((batch_idx * np.ones((rois.shape[0], 1), dtype=np.float32), rois))
    gt = np.where(roidb['gt_classes'] > 0)[0]
    assert len(gt) > 0, 'Empty ground truth empty for image is not allowed. Please check.'
    labels_oh = np.zeros((1, cfg.MODEL.NUM_CLASSES - 1), dtype=np.float32)
    labels = np.zeros((1,), dtype=np.float32)
    for cls in roidb['gt_classes'][gt]:
        labels_oh[0, cls - 1] = 1
        labels[0] = cls - 1
    return dict(labels_int32=labels.astype(np.int32, copy=False),
                labels_oh=labels_oh,
                rois=rois.astype(np.float32, copy=False),
                obn_scores=scores)


def _project_im_rois(im_rois, im_scale_factor, im_crop):
    """Clip to the crop window (x1,y1,x2,y2), shift to its origin, scale.  In place on im_rois."""
    x0, y0, x1, y1 = im_crop[0], im_crop[1], im_crop[2], im_crop[3]
    im_rois[:, 0] = np.minimum(np.maximum(im_rois[:, 0], x0), x1)
    im_rois[:, 1] = np.minimum(np.maximum(im_rois[:, 1], y0), y1)
    im_rois[:, 2] = np.maximum(np.minimum(im_rois[:, 2], x1), x0)
    im_rois[:, 3] = np.maximum(np.minimum(im_rois[:, 3], y1), y0)
    origin = np.tile(im_crop[:2], [im_rois.shape[0], 2])
    return (im_rois - origin) * im_scale_factor
