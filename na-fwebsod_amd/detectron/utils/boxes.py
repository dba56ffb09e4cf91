"""Box helpers of the data path (reference: detectron/utils/boxes.py:66-131, and
detectron/utils/cython_bbox.pyx `bbox_overlaps` restated in numpy: +1 pixel areas)."""
import numpy as np


def unique_boxes(boxes, scale=1.0):
    """Indices of unique boxes (first occurrence), ascending (boxes.py:66-71)."""
    v = np.array([1, 1e3, 1e6, 1e9])
    hashes = np.round(boxes * scale).dot(v).astype(np.int64)
    _, index = np.unique(hashes, return_index=True)
    return np.sort(index)


def filter_small_boxes(boxes, min_size):
    """Keep boxes with width and height both greater than min_size (boxes.py:108-113)."""
    w = boxes[:, 2] - boxes[:, 0] + 1
    h = boxes[:, 3] - boxes[:, 1] + 1
    return np.where((w > min_size) & (h > min_size))[0]


def xywh_to_xyxy(xywh):
    if isinstance(xywh, (list, tuple)):
        assert len(xywh) == 4
        x1, y1 = xywh[0], xywh[1]
        return (x1, y1, x1 + np.maximum(0., xywh[2] - 1.), y1 + np.maximum(0., xywh[3] - 1.))
    if isinstance(xywh, np.ndarray):
        return np.hstack((xywh[:, 0:2], xywh[:, 0:2] + np.maximum(0, xywh[:, 2:4] - 1)))
    raise TypeError('Argument xywh must be a list, tuple, or numpy array.')


def xyxy_to_xywh(xyxy):
    if isinstance(xyxy, (list, tuple)):
        assert len(xyxy) == 4
        return (xyxy[0], xyxy[1], xyxy[2] - xyxy[0] + 1, xyxy[3] - xyxy[1] + 1)
    if isinstance(xyxy, np.ndarray):
        return np.hstack((xyxy[:, 0:2], xyxy[:, 2:4] - xyxy[:, 0:2] + 1))
    raise TypeError('Argument xyxy must be a list, tuple, or numpy array.')


def clip_xyxy_to_image(x1, y1, x2, y2, height, width):
    x1 = np.minimum(width - 1., np.maximum(0., x1))
    y1 = np.minimum(height - 1., np.maximum(0., y1))
    x2 = np.minimum(width - 1., np.maximum(0., x2))
    y2 = np.minimum(height - 1., np.maximum(0., y2))
    return x1, y1, x2, y2


def bbox_overlaps(boxes, query_boxes):
    """[N,K] IoU with +1 pixel areas (cython_bbox.pyx:27-63), bit-identical to the compiled
    reference (tests/test_datasets.py against oracle/_ref): coordinate differences are float32,
    the `+ 1` and the area products are evaluated in double (Cython emits the literal as 1.0),
    box_area / iw / ih are stored as float32, the union `float(...)` is a double expression
    stored as float32, and iw*ih/ua is float32."""
    b = np.asarray(boxes, np.float32)
    q = np.asarray(query_boxes, np.float32)
    out = np.zeros((b.shape[0], q.shape[0]), np.float32)
    if b.shape[0] == 0 or q.shape[0] == 0:
        return out
    f64 = np.float64
    qa = (((q[:, 2] - q[:, 0]).astype(f64) + 1.0) * ((q[:, 3] - q[:, 1]).astype(f64) + 1.0)).astype(np.float32)
    barea = ((b[:, 2] - b[:, 0]).astype(f64) + 1.0) * ((b[:, 3] - b[:, 1]).astype(f64) + 1.0)
    iw = ((np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0])).astype(f64)
          + 1.0).astype(np.float32)
    ih = ((np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1])).astype(f64)
          + 1.0).astype(np.float32)
    ok = (iw > 0) & (ih > 0)
    inter = (np.where(ok, iw, 0).astype(np.float32) * np.where(ok, ih, 0).astype(np.float32))
    ua = (barea[:, None] + qa.astype(f64)[None, :] - inter.astype(f64)).astype(np.float32)
    out[ok] = (inter / ua)[ok]
    return out


def crowd_iou(boxes_xywh, crowd_xywh):
    """pycocotools mask.iou(d, g, iscrowd=[1..]) for boxes: inter / area(d) (continuous
    coordinates, no +1) - the rule _filter_crowd_proposals uses (json_dataset_wsl.py:703-720)."""
    d = np.asarray(boxes_xywh, np.float64)
    g = np.asarray(crowd_xywh, np.float64)
    iw = np.minimum(d[:, None, 0] + d[:, None, 2], g[None, :, 0] + g[None, :, 2]) - \
        np.maximum(d[:, None, 0], g[None, :, 0])
    ih = np.minimum(d[:, None, 1] + d[:, None, 3], g[None, :, 1] + g[None, :, 3]) - \
        np.maximum(d[:, None, 1], g[None, :, 1])
    inter = np.maximum(iw, 0) * np.maximum(ih, 0)
    area = (d[:, 2] * d[:, 3])[:, None]
    return np.where(area > 0, inter / np.maximum(area, 1e-30), 0.0)


def flip_boxes(boxes, im_width):
    """Horizontal flip of [n, 4k] boxes (boxes.py:245-251)."""
    boxes_flipped = boxes.copy()
    boxes_flipped[:, 0::4] = im_width - boxes[:, 2::4] - 1
    boxes_flipped[:, 2::4] = im_width - boxes[:, 0::4] - 1
    return boxes_flipped


def aspect_ratio(boxes, aspect_ratio):
    """Width-relative aspect-ratio transform of [n, 4k] boxes (boxes.py:254-259)."""
    boxes_ar = boxes.copy()
    boxes_ar[:, 0::4] = aspect_ratio * boxes[:, 0::4]
    boxes_ar[:, 2::4] = aspect_ratio * boxes[:, 2::4]
    return boxes_ar


def box_voting(top_dets, all_dets, thresh, scoring_method='ID', beta=1.0):
    """Bounding-box voting (boxes.py:262-318; arXiv 1505.01749): every row of top_dets [n,5]
    becomes the score-weighted average of the all_dets boxes that overlap it by IoU >= thresh;
    the score is left ('ID') or recombined ('TEMP_AVG', 'AVG', 'IOU_AVG', 'GENERALIZED_AVG',
    'QUASI_SUM').  Same numpy calls in the same order as the reference, on top of
    `bbox_overlaps` above: pinned by tests/golden/reference_tta.npz (vote_*)."""
    out = top_dets.copy()
    all_boxes, all_scores = all_dets[:, :4], all_dets[:, 4]
    overlaps = bbox_overlaps(top_dets[:, :4], all_boxes)
    for k in range(out.shape[0]):
        inds = np.where(overlaps[k] >= thresh)[0]
        ws = all_scores[inds]
        out[k, :4] = np.average(all_boxes[inds, :], axis=0, weights=ws)
        if scoring_method == 'ID':
            pass
        elif scoring_method == 'TEMP_AVG':
            # P(class) vs P(not class), softened by the temperature beta, averaged
            P = np.vstack((ws, 1.0 - ws))
            X = np.log(P / np.max(P, axis=0))
            X_exp = np.exp(X / beta)
            out[k, 4] = (X_exp / np.sum(X_exp, axis=0))[0].mean()
        elif scoring_method == 'AVG':
            out[k, 4] = ws.mean()
        elif scoring_method == 'IOU_AVG':
            out[k, 4] = np.average(ws, weights=overlaps[k, inds])
        elif scoring_method == 'GENERALIZED_AVG':
            out[k, 4] = np.mean(ws ** beta) ** (1.0 / beta)
        elif scoring_method == 'QUASI_SUM':
            out[k, 4] = ws.sum() / float(len(ws)) ** beta
        else:
            raise NotImplementedError('Unknown scoring method {}'.format(scoring_method))
    return out


SOFT_NMS_METHODS = {'hard': 0, 'linear': 1, 'gaussian': 2}


def soft_nms(dets, sigma=0.5, overlap_thresh=0.3, score_thresh=0.001, method='linear'):
    """Soft-NMS (boxes.py:321-338 -> cython_nms.pyx:98-196; arXiv 1704.04503) on [n,5] float32
    detections -> (dets_out [m,5] with decayed scores, keep [m] = their original indices), in the
    order the reference emits them.

    The reference is a sequential in-place loop: position i receives the highest-scoring box of
    positions i..N-1 (first maximum wins), every later box that overlaps it has its score
    multiplied by a weight (linear: 1 - IoU above the threshold; gaussian: exp(-IoU^2 / sigma);
    hard: 0 above the threshold), and a box whose score falls under score_thresh is overwritten
    by the LAST box, shrinking N.  Here the decay of one round is one vector expression (float32
    arithmetic, the same expression order); the overwrite-by-the-last-box compaction, which
    fixes the output order and later tie-breaks, is replayed for the (few) failing boxes only."""
    assert method in SOFT_NMS_METHODS, 'Unknown soft_nms method: {}'.format(method)
    if dets.shape[0] == 0:
        return dets, []
    f = np.float32
    b = np.ascontiguousarray(dets, dtype=np.float32).copy()
    sigma, nt, thr = f(sigma), f(overlap_thresh), f(score_thresh)
    m = SOFT_NMS_METHODS[method]
    n = b.shape[0]
    inds = np.arange(n)
    one = f(1)
    i = 0
    while i < n:
        # the maximum of positions i..n-1 (strict <: the first one) swaps into position i
        mp = i + int(np.argmax(b[i:n, 4]))
        if mp != i:
            b[[i, mp]] = b[[mp, i]]
            inds[[i, mp]] = inds[[mp, i]]
        if i + 1 < n:
            t = b[i]
            r = b[i + 1:n]
            area = (r[:, 2] - r[:, 0] + one) * (r[:, 3] - r[:, 1] + one)
            iw = np.minimum(t[2], r[:, 2]) - np.maximum(t[0], r[:, 0]) + one
            ih = np.minimum(t[3], r[:, 3]) - np.maximum(t[1], r[:, 1]) + one
            hit = (iw > 0) & (ih > 0)
            with np.errstate(divide='ignore', invalid='ignore'):
                ua = ((t[2] - t[0] + one) * (t[3] - t[1] + one) + area - iw * ih).astype(f)
                ov = (iw * ih / ua).astype(f)
                if m == 1:
                    w = np.where(ov > nt, one - ov, one)
                elif m == 2:
                    # np.exp of the float32 quotient, evaluated in double, stored as float
                    w = np.exp((-(ov * ov) / sigma).astype(np.float64)).astype(f)
                else:
                    w = np.where(ov > nt, f(0), one)
                new = (w.astype(f) * r[:, 4]).astype(f)
            r[hit, 4] = new[hit]
            fail = hit & (r[:, 4] < thr)
            if fail.any():
                # replay "overwrite by the last box" for the failing positions, ascending; the
                # boxes taken from the end have been decayed above exactly once, as in the loop
                hi = n - 1
                fl = np.zeros((n,), bool)
                fl[i + 1:n] = fail
                for p in (np.flatnonzero(fail) + i + 1):
                    if p > hi:
                        break
                    while hi > p and fl[hi]:
                        hi -= 1
                    if hi == p:
                        hi -= 1
                        break
                    b[p] = b[hi]
                    inds[p] = inds[hi]
                    fl[p] = False
                    hi -= 1
                n = hi + 1
        i += 1
    return b[:n], inds[:n].tolist()
