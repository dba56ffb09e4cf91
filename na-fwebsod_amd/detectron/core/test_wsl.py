"""Inference.  Mirrors detectron/core/test_wsl.py `im_detect_bbox` (:102-178): scale the
image, project the proposals, de-duplicate them on the DEDUP_BOXES grid (hash :125-133),
one forward pass, scatter the scores back to the original proposal order."""
import numpy as np
import torch

from detectron.core.config import cfg
from detectron.roi_data.minibatch_wsl import get_im_scale, im_list_to_blob, prep_im_for_blob
from detectron.utils import boxes as box_utils
from detectron.utils import image as image_utils


def dedup_rois(rois, dedup_boxes):
    """-> (unique rois, index, inv_index) with the reference hash
    round(rois * DEDUP_BOXES) . [1, 1e3, 1e6, 1e9, 1e12]."""
    v = np.array([1, 1e3, 1e6, 1e9, 1e12])
    hashes = np.round(rois * dedup_boxes).dot(v)
    _, index, inv_index = np.unique(hashes, return_index=True, return_inverse=True)
    return rois[index, :], index, inv_index


def project_rois(boxes, im_scale):
    """[n,4] image boxes -> the [n,5] rois blob: the product is taken in float64 and the blob cast
    to float32 afterwards, as the reference does (core/test_wsl.py:998-1026: `im_rois.astype(np.float)
    * scales`, then `astype(np.float32)`) - a float32 product can differ in the last bit, which
    moves a RoIPool bin edge when a coordinate sits on a rounding boundary."""
    rois = np.asarray(boxes).astype(np.float64) * im_scale
    return np.hstack((np.zeros((rois.shape[0], 1)), rois)).astype(np.float32)


def _device_image_blob(dev, im, im_scale, flip):
    """The [1,3,H',W'] input blob prepared on the GPU (naws_prep_image_fwd: mean/std, optional
    flip, cv2-semantics bilinear resize) when the image holds 8-bit pixel values; None otherwise."""
    if not cfg.NAWS.DEVICE_PREP:
        return None
    u8 = im if im.dtype == np.uint8 else im.astype(np.uint8)
    if im.dtype != np.uint8 and not np.array_equal(u8, im):
        return None
    from naws_hip import ops
    h, w = im.shape[:2]
    oh, ow = int(np.round(h * im_scale)), int(np.round(w * im_scale))
    data = torch.zeros((1, 3, oh, ow), device=dev, dtype=torch.float32)
    ops.prep_image(torch.from_numpy(np.ascontiguousarray(u8)).to(dev), data[0], im_scale, flip=flip,
                   means=cfg.PIXEL_MEANS.reshape(-1)[:3], stds=np.asarray(cfg.PIXEL_STDS).reshape(-1)[:3])
    return data


def im_detect_bbox(executor, im, target_scale, target_max_size, boxes, obn_scores, flip=False):
    """im: HxWx3 BGR float/uint8; boxes [n,4] in image pixels (already mirrored when `flip`)
    -> (scores [n, K], pred_boxes [n, 4K]: with TEST.BBOX_REG False the proposals repeated once
    per class, core/test_wsl.py:166-168).  `flip` mirrors the image horizontally."""
    dev = executor.device
    im_scale = get_im_scale(im.shape[:2], target_scale, target_max_size)
    data = _device_image_blob(dev, im, im_scale, flip)
    if data is None:
        imh = np.ascontiguousarray(im[:, ::-1, :]) if flip else im
        blob_im, im_scale = prep_im_for_blob(imh, cfg.PIXEL_MEANS, target_scale, target_max_size)
        data = torch.from_numpy(im_list_to_blob([blob_im])).to(dev)
    rois = project_rois(boxes, im_scale)
    obn = (obn_scores + 1.0).astype(np.float32)
    inv_index = None
    if cfg.DEDUP_BOXES > 0:
        rois, index, inv_index = dedup_rois(rois, cfg.DEDUP_BOXES)
        obn = obn[index, :]
    executor.feed(dict(data=data, rois=torch.from_numpy(rois).to(dev),
                       obn_scores=torch.from_numpy(obn).to(dev), _seg=[0, rois.shape[0]]))
    executor.run()
    scores = executor.fetch('cls_prob').cpu().numpy()
    scores = scores.reshape([-1, scores.shape[-1]])
    if inv_index is not None:
        scores = scores[inv_index, :]
    return scores, np.tile(boxes, (1, scores.shape[1]))


# ---------------------------------------------------------------------------------------
# Test-time augmentation and post-processing (reference: core/test_wsl.py:29-99 im_detect_all,
# :181-281 im_detect_bbox_aug, :284-352 hflip / scale variants, :803-863
# box_results_with_nms_and_limit).  Host logic over the same forward pass; the per-class NMS of
# an image runs on the GPU in one launch pair (naws_nms_sorted_fwd, same arithmetic as the
# reference's cython loop), with the numpy `nms` below as the host form.
# ---------------------------------------------------------------------------------------
flip_boxes = box_utils.flip_boxes


def im_detect_bbox_hflip(executor, im, target_scale, target_max_size, boxes, obn_scores):
    """Detection on the mirrored image (:284-310): the proposals are mirrored, and the returned
    boxes are the mirrored predictions mirrored back (in float32: equal to the proposals for
    integer-valued boxes, which is what the 'ID' coordinate heuristic asserts)."""
    im_width = im.shape[1]
    scores_hf, boxes_hf = im_detect_bbox(executor, im, target_scale, target_max_size,
                                         flip_boxes(boxes, im_width), obn_scores, flip=True)
    return scores_hf, flip_boxes(boxes_hf, im_width)


def im_detect_bbox_scale(executor, im, target_scale, target_max_size, boxes, obn_scores,
                         hflip=False):
    fn = im_detect_bbox_hflip if hflip else im_detect_bbox
    return fn(executor, im, target_scale, target_max_size, boxes, obn_scores)


def im_detect_bbox_aspect_ratio(executor, im, aspect_ratio, boxes, obn_scores, hflip=False):
    """Detection at a width-relative aspect ratio (:330-363): image and proposals are stretched,
    the predictions are mapped back with 1 / aspect_ratio."""
    im_ar = image_utils.aspect_ratio_rel(im, aspect_ratio)
    boxes_ar = box_utils.aspect_ratio(boxes, aspect_ratio)
    fn = im_detect_bbox_hflip if hflip else im_detect_bbox
    scores_ar, pred_ar = fn(executor, im_ar, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, boxes_ar, obn_scores)
    return scores_ar, box_utils.aspect_ratio(pred_ar, 1.0 / aspect_ratio)


def im_detect_bbox_pair(executor, im, target_scale, target_max_size, boxes, obn_scores):
    """The plain and the horizontally mirrored pass of one scale as ONE forward pass over a batch
    of two images (proposal softmax and ReduceSum are per image: engine segments), so the two conv
    bodies overlap on their streams and the head GEMMs run once with 2R rows.  Returns
    (scores, scores_hflip), each [n, C+1] in the order of `boxes`; the same values as two
    separate im_detect_bbox calls."""
    dev = executor.device
    im_scale = get_im_scale(im.shape[:2], target_scale, target_max_size)
    blobs = [_device_image_blob(dev, im, im_scale, f) for f in (False, True)]
    if blobs[0] is None or getattr(executor, 'engine', None) is None:
        s0, _ = im_detect_bbox(executor, im, target_scale, target_max_size, boxes, obn_scores)
        s1, _ = im_detect_bbox_hflip(executor, im, target_scale, target_max_size, boxes, obn_scores)
        return s0, s1
    rois, obns, invs, seg = [], [], [], [0]
    for b, bx in enumerate((boxes, flip_boxes(boxes, im.shape[1]))):
        r = project_rois(bx, im_scale)
        o = (obn_scores + 1.0).astype(np.float32)
        inv = None
        if cfg.DEDUP_BOXES > 0:
            r, index, inv = dedup_rois(r, cfg.DEDUP_BOXES)
            o = o[index, :]
        r[:, 0] = b
        rois.append(r); obns.append(o); invs.append(inv)
        seg.append(seg[-1] + r.shape[0])
    executor.feed(dict(data=torch.cat(blobs, 0), rois=torch.from_numpy(np.vstack(rois)).to(dev),
                       obn_scores=torch.from_numpy(np.vstack(obns)).to(dev), _seg=seg))
    executor.run()
    scores = executor.fetch('cls_prob').cpu().numpy()
    scores = scores.reshape([-1, scores.shape[-1]])
    out = []
    for b in range(2):
        sb = scores[seg[b]:seg[b + 1]]
        out.append(sb[invs[b], :] if invs[b] is not None else sb)
    return out[0], out[1]


def im_detect_bbox_aug(executor, im, boxes, obn_scores):
    """hflip at TEST.SCALE, each BBOX_AUG.SCALES (+flip), each BBOX_AUG.ASPECT_RATIOS (+flip),
    identity last (:181-281); scores combined by BBOX_AUG.SCORE_HEUR ('ID' | 'AVG' | 'UNION'),
    boxes by COORD_HEUR.  -> (scores_c, boxes_c [., 4K])."""
    aug = cfg.TEST.BBOX_AUG
    assert not aug.SCALE_SIZE_DEP, 'Size dependent scaling not implemented'
    assert (aug.SCORE_HEUR == 'UNION') == (aug.COORD_HEUR == 'UNION'), \
        'Score and coord heuristics must be UNION together'
    scores_ts, boxes_ts = [], []

    def add(s, b):
        scores_ts.append(s)
        boxes_ts.append(b)
        if aug.COORD_HEUR == 'ID':
            assert np.array_equal(boxes_ts[0], b), 'boxes at each scale should be the same'

    # a scale's plain + mirrored passes go through the network as one batch of two images
    # (cfg.NAWS.TTA_PAIR_FLIPS; same values, see im_detect_bbox_pair); results are added in the
    # reference's order: hflip at TEST.SCALE, each aug scale (+ its flip), identity last
    pair = bool(cfg.NAWS.TTA_PAIR_FLIPS)
    scores_i = None
    k = cfg.MODEL.NUM_CLASSES
    tiled = np.tile(boxes, (1, k))                               # what im_detect_bbox returns
    tiled_hf = flip_boxes(flip_boxes(tiled, im.shape[1]), im.shape[1])      # ... and _hflip
    if aug.H_FLIP:
        if pair:
            scores_i, s_hf = im_detect_bbox_pair(executor, im, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE,
                                                 boxes, obn_scores)
            add(s_hf, tiled_hf)
        else:
            add(*im_detect_bbox_hflip(executor, im, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, boxes,
                                      obn_scores))
    for scale in aug.SCALES:
        if pair and aug.SCALE_H_FLIP:
            s0, s1 = im_detect_bbox_pair(executor, im, scale, aug.MAX_SIZE, boxes, obn_scores)
            add(s0, tiled)
            add(s1, tiled_hf)
            continue
        add(*im_detect_bbox_scale(executor, im, scale, aug.MAX_SIZE, boxes, obn_scores))
        if aug.SCALE_H_FLIP:
            add(*im_detect_bbox_scale(executor, im, scale, aug.MAX_SIZE, boxes, obn_scores,
                                      hflip=True))
    for ar in aug.ASPECT_RATIOS:
        add(*im_detect_bbox_aspect_ratio(executor, im, ar, boxes, obn_scores))
        if aug.ASPECT_RATIO_H_FLIP:
            add(*im_detect_bbox_aspect_ratio(executor, im, ar, boxes, obn_scores, hflip=True))
    boxes_i = tiled
    if scores_i is None:
        scores_i, boxes_i = im_detect_bbox(executor, im, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, boxes,
                                           obn_scores)
    add(scores_i, boxes_i)
    if aug.SCORE_HEUR == 'ID':
        scores_c = scores_i
    elif aug.SCORE_HEUR == 'AVG':
        scores_c = np.mean(scores_ts, axis=0)
    elif aug.SCORE_HEUR == 'UNION':
        scores_c = np.vstack(scores_ts)
    else:
        raise NotImplementedError('Score heur {} not supported'.format(aug.SCORE_HEUR))
    if aug.COORD_HEUR == 'ID':
        boxes_c = boxes_i
    elif aug.COORD_HEUR == 'AVG':
        boxes_c = np.mean(boxes_ts, axis=0)
    elif aug.COORD_HEUR == 'UNION':
        boxes_c = np.vstack(boxes_ts)
    else:
        raise NotImplementedError('Coord heur {} not supported'.format(aug.COORD_HEUR))
    return scores_c, boxes_c


def nms(dets, thresh):
    """Greedy NMS on [n,5] (x1,y1,x2,y2,score); returns the kept indices in ascending order
    (cython_nms.pyx:36-87: a box is suppressed when IoU >= thresh; `np.where(suppressed == 0)`).
    Host (numpy) form, used when the detections are not on a GPU."""
    n = dets.shape[0]
    if n == 0:
        return []
    dets = np.asarray(dets, np.float32)
    x1, y1, x2, y2, sc = dets[:, 0], dets[:, 1], dets[:, 2], dets[:, 3], dets[:, 4]
    areas = (x2 - x1 + np.float32(1)) * (y2 - y1 + np.float32(1))
    order = np.argsort(-sc, kind='stable')
    suppressed = np.zeros((n,), bool)
    thresh = np.float32(thresh)
    for _i in range(n):
        i = order[_i]
        if suppressed[i]:
            continue
        rest = order[_i + 1:]
        xx1, yy1 = np.maximum(x1[i], x1[rest]), np.maximum(y1[i], y1[rest])
        xx2, yy2 = np.minimum(x2[i], x2[rest]), np.minimum(y2[i], y2[rest])
        inter = (np.maximum(np.float32(0), xx2 - xx1 + np.float32(1)) *
                 np.maximum(np.float32(0), yy2 - yy1 + np.float32(1)))
        ovr = inter / (areas[i] + areas[rest] - inter)
        suppressed[rest[ovr >= thresh]] = True
    return np.where(~suppressed)[0].tolist()


def nms_all_classes(scores, boxes):
    """cls -> kept row indices (ascending) for every foreground class of one image: all classes
    go through one HIP launch pair (naws_nms_sorted_fwd).  The numpy loop below is the reference's
    own host form (its NMS is a CPU cython routine) and runs only when cfg.NAWS.HOST_NMS asks for
    it - there is no silent fallback."""
    num_classes = cfg.MODEL.NUM_CLASSES
    import torch
    if not cfg.NAWS.HOST_NMS:
        from naws_hip import ops
        dev = torch.device('cuda', torch.cuda.current_device())
        sd = torch.as_tensor(np.ascontiguousarray(scores[:, 1:], np.float32), device=dev)
        if boxes.shape[1] > 4:
            # [n, 4K] class-tiled: without box regression (the only form this path has) every
            # class holds the same box, and the launch takes one box set for all classes
            assert np.array_equal(boxes[:, 4:8], boxes[:, -4:])
            boxes = boxes[:, 4:8]
        bd = torch.as_tensor(np.ascontiguousarray(boxes, np.float32), device=dev)
        keep = ops.nms_per_class(bd, sd, cfg.TEST.SCORE_THRESH, cfg.TEST.NMS).cpu().numpy()
        return {j: np.where(keep[j - 1])[0] for j in range(1, num_classes)}
    out = {}
    for j in range(1, num_classes):
        inds = np.where(scores[:, j] > cfg.TEST.SCORE_THRESH)[0]
        bj = boxes[inds, j * 4:(j + 1) * 4] if boxes.shape[1] > 4 else boxes[inds]
        dets = np.hstack((bj, scores[inds, j][:, np.newaxis])).astype(np.float32, copy=False)
        out[j] = inds[nms(dets, cfg.TEST.NMS)]
    return out


def soft_nms_all_classes(all_dets):
    """{class: dets [n_j, 5]} -> {class: soft-NMS'd dets} through naws_soft_nms_fwd (one workgroup
    per class, the reference's output order), or None when a class's list does not fit the
    kernel's LDS image (the numpy form then runs)."""
    from naws_hip import ops
    classes = sorted(all_dets)
    n_max = max([all_dets[j].shape[0] for j in classes] + [1])
    if n_max > ops.SOFT_NMS_MAX or not torch.cuda.is_available():
        return None
    dev = torch.device('cuda', torch.cuda.current_device())
    packed = np.zeros((len(classes), n_max, 5), np.float32)
    counts = np.zeros((len(classes),), np.int32)
    for k, j in enumerate(classes):
        packed[k, :all_dets[j].shape[0]] = all_dets[j]
        counts[k] = all_dets[j].shape[0]
    out, _keep, oc = ops.soft_nms_per_class(
        torch.from_numpy(packed).to(dev), torch.from_numpy(counts).to(dev), cfg.TEST.SOFT_NMS.SIGMA,
        cfg.TEST.NMS, 0.0001, box_utils.SOFT_NMS_METHODS[cfg.TEST.SOFT_NMS.METHOD])
    out, oc = out.cpu().numpy(), oc.cpu().numpy()
    return {j: out[k, :oc[k]].copy() for k, j in enumerate(classes)}


def box_results_with_nms_and_limit(scores, boxes):
    """Per-class score threshold, NMS (greedy, or TEST.SOFT_NMS), optional TEST.BBOX_VOTE
    refinement, then the DETECTIONS_PER_IM best over all classes (core/test_wsl.py:803-863).
    -> (scores, boxes, cls_boxes) with cls_boxes[j] = [n_j,5] for class j (0 = background)."""
    num_classes = cfg.MODEL.NUM_CLASSES
    cls_boxes = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    soft, vote = cfg.TEST.SOFT_NMS.ENABLED, cfg.TEST.BBOX_VOTE.ENABLED
    kept = None if soft else nms_all_classes(scores, boxes)

    def dets_of(j, inds):
        bj = boxes[inds, j * 4:(j + 1) * 4] if boxes.shape[1] > 4 else boxes[inds, :]
        return np.hstack((bj, scores[inds, j][:, np.newaxis])).astype(np.float32, copy=False)

    all_dets = {}
    if soft or vote:
        for j in range(1, num_classes):
            all_dets[j] = dets_of(j, np.where(scores[:, j] > cfg.TEST.SCORE_THRESH)[0])
    soft_dev = None
    if soft and not cfg.NAWS.HOST_NMS:
        soft_dev = soft_nms_all_classes(all_dets)        # every class in one launch (or None)
    for j in range(1, num_classes):
        dets_j = all_dets.get(j)
        if soft and soft_dev is not None:
            nms_dets = soft_dev[j]
        elif soft:
            # the overlap threshold is TEST.NMS, the discard threshold the reference's literal 1e-4
            nms_dets, _ = box_utils.soft_nms(dets_j, sigma=cfg.TEST.SOFT_NMS.SIGMA,
                                             overlap_thresh=cfg.TEST.NMS, score_thresh=0.0001,
                                             method=cfg.TEST.SOFT_NMS.METHOD)
        else:
            nms_dets = dets_of(j, kept[j])
        if vote and nms_dets.shape[0] > 0:
            nms_dets = box_utils.box_voting(nms_dets, dets_j, cfg.TEST.BBOX_VOTE.VOTE_TH,
                                            scoring_method=cfg.TEST.BBOX_VOTE.SCORING_METHOD)
        cls_boxes[j] = nms_dets
    if cfg.TEST.DETECTIONS_PER_IM > 0:
        all_scores = np.hstack([cls_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(all_scores) > cfg.TEST.DETECTIONS_PER_IM:
            th = np.sort(all_scores)[-cfg.TEST.DETECTIONS_PER_IM]
            for j in range(1, num_classes):
                cls_boxes[j] = cls_boxes[j][cls_boxes[j][:, -1] >= th, :]
    im_results = np.vstack([cls_boxes[j] for j in range(1, num_classes)])
    return im_results[:, -1], im_results[:, :-1], cls_boxes


def tta_passes():
    """[(target_scale, max_size, flip)] in the order the reference adds them (:181-281): hflip at
    TEST.SCALE, every BBOX_AUG scale (+ its flip), the identity pass last; one plain pass without
    BBOX_AUG."""
    aug = cfg.TEST.BBOX_AUG
    if not aug.ENABLED:
        return [(cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, False)]
    out = []
    if aug.H_FLIP:
        out.append((cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, True))
    for scale in aug.SCALES:
        out.append((scale, aug.MAX_SIZE, False))
        if aug.SCALE_H_FLIP:
            out.append((scale, aug.MAX_SIZE, True))
    out.append((cfg.TEST.SCALE, cfg.TEST.MAX_SIZE, False))
    return out


DEDUP_MAX_N = 16384                 # naws_roi_dedup_fwd: the bitonic sort's LDS image
DEDUP_MAX_HASH_COORD = 562.0        # |coord| * DEDUP_BOXES below this keeps the hash < 2^49


def device_post_supported(executor, im, n_proposals=None):
    """Whether im_detect_all_device can take this image: the cfg combination it implements, a
    proposal count its sort holds (0 < n <= 16384) and projected coordinates whose dedup hash
    fits the device key (include/naws.h, naws_roi_dedup_fwd); otherwise the numpy path runs."""
    aug = cfg.TEST.BBOX_AUG
    if not (cfg.NAWS.DEVICE_POST and cfg.NAWS.DEVICE_PREP and cfg.DEDUP_BOXES > 0
            and not cfg.NAWS.HOST_NMS and getattr(executor, 'engine', None) is not None
            and im.dtype == np.uint8
            and not cfg.TEST.SOFT_NMS.ENABLED and not cfg.TEST.BBOX_VOTE.ENABLED
            and (not aug.ENABLED or (aug.SCORE_HEUR in ('AVG', 'ID') and aug.COORD_HEUR == 'ID'
                                     and not aug.ASPECT_RATIOS and not aug.SCALE_SIZE_DEP))):
        return False
    if n_proposals is not None and not 0 < n_proposals <= DEDUP_MAX_N:
        return False
    h, w = im.shape[:2]
    top = max(get_im_scale((h, w), s, m) for s, m, _f in tta_passes()) * max(h, w)
    return top * cfg.DEDUP_BOXES < DEDUP_MAX_HASH_COORD


def im_detect_all_device(executor, im, box_proposals, obn_scores):
    """The whole per-image inference of `im_detect_all` with every intermediate in HBM
    (SURVEY.md 8 f-2): one upload of the pixels / proposals, per pass the device image
    preparation (naws_prep_image_fwd), roi projection + dedup hash + np.unique
    (naws_roi_dedup_fwd, all passes in one launch), the forward pass, the scatter-back fused with
    the running TTA sum (naws_tta_accumulate), then the mean, the per-class NMS
    (naws_nms_sorted_fwd) and the DETECTIONS_PER_IM cut (naws_det_limit_fwd).  Downloads: the
    per-pass unique counts (a few ints, they size the launches) and ONE packed result buffer.
    Bit-identical kept sets and scores to the numpy path (tests/test_gpu_infer_post.py)."""
    from naws_hip import ops
    dev = executor.device
    n = box_proposals.shape[0]
    num_classes = cfg.MODEL.NUM_CLASSES
    passes = tta_passes()
    avg = cfg.TEST.BBOX_AUG.ENABLED and cfg.TEST.BBOX_AUG.SCORE_HEUR == 'AVG'
    if cfg.TEST.BBOX_AUG.ENABLED and not avg:
        passes = passes[-1:]                       # 'ID': only the identity pass is used
    h, w = im.shape[:2]
    scales = [get_im_scale((h, w), s, m) for s, m, _f in passes]
    boxes_d = torch.from_numpy(np.ascontiguousarray(box_proposals, np.float32)).to(dev)
    obn_d = torch.from_numpy(np.ascontiguousarray(obn_scores, np.float32).reshape(-1)).to(dev)
    im_d = torch.from_numpy(np.ascontiguousarray(im)).to(dev)
    # plain + mirrored pass of one scale share a forward pass (two images, per-image segments)
    pair_of = {}
    if cfg.NAWS.TTA_PAIR_FLIPS:
        for i, (s, m, f) in enumerate(passes):
            if f:
                for j, (s2, m2, f2) in enumerate(passes):
                    if not f2 and (s2, m2) == (s, m) and j not in pair_of.values():
                        pair_of[i] = j
                        break
    specs = [(scales[i], w, f, 1.0 if i in pair_of else 0.0) for i, (_s, _m, f) in enumerate(passes)]
    dd = ops.roi_dedup(boxes_d, obn_d, specs, cfg.DEDUP_BOXES)
    counts = dd['count'].cpu().tolist()

    def blob(i):
        s, fl = scales[i], passes[i][2]
        oh, ow = int(np.round(h * s)), int(np.round(w * s))
        data = torch.zeros((1, 3, oh, ow), device=dev, dtype=torch.float32)
        ops.prep_image(im_d, data[0], s, flip=fl, means=cfg.PIXEL_MEANS.reshape(-1)[:3],
                       stds=np.asarray(cfg.PIXEL_STDS).reshape(-1)[:3])
        return data

    results = {}                                   # pass -> unique-roi scores [m, K] on the device
    done = set()
    for i in range(len(passes)):
        if i in done:
            continue
        group = [i]
        if i in pair_of:
            group = [pair_of[i], i]                # batch 0 = plain, batch 1 = mirrored
        elif i in pair_of.values():
            group = [i, [k for k, v in pair_of.items() if v == i][0]]
        seg = [0]
        for p in group:
            seg.append(seg[-1] + counts[p])
        executor.feed(dict(
            data=torch.cat([blob(p) for p in group], 0) if len(group) > 1 else blob(group[0]),
            rois=torch.cat([dd['rois'][p, :counts[p]] for p in group], 0),
            obn_scores=torch.cat([dd['obn'][p, :counts[p]] for p in group], 0).reshape(-1, 1),
            _seg=seg))
        executor.run()
        sc = executor.fetch('cls_prob')
        for b, p in enumerate(group):
            results[p] = sc[seg[b]:seg[b + 1]]
            done.add(p)
    acc = torch.empty((n, num_classes), device=dev, dtype=torch.float32)
    for i in range(len(passes)):                   # the reference's summation order
        ops.tta_accumulate(results[i].contiguous(), dd['inv'][i], acc, first=(i == 0))
    if len(passes) > 1:
        ops.tta_finish(acc, len(passes))
    keep = ops.nms_per_class(boxes_d, acc[:, 1:].contiguous(), cfg.TEST.SCORE_THRESH, cfg.TEST.NMS)
    limit = int(cfg.TEST.DETECTIONS_PER_IM)
    cap = max(4 * limit, 1024) if limit > 0 else n * (num_classes - 1)
    while True:
        ints, sc = ops.det_limit(acc, keep, limit, cap)
        packed = torch.cat([ints, sc.view(torch.int32)]).cpu().numpy()      # the one result download
        cnt = int(packed[0])
        if cnt <= cap:
            break
        cap = cnt                                  # more ties at the threshold than the buffer held
    cls_i, row_i = packed[1:1 + cnt], packed[1 + cap:1 + cap + cnt]
    score = packed[1 + 2 * cap:1 + 2 * cap + cnt].view(np.float32)
    cls_boxes = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    for j in range(1, num_classes):
        m = cls_i == j
        cls_boxes[j] = np.hstack((box_proposals[row_i[m]].astype(np.float32, copy=False),
                                  score[m][:, np.newaxis])).astype(np.float32, copy=False)
    return cls_boxes


def im_detect_all(executor, im, box_proposals, obn_scores):
    if device_post_supported(executor, im, len(box_proposals)):
        return im_detect_all_device(executor, im, box_proposals, obn_scores)
    if cfg.TEST.BBOX_AUG.ENABLED:
        scores, boxes = im_detect_bbox_aug(executor, im, box_proposals, obn_scores)
    else:
        scores, boxes = im_detect_bbox(executor, im, cfg.TEST.SCALE, cfg.TEST.MAX_SIZE,
                                       box_proposals, obn_scores)
    scores, boxes, cls_boxes = box_results_with_nms_and_limit(scores, boxes)
    return cls_boxes
