"""COCO-format json + MCG proposal pickle -> roidb for the WSL path.

Same behaviour and roidb schema as the reference's detectron/datasets/json_dataset_wsl.py
(class JsonDataset :51-140; entry fields :142-174; ground-truth sanitising :176-282; proposal
ingestion :493-566 and merge :633-700; crowd filter :703-720; class assignment :723-741;
_filter_no_class :744-754; _sort_proposals :757-762), written against plain `json` instead of
pycocotools and organised around small array helpers.  Keypoints, masks and the pseudo-GT path
(cfg.USE_PSEUDO) belong to model families outside the hot path.

Proposal file (tools/convert_mcg.py:28-60):
  {'boxes': [n_i x 4 (x1 y1 x2 y2)], 'scores': [n_i x 1], 'indexes' | 'ids': [image id]}.
"""
import copy
import json
import logging
import os

import numpy as np
import scipy.sparse

from detectron.core.config import cfg
from detectron.datasets import dataset_catalog
from detectron.utils import boxes as box_utils
from detectron.utils.net_wsl import load_object

logger = logging.getLogger(__name__)

# per-box arrays of a roidb entry: (dtype, trailing shape)
_BOX_FIELDS = {
    'boxes': (np.float32, (4,)), 'obn_scores': (np.float32, (1,)), 'gt_classes': (np.int32, ()),
    'seg_areas': (np.float32, ()), 'is_crowd': (bool, ()), 'box_to_gt_ind_map': (np.int32, ()),
}
_JSON_ONLY_KEYS = ('date_captured', 'url', 'license', 'file_name')


def _grow(entry, overlaps, **columns):
    """Append rows to every per-box array of `entry` (+ the sparse class-overlap matrix)."""
    for key, rows in columns.items():
        dtype = _BOX_FIELDS[key][0]
        entry[key] = np.concatenate([entry[key], np.asarray(rows).astype(dtype, copy=False)], axis=0)
    dense = np.vstack([entry['gt_overlaps'].toarray(), overlaps])
    entry['gt_overlaps'] = scipy.sparse.csr_matrix(dense)


class JsonDataset(object):
    def __init__(self, name):
        assert dataset_catalog.contains(name), 'Unknown dataset name: {}'.format(name)
        self.name = name
        self.image_directory = dataset_catalog.get_im_dir(name)
        self.image_prefix = dataset_catalog.get_im_prefix(name)
        ann_file = dataset_catalog.get_ann_fn(name)
        assert os.path.exists(self.image_directory), \
            'Im dir \'{}\' not found'.format(self.image_directory)
        assert os.path.exists(ann_file), 'Ann fn \'{}\' not found'.format(ann_file)
        with open(ann_file) as f:
            spec = json.load(f)
        self._images = {rec['id']: rec for rec in spec.get('images', ())}
        self._anns = {}
        for a in spec.get('annotations', ()):
            self._anns.setdefault(a['image_id'], []).append(a)
        cats = sorted(spec.get('categories', ()), key=lambda c: c['id'])   # COCO.getCatIds order
        self.classes = ['__background__'] + [c['name'] for c in cats]
        self.num_classes = len(self.classes)
        self.category_to_id_map = {c['name']: c['id'] for c in cats}
        self.json_category_id_to_contiguous_id = {c['id']: i + 1 for i, c in enumerate(cats)}
        self.contiguous_category_id_to_json_id = {
            v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.keypoints = None

    # ------------------------------------------------------------------ roidb
    def get_roidb(self, gt=False, proposal_file=None, min_proposal_size=20, proposal_limit=-1,
                  crowd_filter_thresh=0):
        assert gt is True or crowd_filter_thresh == 0, \
            'Crowd filter threshold must be 0 if ground-truth annotations are not included.'
        roidb = [self._blank_entry(copy.deepcopy(self._images[i])) for i in sorted(self._images)]
        if gt:
            for entry in roidb:
                self._add_gt_annotations(entry)
        if proposal_file is not None:
            self._add_proposals_from_file(roidb, proposal_file, min_proposal_size, proposal_limit,
                                          crowd_filter_thresh)
        _add_class_assignments(roidb)
        return _filter_no_class(self.name, roidb) if gt else roidb

    def _blank_entry(self, entry):
        path = os.path.join(self.image_directory, self.image_prefix + entry['file_name'])
        assert os.path.exists(path), 'Image \'{}\' not found'.format(path)
        entry.update(dataset=self, image=path, flipped=False, has_visible_keypoints=False, segms=[])
        for key, (dtype, tail) in _BOX_FIELDS.items():
            entry[key] = np.empty((0,) + tail, dtype=dtype)
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(np.empty((0, self.num_classes), np.float32))
        for k in _JSON_ONLY_KEYS:
            entry.pop(k, None)
        return entry

    # (older call sites use this name)
    def _prep_roidb_entry(self, entry):
        self._blank_entry(entry)

    def _usable_objects(self, entry):
        """Annotations that survive the reference's sanitising, each with its clipped xyxy box.
        An image whose objects are ALL flagged difficult and truncated yields none (:197-206,222;
        the reference spells the key 'diffcult')."""
        h, w = entry['height'], entry['width']
        keep, all_hard = [], True
        for obj in self._anns.get(entry['id'], ()):
            if obj['area'] < cfg.TRAIN.GT_MIN_AREA or obj.get('ignore', 0) == 1:
                continue
            if obj.get('diffcult', 0) == 0 or obj.get('truncated', 0) == 0:
                all_hard = False
            box = box_utils.clip_xyxy_to_image(*box_utils.xywh_to_xyxy(list(obj['bbox'])), h, w)
            if obj['area'] > 0 and box[2] > box[0] and box[3] > box[1]:
                keep.append((obj, box))
        return [] if all_hard else keep

    def _add_gt_annotations(self, entry):
        objs = self._usable_objects(entry)
        n = len(objs)
        cls = np.array([self.json_category_id_to_contiguous_id[o['category_id']] for o, _ in objs],
                       np.int32).reshape(n)
        crowd = np.array([bool(o.get('iscrowd', 0)) for o, _ in objs], bool).reshape(n)
        overlaps = np.zeros((n, self.num_classes), np.float32)
        overlaps[np.arange(n), cls] = 1.0
        overlaps[crowd, :] = -1.0                # crowd regions never count as a class
        _grow(entry, overlaps,
              boxes=np.array([b for _, b in objs], np.float32).reshape(n, 4),
              obn_scores=np.zeros((n, 1), np.float32), gt_classes=cls,
              seg_areas=np.array([o['area'] for o, _ in objs], np.float32).reshape(n),
              is_crowd=crowd, box_to_gt_ind_map=np.arange(n, dtype=np.int32))

    # -------------------------------------------------------------- proposals
    def _add_proposals_from_file(self, roidb, proposal_file, min_proposal_size, top_k,
                                 crowd_thresh):
        logger.info('Loading proposals from: {}'.format(proposal_file))
        proposals = load_object(proposal_file)
        id_field = 'indexes' if 'indexes' in proposals else 'ids'
        _sort_proposals(proposals, id_field)
        box_list, score_list = [], []
        for entry, pid, boxes, scores in zip(roidb, proposals[id_field], proposals['boxes'],
                                             proposals['scores']):
            assert entry['id'] == pid
            b, s = _select_proposals(np.asarray(boxes), np.asarray(scores), entry,
                                     min_proposal_size, top_k)
            box_list.append(b)
            score_list.append(s)
        assert len(box_list) == len(roidb)
        _merge_proposal_boxes_into_roidb(roidb, box_list, score_list)
        if crowd_thresh > 0:
            _filter_crowd_proposals(roidb, crowd_thresh)


def _select_proposals(boxes, scores, entry, min_size, top_k):
    """Sanity checks, then: first occurrence of each distinct box, both sides > min_size, highest
    score first, at most top_k (:519-545)."""
    assert (boxes[:, :2] >= 0).all() and (boxes[:, 2:] >= boxes[:, :2]).all()
    assert (boxes[:, 2] < entry['width']).all() and (boxes[:, 3] < entry['height']).all(), \
        entry['image']
    for keep_fn in (box_utils.unique_boxes, lambda b: box_utils.filter_small_boxes(b, min_size)):
        keep = keep_fn(boxes)
        boxes, scores = boxes[keep, :], scores[keep]
    order = np.argsort(-scores.flatten())
    boxes, scores = boxes[order, :], scores[order, :]
    if top_k > 0:
        boxes, scores = boxes[:top_k, :], scores[:top_k]
    return boxes, scores


def _merge_proposal_boxes_into_roidb(roidb, box_list, score_list):
    """Proposals become non-gt rows: class overlap = IoU with the best-overlapping gt box (crowd
    boxes included at this stage), recorded under that box's class (:633-700)."""
    assert len(box_list) == len(roidb)
    for entry, boxes, scores in zip(roidb, box_list, score_list):
        n = boxes.shape[0]
        overlaps = np.zeros((n, entry['gt_overlaps'].shape[1]), np.float32)
        to_gt = np.full((n,), -1, np.int32)
        gt_rows = np.flatnonzero(entry['gt_classes'] > 0)
        if gt_rows.size and n:
            iou = box_utils.bbox_overlaps(boxes.astype(np.float32, copy=False),
                                          entry['boxes'][gt_rows].astype(np.float32, copy=False))
            best = iou.argmax(axis=1)
            best_iou = iou[np.arange(n), best]
            hit = np.flatnonzero(best_iou > 0)
            overlaps[hit, entry['gt_classes'][gt_rows][best[hit]]] = best_iou[hit]
            to_gt[hit] = gt_rows[best[hit]]
        _grow(entry, overlaps, boxes=boxes, obn_scores=scores,
              gt_classes=np.zeros((n,), np.int32), seg_areas=np.zeros((n,), np.float32),
              is_crowd=np.zeros((n,), bool), box_to_gt_ind_map=to_gt)


def _filter_crowd_proposals(roidb, crowd_thresh):
    """Proposals lying mostly inside a crowd region get overlap -1 with every class (:703-720;
    pycocotools' iscrowd IoU = intersection / proposal area)."""
    for entry in roidb:
        crowd = np.flatnonzero(entry['is_crowd'] == 1)
        props = np.flatnonzero(entry['gt_classes'] == 0)
        if crowd.size == 0 or props.size == 0:
            continue
        inside = box_utils.crowd_iou(box_utils.xyxy_to_xywh(entry['boxes'][props]),
                                     box_utils.xyxy_to_xywh(entry['boxes'][crowd])).max(axis=1)
        dense = entry['gt_overlaps'].toarray()
        dense[props[inside > crowd_thresh], :] = -1
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(dense)


def _add_class_assignments(roidb):
    for entry in roidb:
        dense = entry['gt_overlaps'].toarray()
        entry['max_classes'] = dense.argmax(axis=1)
        entry['max_overlaps'] = dense.max(axis=1)
        # zero overlap <-> background class, positive overlap <-> a foreground class
        assert (entry['max_classes'][entry['max_overlaps'] == 0] == 0).all()
        assert (entry['max_classes'][entry['max_overlaps'] > 0] != 0).all()


def _filter_no_class(name, roidb):
    """Train/val images whose boxes all resolve to background are dropped (:744-754)."""
    if 'test' in name:
        return roidb
    return [e for e in roidb if int(np.sum(e['max_classes'])) != 0]


def _sort_proposals(proposals, id_field):
    order = np.argsort(proposals[id_field])
    for key in ('boxes', id_field, 'scores'):
        proposals[key] = [proposals[key][i] for i in order]
