"""COCO-json dataset + MCG proposals -> roidb, for the WSL path.

Mirrors detectron/datasets/json_dataset_wsl.py: JsonDataset.__init__ (:54-85), get_roidb
(:87-140), _prep_roidb_entry (:142-174), _add_gt_annotations (:176-282), _add_proposals_from_file
(:493-566), _merge_proposal_boxes_into_roidb (:633-700), _filter_crowd_proposals (:703-720),
_add_class_assignments (:723-741), _filter_no_class (:744-754), _sort_proposals (:757-762).
The COCO json is read with the json module (pycocotools is not needed for the fields this path
uses: images, annotations' bbox / area / category_id / iscrowd / ignore, categories).
Keypoints, segmentation masks and the pseudo-ground-truth path (cfg.USE_PSEUDO) belong to model
families outside the hot path and are not restated.

Proposal file (tools/convert_mcg.py:28-60): pickle of
  {'boxes': [uint16/float [n_i,4] x1 y1 x2 y2], 'scores': [float32 [n_i,1]], 'indexes': [image id]}.
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


class JsonDataset(object):
    """A COCO-format json dataset."""

    def __init__(self, name):
        assert dataset_catalog.contains(name), 'Unknown dataset name: {}'.format(name)
        assert os.path.exists(dataset_catalog.get_im_dir(name)), \
            'Im dir \'{}\' not found'.format(dataset_catalog.get_im_dir(name))
        assert os.path.exists(dataset_catalog.get_ann_fn(name)), \
            'Ann fn \'{}\' not found'.format(dataset_catalog.get_ann_fn(name))
        self.name = name
        self.image_directory = dataset_catalog.get_im_dir(name)
        self.image_prefix = dataset_catalog.get_im_prefix(name)
        with open(dataset_catalog.get_ann_fn(name)) as f:
            js = json.load(f)
        self._images = {im['id']: im for im in js.get('images', [])}
        self._anns_by_image = {}
        for ann in js.get('annotations', []):
            self._anns_by_image.setdefault(ann['image_id'], []).append(ann)
        cats = sorted(js.get('categories', []), key=lambda c: c['id'])     # COCO.getCatIds() is sorted
        category_ids = [c['id'] for c in cats]
        categories = [c['name'] for c in cats]
        self.category_to_id_map = dict(zip(categories, category_ids))
        self.classes = ['__background__'] + categories
        self.num_classes = len(self.classes)
        self.json_category_id_to_contiguous_id = {v: i + 1 for i, v in enumerate(category_ids)}
        self.contiguous_category_id_to_json_id = {
            v: k for k, v in self.json_category_id_to_contiguous_id.items()}
        self.keypoints = None

    def get_roidb(self, gt=False, proposal_file=None, min_proposal_size=20, proposal_limit=-1,
                  crowd_filter_thresh=0):
        assert gt is True or crowd_filter_thresh == 0, \
            'Crowd filter threshold must be 0 if ground-truth annotations are not included.'
        image_ids = sorted(self._images.keys())
        roidb = copy.deepcopy([self._images[i] for i in image_ids])
        for entry in roidb:
            self._prep_roidb_entry(entry)
        if gt:
            for entry in roidb:
                self._add_gt_annotations(entry)
        if proposal_file is not None:
            self._add_proposals_from_file(roidb, proposal_file, min_proposal_size, proposal_limit,
                                          crowd_filter_thresh)
        _add_class_assignments(roidb)
        if gt:
            roidb = _filter_no_class(self.name, roidb)
        return roidb

    def _prep_roidb_entry(self, entry):
        entry['dataset'] = self
        im_path = os.path.join(self.image_directory, self.image_prefix + entry['file_name'])
        assert os.path.exists(im_path), 'Image \'{}\' not found'.format(im_path)
        entry['image'] = im_path
        entry['flipped'] = False
        entry['has_visible_keypoints'] = False
        entry['boxes'] = np.empty((0, 4), dtype=np.float32)
        entry['obn_scores'] = np.empty((0, 1), dtype=np.float32)
        entry['segms'] = []
        entry['gt_classes'] = np.empty((0), dtype=np.int32)
        entry['seg_areas'] = np.empty((0), dtype=np.float32)
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(
            np.empty((0, self.num_classes), dtype=np.float32))
        entry['is_crowd'] = np.empty((0), dtype=bool)
        entry['box_to_gt_ind_map'] = np.empty((0), dtype=np.int32)
        for k in ['date_captured', 'url', 'license', 'file_name']:
            if k in entry:
                del entry[k]

    def _add_gt_annotations(self, entry):
        objs = self._anns_by_image.get(entry['id'], [])
        valid_objs = []
        width, height = entry['width'], entry['height']
        all_diffcult_truncated = True
        for obj in objs:
            if obj['area'] < cfg.TRAIN.GT_MIN_AREA:
                continue
            if 'ignore' in obj and obj['ignore'] == 1:
                continue
            # (sic: the reference reads the key 'diffcult')
            if 'diffcult' in obj:
                if obj['diffcult'] == 0:
                    all_diffcult_truncated = False
            else:
                all_diffcult_truncated = False
            if 'truncated' in obj:
                if obj['truncated'] == 0:
                    all_diffcult_truncated = False
            else:
                all_diffcult_truncated = False
            x1, y1, x2, y2 = box_utils.xywh_to_xyxy(obj['bbox'])
            x1, y1, x2, y2 = box_utils.clip_xyxy_to_image(x1, y1, x2, y2, height, width)
            if obj['area'] > 0 and x2 > x1 and y2 > y1:
                obj = dict(obj, clean_bbox=[x1, y1, x2, y2])
                valid_objs.append(obj)
        if all_diffcult_truncated:
            valid_objs = []
        n = len(valid_objs)
        boxes = np.zeros((n, 4), dtype=entry['boxes'].dtype)
        obn_scores = np.zeros((n, 1), dtype=entry['obn_scores'].dtype)
        gt_classes = np.zeros((n), dtype=entry['gt_classes'].dtype)
        gt_overlaps = np.zeros((n, self.num_classes), dtype=entry['gt_overlaps'].dtype)
        seg_areas = np.zeros((n), dtype=entry['seg_areas'].dtype)
        is_crowd = np.zeros((n), dtype=entry['is_crowd'].dtype)
        box_to_gt_ind_map = np.zeros((n), dtype=entry['box_to_gt_ind_map'].dtype)
        for ix, obj in enumerate(valid_objs):
            cls = self.json_category_id_to_contiguous_id[obj['category_id']]
            boxes[ix, :] = obj['clean_bbox']
            gt_classes[ix] = cls
            seg_areas[ix] = obj['area']
            is_crowd[ix] = obj.get('iscrowd', 0)
            box_to_gt_ind_map[ix] = ix
            if obj.get('iscrowd', 0):
                gt_overlaps[ix, :] = -1.0     # excluded during training
            else:
                gt_overlaps[ix, cls] = 1.0
        entry['boxes'] = np.append(entry['boxes'], boxes, axis=0)
        entry['obn_scores'] = np.append(entry['obn_scores'], obn_scores, axis=0)
        entry['gt_classes'] = np.append(entry['gt_classes'], gt_classes)
        entry['seg_areas'] = np.append(entry['seg_areas'], seg_areas)
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(
            np.append(entry['gt_overlaps'].toarray(), gt_overlaps, axis=0))
        entry['is_crowd'] = np.append(entry['is_crowd'], is_crowd)
        entry['box_to_gt_ind_map'] = np.append(entry['box_to_gt_ind_map'], box_to_gt_ind_map)

    def _add_proposals_from_file(self, roidb, proposal_file, min_proposal_size, top_k,
                                 crowd_thresh):
        logger.info('Loading proposals from: {}'.format(proposal_file))
        proposals = load_object(proposal_file)
        id_field = 'indexes' if 'indexes' in proposals else 'ids'
        _sort_proposals(proposals, id_field)
        box_list, score_list = [], []
        for i, entry in enumerate(roidb):
            boxes = np.asarray(proposals['boxes'][i])
            scores = np.asarray(proposals['scores'][i])
            assert entry['id'] == proposals[id_field][i]
            assert (boxes[:, 0] >= 0).all() and (boxes[:, 1] >= 0).all()
            assert (boxes[:, 2] >= boxes[:, 0]).all() and (boxes[:, 3] >= boxes[:, 1]).all()
            assert (boxes[:, 2] < entry['width']).all(), entry['image']
            assert (boxes[:, 3] < entry['height']).all(), entry['image']
            keep = box_utils.unique_boxes(boxes)
            boxes, scores = boxes[keep, :], scores[keep]
            keep = box_utils.filter_small_boxes(boxes, min_proposal_size)
            boxes, scores = boxes[keep, :], scores[keep]
            sorted_ind = np.argsort(-scores.flatten())          # by confidence
            boxes, scores = boxes[sorted_ind, :], scores[sorted_ind, :]
            if top_k > 0:
                boxes, scores = boxes[:top_k, :], scores[:top_k]
            box_list.append(boxes)
            score_list.append(scores)
        _merge_proposal_boxes_into_roidb(roidb, box_list, score_list)
        if crowd_thresh > 0:
            _filter_crowd_proposals(roidb, crowd_thresh)


def _merge_proposal_boxes_into_roidb(roidb, box_list, score_list):
    assert len(box_list) == len(roidb)
    for i, entry in enumerate(roidb):
        boxes, scores = box_list[i], score_list[i]
        num_boxes = boxes.shape[0]
        gt_overlaps = np.zeros((num_boxes, entry['gt_overlaps'].shape[1]),
                               dtype=entry['gt_overlaps'].dtype)
        box_to_gt_ind_map = -np.ones((num_boxes), dtype=entry['box_to_gt_ind_map'].dtype)
        gt_inds = np.where(entry['gt_classes'] > 0)[0]
        if len(gt_inds) > 0:
            gt_boxes = entry['boxes'][gt_inds, :]
            gt_classes = entry['gt_classes'][gt_inds]
            ov = box_utils.bbox_overlaps(boxes.astype(dtype=np.float32, copy=False),
                                         gt_boxes.astype(dtype=np.float32, copy=False))
            argmaxes = ov.argmax(axis=1)
            maxes = ov.max(axis=1)
            I = np.where(maxes > 0)[0]
            gt_overlaps[I, gt_classes[argmaxes[I]]] = maxes[I]
            box_to_gt_ind_map[I] = gt_inds[argmaxes[I]]
        entry['boxes'] = np.append(entry['boxes'], boxes.astype(entry['boxes'].dtype, copy=False),
                                   axis=0)
        entry['obn_scores'] = np.append(
            entry['obn_scores'], scores.astype(entry['obn_scores'].dtype, copy=False), axis=0)
        entry['gt_classes'] = np.append(entry['gt_classes'],
                                        np.zeros((num_boxes), dtype=entry['gt_classes'].dtype))
        entry['seg_areas'] = np.append(entry['seg_areas'],
                                       np.zeros((num_boxes), dtype=entry['seg_areas'].dtype))
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(
            np.append(entry['gt_overlaps'].toarray(), gt_overlaps, axis=0))
        entry['is_crowd'] = np.append(entry['is_crowd'],
                                      np.zeros((num_boxes), dtype=entry['is_crowd'].dtype))
        entry['box_to_gt_ind_map'] = np.append(
            entry['box_to_gt_ind_map'],
            box_to_gt_ind_map.astype(entry['box_to_gt_ind_map'].dtype, copy=False))


def _filter_crowd_proposals(roidb, crowd_thresh):
    for entry in roidb:
        gt_overlaps = entry['gt_overlaps'].toarray()
        crowd_inds = np.where(entry['is_crowd'] == 1)[0]
        non_gt_inds = np.where(entry['gt_classes'] == 0)[0]
        if len(crowd_inds) == 0 or len(non_gt_inds) == 0:
            continue
        crowd_boxes = box_utils.xyxy_to_xywh(entry['boxes'][crowd_inds, :])
        non_gt_boxes = box_utils.xyxy_to_xywh(entry['boxes'][non_gt_inds, :])
        ious = box_utils.crowd_iou(non_gt_boxes, crowd_boxes)
        bad_inds = np.where(ious.max(axis=1) > crowd_thresh)[0]
        gt_overlaps[non_gt_inds[bad_inds], :] = -1
        entry['gt_overlaps'] = scipy.sparse.csr_matrix(gt_overlaps)


def _add_class_assignments(roidb):
    for entry in roidb:
        gt_overlaps = entry['gt_overlaps'].toarray()
        max_overlaps = gt_overlaps.max(axis=1)
        max_classes = gt_overlaps.argmax(axis=1)
        entry['max_classes'] = max_classes
        entry['max_overlaps'] = max_overlaps
        assert all(max_classes[np.where(max_overlaps == 0)[0]] == 0)
        assert all(max_classes[np.where(max_overlaps > 0)[0]] != 0)


def _filter_no_class(name, roidb):
    if 'test' in name:
        return roidb
    return [entry for entry in roidb if sum(entry['max_classes']) != 0]


def _sort_proposals(proposals, id_field):
    order = np.argsort(proposals[id_field])
    for k in ['boxes', id_field, 'scores']:
        proposals[k] = [proposals[k][i] for i in order]
