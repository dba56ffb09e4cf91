#!/usr/bin/env python3
"""Forward-only inference of the noise-aware WSDDN (reference CLI: tools/test_net_wsl.py
`--cfg FILE [--range a b] [--multi-gpu-testing] [--vis] [--wait] KEY VALUE ...`).  Runs
`im_detect_bbox` (core/test_wsl.py:102-178) on a roidb and writes raw per-proposal class
scores; NMS / TTA / dataset evaluation are the "next" rows of SURVEY.md §8f."""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from detectron.core.config import (assert_and_infer_cfg, cfg, merge_cfg_from_file,  # noqa: E402
                                   merge_cfg_from_list)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Test a WSL network (MI355X hot path)')
    p.add_argument('--cfg', dest='cfg_file', default=None, type=str)
    p.add_argument('--range', dest='range', nargs=2, type=int, default=None)
    p.add_argument('--multi-gpu-testing', dest='multi_gpu_testing', action='store_true')
    p.add_argument('--vis', dest='vis', action='store_true')
    p.add_argument('--wait', dest='wait', default=True, type=bool)
    p.add_argument('--num-images', type=int, default=4)
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    return p.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.cfg_file:
        merge_cfg_from_file(args.cfg_file)
    if args.opts:
        merge_cfg_from_list(args.opts)
    assert_and_infer_cfg()
    from detectron.core import test_wsl
    from detectron.core.executor import NetExecutor
    from detectron.datasets import synthetic
    import detectron.modeling.model_builder_wsl as model_builder
    import detectron.utils.net_wsl as nu
    device = torch.device('cuda', int(os.environ.get('LOCAL_RANK', '0')))
    model = model_builder.create(cfg.MODEL.TYPE, train=False)
    ex = NetExecutor(model, device)
    ex.init_params()
    if cfg.TEST.WEIGHTS and os.path.exists(cfg.TEST.WEIGHTS):
        nu.initialize_from_weights_file(model, cfg.TEST.WEIGHTS, ex, broadcast=False)
    # test roidb: the named dataset + proposal file when on disk (test_engine.py:324-352:
    # JsonDataset.get_roidb(proposal_file, proposal_limit)), else synthetic entries
    from detectron.datasets import dataset_catalog
    names = tuple(cfg.TEST.DATASETS)
    real = bool(names) and dataset_catalog.contains(names[0]) and \
        os.path.exists(dataset_catalog.get_ann_fn(names[0]))
    if real:
        from detectron.datasets.json_dataset_wsl import JsonDataset
        from detectron.roi_data.minibatch_wsl import _read_image
        pf = cfg.TEST.PROPOSAL_FILES[0] if len(cfg.TEST.PROPOSAL_FILES) else None
        roidb = JsonDataset(names[0]).get_roidb(proposal_file=pf,
                                                proposal_limit=cfg.TEST.PROPOSAL_LIMIT)
    else:
        roidb = synthetic.make_roidb(args.num_images, min(cfg.TEST.PROPOSAL_LIMIT, 2000),
                                     cfg.MODEL.NUM_CLASSES - 1, seed=cfg.RNG_SEED)
    lo, hi = args.range if args.range else (0, len(roidb))
    num_classes = cfg.MODEL.NUM_CLASSES
    # all_boxes[cls][image] = N x 5 (x1, y1, x2, y2, score), test_engine.py:236-300
    all_boxes = [[[] for _ in range(len(roidb))] for _ in range(num_classes)]
    for i in range(lo, hi):
        e = roidb[i]
        if real:
            im = _read_image(e).astype(np.float32)
        else:
            im = synthetic.make_image(e).transpose(1, 2, 0) + synthetic.PIXEL_MEANS_BGR
        sel = (e['gt_classes'] == 0) if real else slice(None)    # proposals only (test_engine.py:256)
        if e['boxes'][sel].shape[0] == 0:                        # test_engine.py:257-258: skip
            for j in range(1, num_classes):
                all_boxes[j][i] = np.zeros((0, 5), np.float32)
            print('image %d: no proposals' % i)
            continue
        cls_boxes = test_wsl.im_detect_all(ex, im.astype(np.float32), e['boxes'][sel],
                                           e['obn_scores'][sel])
        for j in range(1, num_classes):
            all_boxes[j][i] = cls_boxes[j]
        n_det = sum(len(cls_boxes[j]) for j in range(1, num_classes))
        top = max([cls_boxes[j][:, 4].max() for j in range(1, num_classes) if len(cls_boxes[j])] or [0.0])
        print('image %d: %d proposals -> %d detections, top score %.4g' % (
            i, int(np.sum(sel)) if real else e['boxes'].shape[0], n_det, top))
    from detectron.core.config import get_output_dir
    out_dir = get_output_dir(names if names else ('synthetic',), training=False)
    det_file = os.path.join(out_dir, 'detections.pkl' if not args.range else
                            'detection_range_%s_%s.pkl' % (lo, hi))
    nu.save_object(dict(all_boxes=all_boxes, cfg=str(cfg)), det_file)
    print('Wrote detections to: {}'.format(os.path.abspath(det_file)))
    return all_boxes


if __name__ == '__main__':
    main()
