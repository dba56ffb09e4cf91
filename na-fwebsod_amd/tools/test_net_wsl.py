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
    roidb = synthetic.make_roidb(args.num_images, min(cfg.TEST.PROPOSAL_LIMIT, 2000),
                                 cfg.MODEL.NUM_CLASSES - 1, seed=cfg.RNG_SEED)
    lo, hi = args.range if args.range else (0, len(roidb))
    for i in range(lo, hi):
        e = roidb[i]
        im = synthetic.make_image(e).transpose(1, 2, 0) + synthetic.PIXEL_MEANS_BGR
        scores, boxes = test_wsl.im_detect_bbox(ex, im.astype(np.float32), cfg.TEST.SCALE,
                                                cfg.TEST.MAX_SIZE, e['boxes'], e['obn_scores'])
        print('image %d: %d proposals, top class score %.4g' % (i, boxes.shape[0], scores[:, 1:].max()))


if __name__ == '__main__':
    main()
