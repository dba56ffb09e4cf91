#!/usr/bin/env python3
"""Inference of the noise-aware WSDDN over a dataset (reference CLI: tools/test_net_wsl.py
`--cfg FILE [--range a b] [--multi-gpu-testing] [--vis] [--wait] KEY VALUE ...`):

    python tools/test_net_wsl.py --cfg configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml \
        [--multi-gpu-testing] TEST.WEIGHTS model_final.pkl NUM_GPUS 8

One process runs every image (multi-scale + flip TTA, NMS, detections.pkl); with
--multi-gpu-testing this process only starts one fresh child per GPU (`--range a b`, one visible
device each) and collates their range files - it never touches a GPU itself
(detectron/core/test_engine_wsl.py, detectron/utils/subprocess.py)."""
import argparse
import logging
import os
import pprint
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

from detectron.core.config import (assert_and_infer_cfg, cfg, merge_cfg_from_file,  # noqa: E402
                                   merge_cfg_from_list)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Test a WSL network (MI355X hot path)')
    p.add_argument('--cfg', dest='cfg_file', default=None, type=str)
    p.add_argument('--range', dest='range', nargs=2, type=int, default=None,
                   help='start (inclusive) and end (exclusive) indices')
    p.add_argument('--multi-gpu-testing', dest='multi_gpu_testing', action='store_true',
                   help='using cfg.NUM_GPUS for inference')
    p.add_argument('--vis', dest='vis', action='store_true')
    p.add_argument('--wait', dest='wait', default=True, type=bool, help='wait until net file exists')
    p.add_argument('--num-images', type=int, default=None,
                   help='synthetic fallback only: number of images (NAWS.SYNTHETIC_TEST_IMAGES)')
    p.add_argument('opts', default=None, nargs=argparse.REMAINDER)
    return p.parse_args(argv)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(filename)s:%(lineno)4d: %(message)s')
    logger = logging.getLogger(__name__)
    args = parse_args(argv)
    if args.cfg_file:
        merge_cfg_from_file(args.cfg_file)
    if args.opts:
        merge_cfg_from_list(args.opts)
    if args.num_images is not None:
        merge_cfg_from_list(['NAWS.SYNTHETIC_TEST_IMAGES', args.num_images])
    assert_and_infer_cfg()
    logger.info('Testing with config:')
    logger.info(pprint.pformat(cfg))
    # (reference :110-112: a weights file that is still being written by a training job)
    waited = 0
    while cfg.TEST.WEIGHTS and not os.path.exists(cfg.TEST.WEIGHTS) and args.wait and waited < 3600:
        logger.info('Waiting for \'{}\' to exist...'.format(cfg.TEST.WEIGHTS))
        time.sleep(10)
        waited += 10
    # still missing (no --wait, or the hour is over): an error, not a run on random weights
    from detectron.core.test_engine_wsl import check_weights_file
    check_weights_file(cfg.TEST.WEIGHTS)
    from detectron.core import test_engine_wsl
    res = test_engine_wsl.run_inference(cfg.TEST.WEIGHTS, ind_range=args.range,
                                        multi_gpu_testing=args.multi_gpu_testing,
                                        gpu_id=int(os.environ.get('LOCAL_RANK', '0')))
    if args.range is not None:
        return res[0]                     # a child: (all_boxes, all_segms, all_keyps) of its range
    from detectron.core.config import get_output_dir
    from detectron.utils.net_wsl import load_object
    name, _pf = test_engine_wsl.get_inference_dataset(max(0, len(cfg.TEST.DATASETS) - 1))
    logger.info('results: %s', res)
    return load_object(os.path.join(get_output_dir(name, training=False), 'detections.pkl'))['all_boxes']


if __name__ == '__main__':
    main()
