#!/usr/bin/env python3
"""Train the noise-aware WSDDN on MI355X.  Same CLI as the reference tool
(tools/train_net_wsl.py:52-84):

    python tools/train_net_wsl.py --cfg configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml \
        [--multi-gpu-testing] [--skip-test] KEY VALUE ...

Multi-GPU: one process per GPU, e.g.
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/train_net_wsl.py --cfg ...
(NUM_GPUS in the yaml is overridden by WORLD_SIZE x NAWS.IMS_PER_GPU for the SGD normaliser).
"""
import argparse
import logging
import os
import pprint
import random
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))

import numpy as np  # noqa: E402

from detectron.core.config import (assert_and_infer_cfg, cfg, merge_cfg_from_file,  # noqa: E402
                                   merge_cfg_from_list)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train a network with Detectron (MI355X hot path)')
    p.add_argument('--cfg', dest='cfg_file', help='Config file for training (and optionally testing)',
                   default=None, type=str)
    p.add_argument('--multi-gpu-testing', dest='multi_gpu_testing', action='store_true',
                   help='Use cfg.NUM_GPUS GPUs for inference')
    p.add_argument('--skip-test', dest='skip_test', action='store_true',
                   help='Do not test the final model')
    p.add_argument('--max-iter', dest='max_iter', type=int, default=None,
                   help='stop after this many iterations (smoke runs)')
    p.add_argument('opts', help='See detectron/core/config.py for all options', default=None,
                   nargs=argparse.REMAINDER)
    if argv is None and len(sys.argv) == 1:
        p.print_help()
        sys.exit(1)
    return p.parse_args(argv)


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format='%(levelname)s %(filename)s:%(lineno)4d: %(message)s')
    logger = logging.getLogger(__name__)
    args = parse_args(argv)
    if args.cfg_file is not None:
        merge_cfg_from_file(args.cfg_file)
    if args.opts:
        merge_cfg_from_list(args.opts)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    merge_cfg_from_list(['NUM_GPUS', world * cfg.NAWS.IMS_PER_GPU])
    assert_and_infer_cfg()
    logger.info('Training with config:')
    logger.info(pprint.pformat(cfg))
    # The reference's loader threads of all GPUs draw scale / distortion / crop / mixup values from
    # ONE process-wide stream (tools/train_net_wsl.py:113), so every GPU sees different values;
    # with one process per GPU each rank needs its own stream (rank 0 keeps the reference seed).
    # Python's `random` (mixup partner, loader_wsl.py:136-144) is seeded too.
    rank = int(os.environ.get('RANK', '0'))
    np.random.seed(cfg.RNG_SEED + rank)
    random.seed(cfg.RNG_SEED + rank)
    from detectron.utils import train_wsl
    checkpoints = train_wsl.train_model(max_iter=args.max_iter)
    if world > 1:
        # the ranks of a torchrun job end together; rank 0 alone runs the test (single process,
        # or - with --multi-gpu-testing - as the parent of one fresh child per GPU)
        import torch.distributed as dist
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
    results = None
    if not args.skip_test and rank == 0:
        # Test the trained model (reference tools/train_net_wsl.py:118-160: the final checkpoint
        # with the yaml's TTA; the per-snapshot re-tests without TTA follow)
        # N ranks trained: test through one fresh child per GPU by default, so that all N GPUs
        # work and no test shares this process's HIP context and cached memory (rank 0 is only
        # the children's parent; its engine / executor are gone once train_model has returned)
        multi = args.multi_gpu_testing or world > 1
        results = test_model(checkpoints['final'], multi, world)
        print('reprint snapshot name for the result: ', checkpoints['final'])
        _ = checkpoints.pop('final', None)
        if checkpoints:
            from detectron.core import config as core_config
            core_config.cfg.immutable(False)
            core_config.cfg.TEST.BBOX_AUG.ENABLED = False
            core_config.cfg.VIS = False
            core_config.cfg.immutable(True)
            for snapshot in sorted(checkpoints.keys(), reverse=True):
                test_model(checkpoints[snapshot], multi, world)
                print('reprint snapshot name for the result: ', snapshot, checkpoints[snapshot])
    return results


def test_model(model_file, multi_gpu_testing, num_gpus=None):
    """Test a model (reference tools/train_net_wsl.py:163-171): every dataset of TEST.DATASETS
    through the test engine; parameters and activations of the training run are released first."""
    import gc
    import torch
    from detectron.core import test_engine_wsl
    gc.collect()
    torch.cuda.empty_cache()
    if not multi_gpu_testing or num_gpus is None or num_gpus == cfg.NUM_GPUS:
        return test_engine_wsl.run_inference(model_file, multi_gpu_testing=multi_gpu_testing)
    # training counted NUM_GPUS = processes x images per process (the SGD normaliser); the test
    # engine starts one child per GPU = per training process
    from detectron.core import config as core_config
    saved = cfg.NUM_GPUS
    core_config.cfg.immutable(False)
    core_config.cfg.NUM_GPUS = int(num_gpus)
    core_config.cfg.immutable(True)
    try:
        return test_engine_wsl.run_inference(model_file, multi_gpu_testing=True)
    finally:
        core_config.cfg.immutable(False)
        core_config.cfg.NUM_GPUS = saved
        core_config.cfg.immutable(True)


if __name__ == '__main__':
    main()
