"""Primitives for running one tool per GPU over index ranges of a dataset (reference:
detectron/utils/subprocess.py:40-136): `process_in_parallel` starts cfg.NUM_GPUS FRESH child
processes of `binary --range start end --cfg <snapshot.yaml> NUM_GPUS 1 opts...`, each confined
to one GPU, streams the first child's output, and returns the children's `<tag>_range_<s>_<e>.pkl`
payloads in range order.  The calling process does no GPU work here."""
import json
import logging
import os
import subprocess as _sp
import sys

import numpy as np

from detectron.core.config import cfg
from detectron.utils.net_wsl import load_object
import detectron.utils.env as envu

logger = logging.getLogger(__name__)

_RANK_VARS = ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'LOCAL_WORLD_SIZE', 'GROUP_RANK', 'MASTER_ADDR',
              'MASTER_PORT', 'TORCHELASTIC_RUN_ID')


def visible_gpu_inds(environ, num_gpus):
    """The GPU index of every child (reference :57-65 reads CUDA_VISIBLE_DEVICES, else counts
    down from NUM_GPUS - 1).  On ROCm the list is HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES is
    honoured as its alias); an index may repeat - two children on one GPU - which is how the
    one-GPU box exercises this path."""
    vis = environ.get('HIP_VISIBLE_DEVICES') or environ.get('CUDA_VISIBLE_DEVICES')
    if vis:
        inds = [int(x) for x in vis.split(',')]
        assert -1 not in inds, 'Hiding GPU indices using the \'-1\' index is not supported'
        return inds[:num_gpus] if len(inds) >= num_gpus else inds
    return list(reversed(range(num_gpus)))


def split_ranges(total_range_size, parts):
    """[(start, end)) per child: np.array_split of range(total) (reference :55, :68-69); children
    whose share is empty are dropped (the reference would index an empty array)."""
    out = []
    for sub in np.array_split(np.arange(total_range_size), parts):
        if len(sub):
            out.append((int(sub[0]), int(sub[-1]) + 1))
    return out


def child_command(binary, start, end, cfg_file, opts):
    return [envu.python_binary(), binary, '--range', str(start), str(end), '--cfg', cfg_file,
            'NUM_GPUS', '1'] + [str(o) for o in opts]


def child_env(environ, gpu_ind):
    """The child's environment: the parent's without any rank variables (a child is a single
    process on one GPU, whatever launched the parent), one visible GPU, and the datasets
    registered at run time in the parent."""
    from detectron.datasets import dataset_catalog
    env = {k: v for k, v in environ.items() if k not in _RANK_VARS and k != 'CUDA_VISIBLE_DEVICES'}
    env['HIP_VISIBLE_DEVICES'] = str(gpu_ind)
    reg = dataset_catalog.registered()
    if reg:
        env['NAWS_DATASET_REGISTRY'] = json.dumps(reg)
    return env


def process_in_parallel(tag, total_range_size, binary, output_dir, opts=()):
    cfg_file = os.path.join(output_dir, '{}_range_config.yaml'.format(tag))
    with open(cfg_file, 'w') as f:
        envu.yaml_dump(cfg, stream=f)
    gpu_inds = visible_gpu_inds(os.environ, cfg.NUM_GPUS)
    ranges = split_ranges(total_range_size, len(gpu_inds))
    processes = []
    for i, ((start, end), gpu_ind) in enumerate(zip(ranges, gpu_inds)):
        cmd = child_command(binary, start, end, cfg_file, opts)
        logger.info('{} range command {}: {}'.format(tag, i, ' '.join(cmd)))
        if i == 0:
            out = _sp.PIPE
        else:
            out = open(os.path.join(output_dir, '%s_range_%s_%s.stdout' % (tag, start, end)), 'w')
        p = _sp.Popen(cmd, env=child_env(os.environ, gpu_ind), stdout=out, stderr=_sp.STDOUT,
                      bufsize=1, universal_newlines=True)
        processes.append((i, p, start, end, out))
    outputs = []
    for i, p, start, end, out in processes:
        log_subprocess_output(i, p, output_dir, tag, start, end)
        if i > 0:
            out.close()
        range_file = os.path.join(output_dir, '%s_range_%s_%s.pkl' % (tag, start, end))
        outputs.append(load_object(range_file))
    return outputs


def log_subprocess_output(i, p, output_dir, tag, start, end):
    """The first child's output in real time, the others' once they have finished, in order
    (reference :110-136).  A child that fails stops the parent."""
    outfile = os.path.join(output_dir, '%s_range_%s_%s.stdout' % (tag, start, end))
    logger.info('# ' + '-' * 76 + ' #')
    logger.info('stdout of subprocess %s with range [%s, %s]' % (i, start + 1, end))
    logger.info('# ' + '-' * 76 + ' #')
    if i == 0:
        with open(outfile, 'w') as f:
            for line in iter(p.stdout.readline, ''):
                print(line.rstrip())
                f.write(line)
        p.stdout.close()
        ret = p.wait()
    else:
        ret = p.wait()
        with open(outfile, 'r') as f:
            print(''.join(f.readlines()))
    sys.stdout.flush()
    assert ret == 0, 'Range subprocess failed (exit code: {})'.format(ret)
