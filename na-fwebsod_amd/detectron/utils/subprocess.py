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
    """The visible-device token of every child (reference :57-65 reads CUDA_VISIBLE_DEVICES, else
    counts down from NUM_GPUS - 1).  On ROCm the list is HIP_VISIBLE_DEVICES (CUDA_VISIBLE_DEVICES
    is honoured as its alias; ROCR_VISIBLE_DEVICES, which renumbers the devices HIP sees, is
    consulted when neither is set - then the children get HIP indices 0..n-1 of that renumbered
    set).  Tokens are passed through as STRINGS: an entry may be an index or a `GPU-<uuid>`.  An
    index may repeat - two children on one GPU - which is how the one-GPU box exercises this path.
    A list shorter than NUM_GPUS starts fewer children, with a warning."""
    vis = environ.get('HIP_VISIBLE_DEVICES') or environ.get('CUDA_VISIBLE_DEVICES')
    if vis:
        inds = [x.strip() for x in vis.split(',') if x.strip()]
        assert '-1' not in inds, 'Hiding GPU indices using the \'-1\' index is not supported'
        inds = [int(x) if x.isdigit() else x for x in inds]
    elif environ.get('ROCR_VISIBLE_DEVICES'):
        n = len([x for x in environ['ROCR_VISIBLE_DEVICES'].split(',') if x.strip()])
        inds = list(reversed(range(min(n, num_gpus))))
    else:
        return list(reversed(range(num_gpus)))
    if len(inds) < num_gpus:
        logger.warning('NUM_GPUS is %d but only %d device(s) are visible (%s): starting %d '
                       'child(ren)', num_gpus, len(inds), ','.join(str(i) for i in inds), len(inds))
    return inds[:num_gpus]


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
    try:
        for i, p, start, end, out in processes:
            log_subprocess_output(i, p, output_dir, tag, start, end,
                                  others=[q for _j, q, _s, _e, _o in processes if q is not p])
            if i > 0:
                out.close()
            range_file = os.path.join(output_dir, '%s_range_%s_%s.pkl' % (tag, start, end))
            outputs.append(load_object(range_file))
    except BaseException:
        # a failed child (or an interrupt): no sibling is left running on its GPU
        terminate_all([p for _i, p, _s, _e, _o in processes])
        raise
    return outputs


def terminate_all(procs, grace=10.0):
    """terminate(), then kill() whatever has not exited after `grace` seconds."""
    import time
    live = [p for p in procs if p.poll() is None]
    for p in live:
        p.terminate()
    t0 = time.time()
    for p in live:
        try:
            p.wait(timeout=max(0.1, grace - (time.time() - t0)))
        except _sp.TimeoutExpired:
            p.kill()
            p.wait()


def first_failed(procs):
    for p in procs:
        rc = p.poll()
        if rc is not None and rc != 0:
            return rc
    return None


def log_subprocess_output(i, p, output_dir, tag, start, end, others=()):
    """The first child's output in real time, the others' once they have finished, in order
    (reference :110-136).  A child that fails stops the parent - also a LATER child while an
    earlier one is still running (`others` are polled while this one is waited for)."""
    outfile = os.path.join(output_dir, '%s_range_%s_%s.stdout' % (tag, start, end))
    logger.info('# ' + '-' * 76 + ' #')
    logger.info('stdout of subprocess %s with range [%s, %s]' % (i, start + 1, end))
    logger.info('# ' + '-' * 76 + ' #')

    def check_others():
        rc = first_failed(others)
        assert rc is None, 'Range subprocess failed (exit code: {})'.format(rc)
    if i == 0:
        with open(outfile, 'w') as f:
            for line in iter(p.stdout.readline, ''):
                print(line.rstrip())
                f.write(line)
                check_others()
        p.stdout.close()
        ret = p.wait()
    else:
        while True:
            try:
                ret = p.wait(timeout=1.0)
                break
            except _sp.TimeoutExpired:
                check_others()
        with open(outfile, 'r') as f:
            print(''.join(f.readlines()))
    sys.stdout.flush()
    assert ret == 0, 'Range subprocess failed (exit code: {})'.format(ret)
