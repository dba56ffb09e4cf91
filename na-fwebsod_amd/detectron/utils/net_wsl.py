"""Weight files.  Same pickle schema as the reference (detectron/utils/net_wsl.py:140-180
`save_model_to_weights_file`, :46-137 `initialize_gpu_from_weights_file`, io.py:39-83):
`{'blobs': {unscoped_name: ndarray, name + '_momentum': ndarray, ...}, 'cfg': yaml}`, FC
weights [out,in], conv weights [Cout,Cin,kh,kw], loaded with latin1 for py2 pickles."""
import logging
import os
import pickle
import tempfile

import numpy as np
import yaml

from detectron.core.config import cfg

logger = logging.getLogger(__name__)


def load_object(file_name):
    with open(file_name, 'rb') as f:
        return pickle.load(f, encoding='latin1')


def save_object(obj, file_name, pickle_format=2):
    """Atomic write (io.py:39-70)."""
    file_name = os.path.abspath(file_name)
    fd, tmp = tempfile.mkstemp(dir=os.path.dirname(file_name), suffix='.tmp')
    with os.fdopen(fd, 'wb') as f:
        pickle.dump(obj, f, pickle_format)
    os.replace(tmp, file_name)


def resolve_source_name(dst_name, src_blobs):
    """Which blob of the weights file initialises `dst_name`.  A ']_' in the name aliases the
    part after it: '_[noisy]_fc6_w' is initialised from 'fc6_w' when the file has no blob of
    its own name (net_wsl.py:79-87)."""
    if dst_name in src_blobs:
        return dst_name
    pos = dst_name.find(']_')
    if pos >= 0 and dst_name[pos + 2:] in src_blobs:
        return dst_name[pos + 2:]
    return None


def initialize_from_weights_file(model, weights_file, executor, broadcast=True):
    import torch
    src = load_object(weights_file)
    src_blobs = src['blobs'] if 'blobs' in src else src
    blobs, preserved = {}, {}
    have = executor.blobs(with_momentum=False)
    for name in model.params:
        s = resolve_source_name(name, src_blobs)
        if s is None:
            logger.info('{:s} not found in the weights file: keeping its initialisation'.format(name))
            blobs[name] = have[name]
            continue
        arr = np.asarray(src_blobs[s], dtype=np.float32)
        assert tuple(arr.shape) == tuple(model.param_shapes[name]), \
            'Workspace blob {} with shape {} does not match weights file shape {}'.format(
                name, model.param_shapes[name], arr.shape)
        blobs[name] = torch.from_numpy(arr)
        if s + '_momentum' in src_blobs:
            blobs[name + '_momentum'] = torch.from_numpy(
                np.asarray(src_blobs[s + '_momentum'], np.float32))
    # blobs of the file that are not parameters OF THIS MODEL BY NAME are kept for re-saving
    # (:129-137) - including one that only initialised a '_[xyz]_' twin through the alias rule
    names = set(model.params)
    for k, v in src_blobs.items():
        if k not in names and not k.endswith('_momentum') and v is not None:
            preserved['__preserve__/' + k] = v
    model.preserved_blobs = preserved
    executor.load_blobs(blobs)
    if broadcast:
        executor.broadcast_parameters()


def save_model_to_weights_file(weights_file, model, executor):
    logger.info('Saving parameters and momentum to {}'.format(os.path.abspath(weights_file)))
    blobs = {}
    for k, v in executor.blobs(with_momentum=True).items():
        if k in model.params or (k.endswith('_momentum') and
                                 k[:-len('_momentum')] in model.TrainableParams()):
            blobs[k] = v.detach().cpu().numpy()
    # preserved blobs are saved under their UNSCOPED name ('__preserve__/fc1000_w' -> 'fc1000_w',
    # net_wsl.py:170-178 / c2.py:97-102), never over a blob saved above
    for k, v in getattr(model, 'preserved_blobs', {}).items():
        unscoped = k[k.rfind('/') + 1:]
        if unscoped not in blobs:
            blobs[unscoped] = v
    # (the reference reads this string back as an AttrDict: net_wsl.py:64-66, :277)
    import detectron.utils.env as envu
    save_object(dict(blobs=blobs, cfg=envu.yaml_dump(cfg, reference_format=True)), weights_file)


def _plain(node):
    if isinstance(node, dict):
        return {k: _plain(v) for k, v in node.items()}
    if isinstance(node, np.ndarray):
        return node.tolist()
    if isinstance(node, tuple):
        return list(node)
    return node
