#!/usr/bin/env python3
"""Checkpoint-format fixtures from the IMPORTED reference Python (runs only in
the build container, reusing make_golden_from_reference.py's stub-import harness):

  reference_checkpoint.pkl        what the reference's own `save_model_to_weights_file`
                                  (detectron/utils/net_wsl.py:140-180) writes for a small model state
                                  (parameters incl. a '_[noisy]_' twin, momentum, a preserved blob,
                                  the cfg dump) - the pickle as it lands on disk
  reference_checkpoint_load.npz   the blobs the reference's `initialize_gpu_from_weights_file`
                                  (:51-137) FEEDS for a pretrained-style file: the '_[xyz]_foo' <- 'foo'
                                  fallback, momentum only where the file has it, float64 sources cast
                                  to float32, unused blobs kept under '__preserve__/'

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_checkpoint.py
"""
import os
import sys
from unittest import mock

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_from_reference as base  # noqa: E402

REF = base.REF


class _Model(object):
    """The three accessors the reference functions call (detectron/modeling/detector.py)."""
    params = ['gpu_0/fc6_w', 'gpu_0/fc6_b', 'gpu_0/_[noisy]_fc6_w', 'gpu_0/_[noisy]_fc6_b',
              'gpu_0/fc8c_w', 'gpu_0/_[noisy]_fc7_w']       # (no plain fc7_w in the model)

    def TrainableParams(self):
        return list(self.params)

    def GetComputedParams(self):
        return []


def state(seed):
    rng = np.random.RandomState(seed)
    return {n: rng.standard_normal(s).astype(np.float32) for n, s in (
        ('fc6_w', (4, 6)), ('fc6_b', (4,)), ('_[noisy]_fc6_w', (4, 6)), ('_[noisy]_fc6_b', (4,)),
        ('fc8c_w', (3, 4)), ('_[noisy]_fc7_w', (2, 2)))}


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, base._StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    from detectron.utils import net_wsl as rn
    import detectron.utils.c2 as rc2
    # caffe2.python.scope._NAMESCOPE_SEPARATOR (Caffe2's published constant) is a stub here
    rc2.scope._NAMESCOPE_SEPARATOR = '/'

    # ---------------------------------------------------------------- save
    ws = {}
    st = state(3)
    for n, v in st.items():
        ws['gpu_0/' + n] = v
        ws['gpu_0/' + n + '_momentum'] = (v * 0.5).astype(np.float32)
    ws['__preserve__/fc1000_w'] = np.arange(8, dtype=np.float32).reshape(4, 2)
    rn.workspace = mock.MagicMock()
    rn.workspace.FetchBlob = lambda name: ws[str(name)]
    rn.workspace.Blobs = lambda: list(ws)
    out_pkl = os.path.join(HERE, 'reference_checkpoint.pkl')
    rn.save_model_to_weights_file(out_pkl, _Model())

    # ---------------------------------------------------------------- load
    rng = np.random.RandomState(5)
    pre = {'fc6_w': rng.standard_normal((4, 6)),                       # float64, as old pickles hold
           'fc6_b': rng.standard_normal((4,)).astype(np.float32),
           'fc6_w_momentum': rng.standard_normal((4, 6)).astype(np.float32),
           'fc8c_w': rng.standard_normal((3, 4)).astype(np.float32),
           'fc7_w': rng.standard_normal((2, 2)).astype(np.float32),       # only reachable through the alias
           'fc1000_w': np.arange(8, dtype=np.float32).reshape(4, 2),
           'fc1000_w_momentum': np.ones((4, 2), np.float32),            # momentum of an unused blob: dropped
           'unused_none': None}
    src = os.path.join(HERE, '_pretrained_tmp.pkl')
    from detectron.utils.io import save_object
    save_object({'blobs': pre}, src)
    fed = {}
    rn.workspace = mock.MagicMock()
    rn.workspace.Blobs = lambda: []
    rn.workspace.FeedBlob = lambda name, v: fed.__setitem__(str(name), np.array(v))
    rn.core = mock.MagicMock()
    rn.core.ScopedName = lambda n: 'gpu_0/' + n
    rn.initialize_gpu_from_weights_file(_Model(), src, gpu_id=0)
    os.remove(src)
    out = {'src/' + k: v for k, v in pre.items() if v is not None}
    for k, v in fed.items():
        out['fed/' + k] = v
        out['fed_dtype/' + k] = np.array(str(v.dtype))
    np.savez_compressed(os.path.join(HERE, 'reference_checkpoint_load.npz'), **out)
    print('saved blobs:', sorted(__import__('pickle').load(open(out_pkl, 'rb'), encoding='latin1')['blobs']))
    print('fed:', sorted(fed))


if __name__ == '__main__':
    main()
