#!/usr/bin/env python3
"""Generate golden fixtures by IMPORTING the reference's pure-Python code.

Runs ONLY in the build container (needs /root/reference; the GPU box has neither the
reference nor this script's imports).  Its outputs, committed next to it, are data:
  reference_python.json   cfg values after merging the na_wsddn yaml, lr schedule samples,
                          the recorded op trace of the reference graph builders
                          (train + test mode)
  reference_rois.npz      input/output pairs of roi_data.wsl._project_im_rois and
                          _sample_rois on seeded inputs

Third-party packages the reference imports but this image lacks (caffe2, cv2, future,
past, pycocotools, the two cython extensions) are replaced by MagicMock modules — none of
the functions exercised below calls into them.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_from_reference.py
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
from unittest import mock

import numpy as np
import yaml

REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))
STUB_PREFIXES = ('caffe2', 'cv2', 'future', 'past', 'pycocotools',
                 'detectron.utils.cython_bbox', 'detectron.utils.cython_nms')


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if any(name == p or name.startswith(p + '.') for p in STUB_PREFIXES):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)
        return None

    def create_module(self, spec):
        m = mock.MagicMock()
        m.__path__ = []
        m.__name__ = spec.name
        m.__spec__ = spec
        return m

    def exec_module(self, module):
        pass


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    cfg = rcfg.cfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', 4])

    out = {}
    # ---- cfg values (only sub-trees the hot path reads)
    def plain(v):
        if isinstance(v, np.ndarray):
            return v.tolist()
        if isinstance(v, dict):
            return {k: plain(x) for k, x in v.items()}
        if isinstance(v, tuple):
            return list(v)
        return v
    keep = ['MODEL', 'TRAIN', 'TEST', 'SOLVER', 'FAST_RCNN', 'WSL', 'WEBLY', 'DATA_LOADER']
    tree = {}
    for k in keep:
        tree[k] = plain(dict(cfg[k]))
    for k in ('NUM_GPUS', 'USE_NCCL', 'DEDUP_BOXES', 'PIXEL_MEANS', 'RNG_SEED', 'MEMONGER', 'VIS',
              'VIS_TH', 'OUTPUT_DIR'):
        tree[k] = plain(cfg[k])
    out['cfg'] = tree

    # ---- lr schedule
    from detectron.utils import lr_policy
    its = [0, 1, 499, 500, 149999, 150000, 150001, 199999]
    out['lr'] = {str(i): float(lr_policy.get_lr_at_iter(i)) for i in its}
    out['lr_bits'] = {str(i): int(np.float32(lr_policy.get_lr_at_iter(i)).view(np.uint32))
                      for i in its}

    # ---- roi projection / sampling
    from detectron.roi_data import wsl as rwsl
    rng = np.random.RandomState(11)
    boxes = np.round(rng.uniform(0, 500, (40, 4))).astype(np.float32)
    boxes[:, 2:] = boxes[:, :2] + np.round(rng.uniform(21, 300, (40, 2))).astype(np.float32)
    crop = np.array([12, 30, 460, 371], np.int32)           # x1,y1,x2,y2
    proj = rwsl._project_im_rois(boxes.copy(), 1.171875, crop)
    entry = dict(boxes=boxes.copy(), obn_scores=rng.uniform(0, 1, (40, 1)).astype(np.float32),
                 gt_classes=np.zeros((40,), np.int32))
    entry['gt_classes'][0] = 7
    entry['gt_classes'][5] = 3
    blob = rwsl._sample_rois(entry, 1.171875, crop, 2)
    np.savez(os.path.join(HERE, 'reference_rois.npz'), boxes=boxes, crop=crop,
             scale=np.float64(1.171875), projected=proj, obn_scores=entry['obn_scores'],
             gt_classes=entry['gt_classes'], s_rois=blob['rois'], s_obn=blob['obn_scores'],
             s_labels_int32=blob['labels_int32'], s_labels_oh=blob['labels_oh'],
             batch_size_per_im=np.int64(cfg.TRAIN.BATCH_SIZE_PER_IM))

    # ---- graph trace of the reference builders (recording model, Appendix A.2)
    from detectron.modeling import VGG16, wsl_heads, webly_heads
    for mod in (wsl_heads, webly_heads):
        mod.const_fill = lambda v: ('ConstantFill', {'value': v})
        mod.gauss_fill = lambda s: ('GaussianFill', {'std': s})

    def trace(train):
        ops = []

        class Rec(object):
            def __init__(self):
                self.train = train
                self.num_classes = cfg.MODEL.NUM_CLASSES
                self.losses, self.metrics = [], []
                self.net = self
                self.param_init_net = self

            def AddLosses(self, l):
                self.losses += l if isinstance(l, list) else [l]

            def AddMetrics(self, m):
                self.metrics += m if isinstance(m, list) else [m]

            def RoIFeatureTransform(self, blobs_in, blob_out, blob_rois='rois', method='RoIPoolF',
                                    resolution=7, spatial_scale=1. / 16., sampling_ratio=0):
                ops.append([method, [blobs_in, blob_rois], [blob_out, '_argmax_' + blob_out],
                            {'pooled_h': resolution, 'pooled_w': resolution,
                             'spatial_scale': spatial_scale}])
                return blob_out

            def __getattr__(self, op):
                def f(ins, outs=None, *a, **kw):
                    ins_l = ins if isinstance(ins, list) else [ins]
                    o = outs if outs is not None else ins
                    outs_l = o if isinstance(o, list) else [o]
                    extra = {}
                    if op in ('Conv', 'FC'):
                        extra = {'dims': [int(x) for x in a[:3]]}
                    kws = {k: (v if isinstance(v, (int, float, str, bool, list, tuple)) else str(v))
                           for k, v in kw.items() if k not in ('weight_init', 'bias_init')}
                    kws.update(extra)
                    ops.append([op, [str(x) for x in ins_l], [str(x) for x in outs_l], kws])
                    return outs_l[0] if len(outs_l) == 1 else tuple(outs_l)
                return f

        m = Rec()
        blob, dim, scale = VGG16.add_VGG16_conv5_body_origin(m)
        m.StopGradient(blob, blob)
        ls, dims = webly_heads.add_VGG16_roi_2fc_noise_head(m, blob, dim, scale)
        webly_heads.add_webly_outputs(m, ls, dims)
        if train:
            webly_heads.add_webly_losses(m)
        return dict(ops=ops, losses=m.losses, metrics=m.metrics, body=[str(blob), dim, scale],
                    head=[[str(x) for x in ls], dims])

    out['trace_train'] = trace(True)
    out['trace_test'] = trace(False)
    # the same head with cfg.WSL.MIN_ENTROPY_LOSS (webly_heads.py:208-214): only the tail differs
    cfg.immutable(False) if hasattr(cfg, 'immutable') else None
    cfg.WSL.MIN_ENTROPY_LOSS = True
    t2 = trace(True)
    cfg.WSL.MIN_ENTROPY_LOSS = False
    n0 = len(out['trace_train']['ops'])
    assert t2['ops'][:n0] == out['trace_train']['ops']
    out['trace_train_min_entropy_tail'] = dict(ops=t2['ops'][n0:], losses=t2['losses'])
    with open(os.path.join(HERE, 'reference_python.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print('train ops', len(out['trace_train']['ops']), 'test ops', len(out['trace_test']['ops']))


if __name__ == '__main__':
    main()
