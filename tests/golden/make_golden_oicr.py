#!/usr/bin/env python3
"""Op traces of the reference graph builders for the SURVEY.md §8 f-4 paths (build container only;
same stub-import harness as make_golden_from_reference.py):

  reference_oicr.json
    wsl_oicr_train / wsl_oicr_test   generalized_wsl WITHOUT the webly head (WEBLY.WEBLY_ON False,
                                     ROI_BOX_HEAD wsl_heads.add_VGG16_roi_2fc_head) and WSL.OICR:
                                     conv body, 2-fc head, wsl_heads.add_wsl_outputs (+ the three
                                     cls_score refinement branches, wsl_heads.py:134-156) and
                                     add_wsl_losses -> add_oicr_losses (:512-560)
    entropy_weight                   webly_heads.add_entropy_weight (:219-262) on its own (the
                                     reference never calls it: the call at :131 is commented out)

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_oicr.py
"""
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden_from_reference as base  # noqa: E402

REF = base.REF


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, base._StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    cfg = rcfg.cfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', 4, 'WEBLY.WEBLY_ON', False, 'WSL.OICR', True,
                              'FAST_RCNN.ROI_BOX_HEAD', 'wsl_heads.add_VGG16_roi_2fc_head'])
    from detectron.modeling import VGG16, wsl_heads, webly_heads
    import detectron.utils.blob as blob_utils
    for mod in (wsl_heads, webly_heads):
        mod.const_fill = lambda v: ('ConstantFill', {'value': v})
        mod.gauss_fill = lambda s: ('GaussianFill', {'std': s})

    def recorder(train):
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
                    if op in ('FC',) and 'weight_init' in kw:
                        kws['weight_init'] = list(kw['weight_init'])
                    kws.update(extra)
                    if 'uuid' in kws:
                        kws['uuid'] = 0            # random per build (uuid4): not part of the graph
                    ops.append([op, [str(x) for x in ins_l], [str(x) for x in outs_l], kws])
                    return outs_l[0] if len(outs_l) == 1 else tuple(outs_l)
                return f

        return Rec(), ops

    out = {}
    for train in (True, False):
        m, ops = recorder(train)
        blob, dim, scale = VGG16.add_VGG16_conv5_body_origin(m)
        m.StopGradient(blob, blob)
        blob_frcn, dim_frcn = wsl_heads.add_VGG16_roi_2fc_head(m, blob, dim, scale)
        wsl_heads.add_wsl_outputs(m, blob_frcn, dim_frcn)
        lg = wsl_heads.add_wsl_losses(m) if train else None
        out['wsl_oicr_train' if train else 'wsl_oicr_test'] = dict(
            ops=ops, losses=m.losses, metrics=m.metrics,
            loss_gradients=sorted(lg) if lg else None)
    m, ops = recorder(True)
    w = webly_heads.add_entropy_weight(m, 'rois_pred', 'rois')
    out['entropy_weight'] = dict(ops=ops, weight=str(w))
    with open(os.path.join(HERE, 'reference_oicr.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    for k, v in out.items():
        print(k, len(v['ops']), 'ops;', [o[0] for o in v['ops'][-16:]])


if __name__ == '__main__':
    main()
