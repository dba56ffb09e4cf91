#!/usr/bin/env python3
"""Golden fixture for the learning-rate feed + momentum correction of the training loop, captured
from the IMPORTED reference (`detectron/modeling/detector.py:509-559, 581-586`):
`DetectionModelHelper.UpdateWorkspaceLr` is driven through a sequence of learning rates with a
recording workspace (caffe2 is a MagicMock: FetchBlob / FeedBlob / CreateOperator are replaced by
recorders), and every call's outcome - the lr fed, whether `_CorrectMomentum` ran and with which
`Scale` factor - is written to reference_lr_update.json.

Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_lr_update.py
"""
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_from_reference import REF, _StubFinder  # noqa: E402


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    import caffe2.python.cnn as c2cnn                 # (a MagicMock module)
    c2cnn.CNNModelHelper = type('CNNModelHelper', (object,), {})   # a real base class for the helper
    import detectron.modeling.detector as det
    cfg = rcfg.cfg
    rcfg.merge_cfg_from_file(os.path.join(REF, 'configs/flickr_voc/na_wsddn_V-16-C5_1x.yaml'))
    rcfg.merge_cfg_from_list(['NUM_GPUS', 1])

    state = {'lr': np.float32(0.0)}
    events = []

    class _WS(object):
        @staticmethod
        def FetchBlob(name):
            assert name == 'gpu_0/lr'
            return np.array([state['lr']], dtype=np.float32)

        @staticmethod
        def FeedBlob(name, arr):
            assert name == 'gpu_0/lr' and arr.dtype == np.float32
            state['lr'] = arr[0]
            events.append(('feed', float(arr[0])))

        @staticmethod
        def RunOperatorOnce(op):
            events.append(('scale', op))

    class _Core(object):
        @staticmethod
        def CreateOperator(kind, ins, outs, **kw):
            assert kind == 'Scale' and ins == outs
            return (ins[0], float(kw['scale']))

    det.workspace = _WS
    det.core = _Core
    import contextlib
    det.c2_utils.CudaScope = lambda i: contextlib.nullcontext()

    class _Helper(object):
        UpdateWorkspaceLr = det.DetectionModelHelper.UpdateWorkspaceLr
        _SetNewLr = det.DetectionModelHelper._SetNewLr
        _CorrectMomentum = det.DetectionModelHelper._CorrectMomentum

        def TrainableParams(self, gpu_id=-1):
            return ['gpu_0/fc6_w', 'gpu_0/fc6_b']

    h = _Helper()
    # the schedule's own values (warm-up off in this yaml: 1e-3 until 150k, then 1e-4) plus the
    # edge cases of the rule: no change, a change inside the 1.1 threshold, from ~0, to a larger lr
    seq = [1e-3, 1e-3, 1e-4, 1.05e-4, 1.2e-4, 1e-8, 5e-8, 1e-3, 3.3333334e-4, 0.0, 1e-3]
    cases = []
    for it, lr in enumerate(seq):
        before = float(state['lr'])
        del events[:]
        ret = h.UpdateWorkspaceLr(it, np.float32(lr))
        scales = [e[1] for e in events if e[0] == 'scale']
        cases.append(dict(cur_lr=before, new_lr=float(np.float32(lr)), returned=float(ret),
                          fed=[e[1] for e in events if e[0] == 'feed'],
                          momentum_scaled=sorted(set(s[0] for s in scales)),
                          correction=(scales[0][1] if scales else None),
                          ratio=float(det._get_lr_change_ratio(np.float32(before), np.float32(lr)))))
    out = dict(SCALE_MOMENTUM=bool(cfg.SOLVER.SCALE_MOMENTUM),
               SCALE_MOMENTUM_THRESHOLD=float(cfg.SOLVER.SCALE_MOMENTUM_THRESHOLD), cases=cases)
    with open(os.path.join(HERE, 'reference_lr_update.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for c in cases:
        print(c)


if __name__ == '__main__':
    main()
