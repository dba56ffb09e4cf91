#!/usr/bin/env python3
"""Golden fixture for the BEHAVIOUR of the config system ("same cfg keys,
unknown-key KeyError, immutability"): each case below runs on a fresh copy of the IMPORTED
reference's `detectron.core.config` (`merge_cfg_from_list` / `merge_cfg_from_cfg`,
config.py:1236-1420: value decoding, type coercion, deprecated and renamed keys, unknown keys) and
its outcome - the resulting value with its type, or the exception class and message - is written to
reference_cfg_behaviour.json.  Runs ONLY in the build container (needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_cfg_behaviour.py
"""
import copy
import json
import os
import sys

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_from_reference import REF, _StubFinder  # noqa: E402

LIST_CASES = [
    ['SOLVER.BASE_LR', '1'], ['SOLVER.BASE_LR', 1], ['SOLVER.BASE_LR', '0.5'], ['SOLVER.BASE_LR', 2.5],
    ['SOLVER.BASE_LR', 'fast'], ['SOLVER.BASE_LR', 'None'], ['MODEL.TYPE', 5], ['MODEL.TYPE', 'abc'],
    ['MODEL.TYPE', '(1,2)'], ['MODEL.TYPE', 'None'], ['TRAIN.SCALES', '[480, 600]'],
    ['TRAIN.SCALES', '(480,)'], ['TRAIN.SCALES', 480], ['TRAIN.SCALES', [480, 576]],
    ['TRAIN.DATASETS', 'a'], ['TRAIN.DATASETS', "('a', 'b')"], ['NUM_GPUS', '2.0'], ['NUM_GPUS', '8'],
    ['NUM_GPUS', True], ['TRAIN.USE_FLIPPED', 1], ['TRAIN.USE_FLIPPED', 'False'],
    ['TRAIN.USE_FLIPPED', 'false'], ['PIXEL_MEANS', '[[[1.0, 2.0, 3.0]]]'], ['PIXEL_MEANS', '[[[1, 2, 3]]]'],
    ['FINAL_MSG', 'x'], ['MODEL.DILATION', 2], ['TRAIN.DROPOUT', 0.5], ['TRAIN.DATASET', 'x'],
    ['MODEL.ROI_HEAD', 'x'], ['TEST.SCALES', '(600,)'], ['SOLVER.NOPE', 1], ['NOPE.X', 1],
    ['SOLVER.BASE_LR'], ['MODEL.NUM_CLASSES', '21'], ['EXPECTED_RESULTS', '[[1, 2]]'],
    ['OUTPUT_DIR', '/tmp/x'], ['OUTPUT_DIR', 7], ['RNG_SEED', 'None'], ['RNG_SEED', '11'],
    ['WSL.DILATION', '1'], ['WSL.ITER_SIZE', 2.0], ['SOLVER.STEPS', '[0, 150000]'],
    ['SOLVER.STEPS', '(0, 150000)'], ['SOLVER.GAMMA', '1e-1'], ['TEST.BBOX_AUG.SCALES', '[480]'],
]
DICT_CASES = [
    {'MODEL': {'NO_SUCH_KEY': 1}}, {'MODEL': {'DILATION': 1}}, {'MODEL': {'ROI_HEAD': 'x'}},
    {'TRAIN': {'SCALES': [480]}}, {'SOLVER': {'BASE_LR': 1}}, {'SOLVER': {'BASE_LR': 0.25}},
    {'SOLVER': 5}, {'TRAIN': {'DATASET': 'x'}}, {'FINAL_MSG': 'bye'}, {'NOPE': {'X': 1}},
    {'MODEL': {'TYPE': 3}}, {'TRAIN': {'DATASETS': ['a', 'b']}}, {'PIXEL_MEANS': [[[1, 2, 3]]]},
    {'WSL': {'NOPE': True}}, {'TEST': {'BBOX_AUG': {'SCALES': [480, 576], 'NOPE': 1}}},
]


def describe(v):
    if isinstance(v, np.ndarray):
        return {'type': 'ndarray', 'dtype': str(v.dtype), 'value': v.tolist()}
    if isinstance(v, tuple):
        return {'type': 'tuple', 'value': [describe(x) for x in v]}
    if isinstance(v, list):
        return {'type': 'list', 'value': [describe(x) for x in v]}
    if isinstance(v, dict):
        return {'type': 'dict', 'keys': sorted(v.keys())}
    return {'type': type(v).__name__, 'value': v}


def lookup(cfg, full_key):
    node = cfg
    for p in full_key.split('.'):
        node = node[p]
    return node


def first_leaf(d, stack=()):
    for k, v in d.items():
        if isinstance(v, dict):
            return first_leaf(v, stack + (k,))
        return '.'.join(stack + (k,))


def main():
    sys.dont_write_bytecode = True
    sys.meta_path.insert(0, _StubFinder())
    sys.path.insert(0, REF)
    import future.utils
    future.utils.iteritems = lambda d: iter(d.items())
    import detectron.utils.env as envu
    envu.yaml_load = lambda s: yaml.load(s, Loader=yaml.FullLoader)
    from detectron.core import config as rcfg
    from detectron.utils.collections import AttrDict
    pristine = copy.deepcopy(rcfg.__C)

    def fresh():
        rcfg.__C.immutable(False)
        for k in list(rcfg.__C.keys()):
            del rcfg.__C[k]
        for k, v in copy.deepcopy(pristine).items():
            rcfg.__C[k] = v

    def attr(d):
        return AttrDict({k: attr(v) if isinstance(v, dict) else v for k, v in d.items()})

    out = {'list': [], 'dict': []}
    for case in LIST_CASES:
        fresh()
        rec = {'args': case}
        try:
            rcfg.merge_cfg_from_list(list(case))
            try:
                rec['result'] = describe(lookup(rcfg.__C, case[0]))
            except KeyError:
                rec['result'] = {'type': 'absent'}          # deprecated keys are ignored
        except BaseException as e:          # noqa: B902
            rec['error'] = [type(e).__name__, str(e)]
        out['list'].append(rec)
    for case in DICT_CASES:
        fresh()
        rec = {'args': case}
        try:
            rcfg.merge_cfg_from_cfg(attr(case))
            key = first_leaf(case)
            try:
                rec['result'] = describe(lookup(rcfg.__C, key))
            except KeyError:
                rec['result'] = {'type': 'absent'}
        except BaseException as e:          # noqa: B902
            rec['error'] = [type(e).__name__, str(e)]
        out['dict'].append(rec)
    # immutability (config.py / collections.py AttrDict.immutable)
    fresh()
    rcfg.assert_and_infer_cfg(cache_urls=False)
    imm = []
    for stmt in ('cfg.NUM_GPUS = 8', 'cfg.TRAIN.SCALES = (1,)', 'cfg.NEW_KEY = 1'):
        try:
            exec(stmt, {'cfg': rcfg.cfg})
            imm.append([stmt, None])
        except BaseException as e:          # noqa: B902
            imm.append([stmt, type(e).__name__])
    out['immutable'] = imm
    with open(os.path.join(HERE, 'reference_cfg_behaviour.json'), 'w') as f:
        json.dump(out, f, indent=1)
    for group in ('list', 'dict'):
        for rec in out[group]:
            print(rec['args'], '->', rec.get('result', rec.get('error')))
    print(imm)


if __name__ == '__main__':
    main()
