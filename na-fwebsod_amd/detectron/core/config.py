"""Global config for the MI355X-native NA-fWebSOD hot path.

Plugin surface mirrored from the reference (detectron/core/config.py:60-1105 for
the key tree, :1178-1396 for the merge helpers): a module-level `cfg` tree that
is filled from a yaml file (`merge_cfg_from_file`) and `KEY VALUE` pairs
(`merge_cfg_from_list`), type-checked against the defaults, and frozen by
`assert_and_infer_cfg`.  Only the sub-trees the hot path reads
are declared; a key outside them raises KeyError exactly like an unknown key
does in the reference, so a yaml written for another model family fails loudly
rather than being half-applied.
"""
import ast
import copy
import os

import numpy as np
import yaml


class CfgNode(dict):
    """dict with attribute access and a recursive read-only switch."""

    _FROZEN = '__frozen__'

    def __init__(self, init=None):
        super(CfgNode, self).__init__()
        self.__dict__[CfgNode._FROZEN] = False
        for k, v in (init or {}).items():
            self[k] = CfgNode(v) if isinstance(v, dict) and not isinstance(v, CfgNode) else v

    def __getattr__(self, name):
        if name in self:
            return self[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if self.__dict__[CfgNode._FROZEN]:
            raise AttributeError(
                'Attempted to set "{}" to "{}", but the config is immutable'.format(name, value))
        self[name] = value

    def immutable(self, flag):
        self.__dict__[CfgNode._FROZEN] = flag
        for v in self.values():
            if isinstance(v, CfgNode):
                v.immutable(flag)

    def is_immutable(self):
        return self.__dict__[CfgNode._FROZEN]

    def __deepcopy__(self, memo):
        out = CfgNode()
        for k, v in self.items():
            out[k] = copy.deepcopy(v, memo)
        return out


AttrDict = CfgNode  # the reference's name for this class (detectron/utils/collections.py)


def _defaults():
    return {
        'MODEL': {
            'TYPE': '', 'CONV_BODY': '', 'NUM_CLASSES': -1, 'MASK_ON': False,
            'KEYPOINTS_ON': False, 'RPN_ONLY': False, 'FASTER_RCNN': False,
            'EXECUTION_TYPE': 'dag',
            # read back by the reference's weight-file loader (net_wsl.py:274-290); unused here (no
            # bbox regression on this path)
            'BBOX_REG_WEIGHTS': (10., 10., 5., 5.),
        },
        'TRAIN': {
            'WEIGHTS': '', 'DATASETS': (), 'PROPOSAL_FILES': (), 'SCALES': (600,),
            'MAX_SIZE': 1000, 'IMS_PER_BATCH': 2, 'BATCH_SIZE_PER_IM': 64,
            'USE_FLIPPED': True, 'SNAPSHOT_ITERS': 80000, 'ASPECT_GROUPING': True,
            'CROWD_FILTER_THRESH': 0.7, 'GT_MIN_AREA': -1, 'FG_THRESH': 0.5, 'BG_THRESH_HI': 0.5,
            'BG_THRESH_LO': 0.0, 'FREEZE_CONV_BODY': False,
            'AUTO_RESUME': True, 'COPY_WEIGHTS': False, 'FREEZE_AT': 2,
        },
        'DATA_LOADER': {'NUM_THREADS': 4, 'MINIBATCH_QUEUE_SIZE': 64, 'BLOBS_QUEUE_CAPACITY': 8},
        'TEST': {
            'WEIGHTS': '', 'DATASETS': (), 'SCALE': 600, 'MAX_SIZE': 1000, 'NMS': 0.3,
            'BBOX_REG': True, 'PROPOSAL_FILES': (), 'PROPOSAL_LIMIT': 2000,
            'DETECTIONS_PER_IM': 100, 'SCORE_THRESH': 0.05, 'COMPETITION_MODE': True,
            'FORCE_JSON_DATASET_EVAL': False, 'PRECOMPUTED_PROPOSALS': True,
            'BBOX_AUG': {
                'ENABLED': False, 'SCORE_HEUR': 'UNION', 'COORD_HEUR': 'UNION', 'H_FLIP': False,
                'SCALES': (), 'MAX_SIZE': 4000, 'SCALE_H_FLIP': False, 'SCALE_SIZE_DEP': False,
                'AREA_TH_LO': 50 ** 2, 'AREA_TH_HI': 180 ** 2, 'ASPECT_RATIOS': (),
                'ASPECT_RATIO_H_FLIP': False,
            },
            # core/config.py:408-435: soft NMS (overlap threshold = TEST.NMS) and box voting
            'SOFT_NMS': {'ENABLED': False, 'METHOD': 'linear', 'SIGMA': 0.5},
            'BBOX_VOTE': {'ENABLED': False, 'VOTE_TH': 0.8, 'SCORING_METHOD': 'ID',
                          'SCORING_METHOD_BETA': 1.0},
        },
        'SOLVER': {
            'BASE_LR': 0.001, 'LR_POLICY': 'step', 'GAMMA': 0.1, 'STEP_SIZE': 30000,
            'STEPS': [], 'LRS': [], 'MAX_ITER': 40000, 'MOMENTUM': 0.9, 'WEIGHT_DECAY': 0.0005,
            'WEIGHT_DECAY_GN': 0.0, 'WARM_UP_ITERS': 500, 'WARM_UP_FACTOR': 1.0 / 3.0,
            'WARM_UP_METHOD': 'linear', 'SCALE_MOMENTUM': True, 'SCALE_MOMENTUM_THRESHOLD': 1.1,
            'LOG_LR_CHANGE_THRESHOLD': 1.1,
        },
        'FAST_RCNN': {
            'ROI_BOX_HEAD': '', 'MLP_HEAD_DIM': 1024, 'ROI_XFORM_METHOD': 'RoIPoolF',
            'ROI_XFORM_SAMPLING_RATIO': 0, 'ROI_XFORM_RESOLUTION': 14,
        },
        # other model families: only their on/off switches exist, and must stay off
        'RPN': {'RPN_ON': False},
        'FPN': {'FPN_ON': False, 'MULTILEVEL_ROIS': False},
        'RETINANET': {'RETINANET_ON': False},
        'MRCNN': {'ROI_MASK_HEAD': ''},
        'KRCNN': {'ROI_KEYPOINTS_HEAD': ''},
        'WSL': {
            'WSL_ON': False, 'ITER_SIZE': 1, 'DEBUG': False, 'SAMPLE': False, 'SAMPLE_ITER': 1280,
            'CPG': False, 'CSC': False, 'CENTER_LOSS': False, 'CONTEXT': False, 'OICR': False,
            'PCL': False, 'CMIL': False, 'MEAN_LOSS': False, 'USE_DISTORTION': True,
            'SATURATION': 1.5, 'EXPOSURE': 1.5, 'USE_CROP': True, 'CROP': 0.9, 'DILATION': 1,
            'MIN_ENTROPY_LOSS': False,
        },
        'WEBLY': {
            'WEBLY_ON': False, 'ENTROPY': False, 'MINING': False, 'BAGGING_MIXUP': False,
            'BAGGING_MIXUP_ALPHA': 1.5,
        },
        'NUM_GPUS': 1,
        'USE_NCCL': False,
        'DEDUP_BOXES': 1 / 16.,
        'PIXEL_MEANS': np.array([[[102.9801, 115.9465, 122.7717]]]),
        'PIXEL_STDS': np.array([[[1.0, 1.0, 1.0]]]),
        'RNG_SEED': 3,
        'EPS': 1e-14,
        'ROOT_DIR': os.getcwd(),
        'OUTPUT_DIR': '/tmp',
        'MEMONGER': False,
        'MEMONGER_SHARE_ACTIVATIONS': False,
        'VIS': False,
        'VIS_TH': 0.9,
        'EXPECTED_RESULTS': [],
        'EXPECTED_RESULTS_RTOL': 0.1,
        'EXPECTED_RESULTS_ATOL': 0.005,
        # MI355X-native additions (not in the reference)
        'NAWS': {
            'IMS_PER_GPU': 1,        # images per GPU process (the reference supports only 1)
            'ALLREDUCE_CHUNKS': 0,   # 0 = auto (4 at world_size 2, 2 above, 1 alone); >1: cut fc6 wgrad into row chunks, each all-reduced while
                                     # the next chunk's GEMM runs (default: one launch; the
                                     # collective hides under the next iteration's conv body)
            'DEVICE_PREP': True,     # loader + inference: float conversion / mean / flip / crop / resize / CHW padding
                                     # of the images on the GPU (naws_prep_image_fwd); threads only decode
            'HOST_NMS': False,       # True: per-class NMS with the numpy loop instead of the HIP kernel
            'TTA_PAIR_FLIPS': True,  # inference TTA: a scale's plain + mirrored pass as one batch of 2
            'LAGGED_STATS': False,   # training loop: read iteration i's scalars while i+1 runs (no per-iteration host sync;
                                     # the NaN / failed-loader stop then comes one iteration late).  Measured on the
                                     # synthetic roidb at 2 images per GPU: 19.3-19.8 ms per iteration either way
            'DEVICE_POST': True,     # inference: roi projection / dedup hash / scatter-back / TTA mean /
                                     # DETECTIONS_PER_IM cut on the GPU (csrc/infer_ops.hip): one result
                                     # download per image; False = the numpy path of the reference
            'PIPELINE_UPDATE': True,  # NUM_GPUS > 1, fp16x2 plan: the deferred update runs piece by piece as the
                                     # gradient messages arrive (fc6 biases, fc6_w in two row pieces, the rest) and
                                     # the next iteration's fc6 forward starts each piece behind ITS update, so the
                                     # tail of the exchange hides under fc6 forward as well (2 ranks: projected
                                     # 19.2 -> see DESIGN 5); bit-identical parameters (tests/test_gpu_two_ranks.py)
            'SYNTHETIC_TEST_IMAGES': 4,   # test engine, datasets that are not on disk: this many seeded synthetic
                                     # images stand in (tools/test_net_wsl.py --num-images)
            'SHARDED_UPDATE': False,  # NUM_GPUS > 1, fp16x2 plan: fc6_w's gradient rows are reduced to one owner
                                     # rank each, the owner updates its 8192 / N rows (fp32 master rows and momentum
                                     # live there only) and the updated rows + scale words return by all-gather:
                                     # same bytes on the links as the all-reduce; update traffic beside the next
                                     # conv body 4.9 / N + 1.6 (N-1)/N GB instead of 4.9 GB (a rank splits only the
                                     # rows it does not own after the gather); parameters bit-identical (tests).  A checkpoint
                                     # then needs engine.gather_sharded_state() on every rank (the training loop
                                     # calls it).  Unmeasured on hardware: no multi-GPU node was available
            'MFMA_DTYPE': 'fp16x2',  # 'fp32': fp32 MFMA everywhere; 'fp32x3': fc6/fc7 GEMMs as exact
                                     # 3-way bf16 splits on the bf16 MFMA (fp32-accurate, faster);
                                     # 'fp16x2': the same GEMMs as row-scaled 2-way f16 splits on
                                     # the f16 MFMA (same measured accuracy, faster still);
                                     # 'bf16': conv2..conv5 + fc6/fc7 operands rounded to bf16,
                                     # fp32 accumulation; storage, fc8, loss and SGD stay fp32
        },
    }


# Switches that select code outside the hot path: accepted only at their default.
_OFF_PATH_SWITCHES = (
    'MODEL.MASK_ON', 'MODEL.KEYPOINTS_ON', 'MODEL.RPN_ONLY', 'MODEL.FASTER_RCNN', 'RPN.RPN_ON',
    'FPN.FPN_ON', 'RETINANET.RETINANET_ON', 'WSL.CPG', 'WSL.CSC', 'WSL.CENTER_LOSS',
    'WSL.CONTEXT', 'WSL.PCL', 'WSL.CMIL',
)

cfg = CfgNode(_defaults())
__C = cfg


def reset_cfg():
    """Restore the defaults (tests)."""
    cfg.immutable(False)
    fresh = CfgNode(_defaults())
    for k in list(cfg.keys()):
        del cfg[k]
    for k, v in fresh.items():
        cfg[k] = v


def assert_and_infer_cfg(cache_urls=True, make_immutable=True):
    """ref: detectron/core/config.py:1178-1194.  URL caching has no meaning offline."""
    for key in _OFF_PATH_SWITCHES:
        node = cfg
        parts = key.split('.')
        for p in parts[:-1]:
            node = node[p]
        if node[parts[-1]]:
            raise NotImplementedError(
                '{} selects a model family outside the MI355X hot path (SURVEY.md §8)'.format(key))
    if make_immutable:
        cfg.immutable(True)


def get_output_dir(datasets, training=True):
    """<OUTPUT_DIR>/<train|test>/<dataset>/<MODEL.TYPE>  (ref: config.py:1210-1221)."""
    name = datasets if isinstance(datasets, str) else ':'.join(datasets)
    outdir = os.path.join(cfg.OUTPUT_DIR, 'train' if training else 'test', name, cfg.MODEL.TYPE)
    os.makedirs(outdir, exist_ok=True)
    return outdir


class _CfgLoader(yaml.SafeLoader):
    """safe_load + the ONE python tag a reference-written cfg string carries on its mappings
    (`!!python/object/new:detectron.utils.collections.AttrDict {dictitems, state}`, the yaml.dump
    of the reference's AttrDict: env.py:91) and `!!python/tuple`; anything else stays refused."""


def _construct_attrdict(loader, node):
    m = loader.construct_mapping(node, deep=True)
    return m.get('dictitems', {})


_CfgLoader.add_constructor('tag:yaml.org,2002:python/object/new:detectron.utils.collections.AttrDict',
                           _construct_attrdict)
_CfgLoader.add_constructor('tag:yaml.org,2002:python/object/new:utils.collections.AttrDict',
                           _construct_attrdict)
_CfgLoader.add_constructor('tag:yaml.org,2002:python/tuple',
                           lambda loader, node: tuple(loader.construct_sequence(node, deep=True)))


def load_cfg(cfg_to_load):
    if hasattr(cfg_to_load, 'read'):
        cfg_to_load = cfg_to_load.read()
    return yaml.load(cfg_to_load, Loader=_CfgLoader)


def merge_cfg_from_file(cfg_filename):
    with open(cfg_filename, 'r') as f:
        tree = load_cfg(f)
    _merge(tree or {}, cfg, [])


def merge_cfg_from_cfg(cfg_other):
    _merge(cfg_other, cfg, [])


# Options the reference still accepts and ignores / refuses with a pointer to the new name
# (detectron/core/config.py:1109-1164): same keys, same messages - a yaml or command line written
# for the reference behaves identically here.
_DEPRECATED_KEYS = {
    'FINAL_MSG', 'MODEL.DILATION', 'ROOT_GPU_ID', 'RPN.ON', 'TRAIN.BBOX_NORMALIZE_TARGETS_PRECOMPUTED',
    'TRAIN.DROPOUT', 'USE_GPU_NMS', 'TEST.NUM_TEST_IMAGES',
}
_RENAMED_KEYS = {
    'EXAMPLE.RENAMED.KEY': 'EXAMPLE.KEY',
    'MODEL.PS_GRID_SIZE': 'RFCN.PS_GRID_SIZE',
    'MODEL.ROI_HEAD': 'FAST_RCNN.ROI_BOX_HEAD',
    'MRCNN.MASK_HEAD_NAME': 'MRCNN.ROI_MASK_HEAD',
    'TRAIN.DATASET': ('TRAIN.DATASETS',
                      "Also convert to a tuple, e.g., 'coco_2014_train' -> ('coco_2014_train',) or "
                      "'coco_2014_train:coco_2014_valminusminival' -> "
                      "('coco_2014_train', 'coco_2014_valminusminival')"),
    'TRAIN.PROPOSAL_FILE': ('TRAIN.PROPOSAL_FILES',
                            "Also convert to a tuple, e.g., 'path/to/file' -> ('path/to/file',) or "
                            "'path/to/file1:path/to/file2' -> ('path/to/file1', 'path/to/file2')"),
    'TEST.SCALES': ('TEST.SCALE',
                    "Also convert from a tuple, e.g. (600, ), to a integer, e.g. 600."),
    'TEST.DATASET': ('TEST.DATASETS',
                     "Also convert from a string, e.g 'coco_2014_minival', to a tuple, e.g. "
                     "('coco_2014_minival', )."),
    'TEST.PROPOSAL_FILE': ('TEST.PROPOSAL_FILES',
                           "Also convert from a string, e.g. '/path/to/props.pkl', to a tuple, e.g. "
                           "('/path/to/props.pkl', )."),
}


def _key_is_deprecated(full_key):
    if full_key in _DEPRECATED_KEYS:
        import logging
        logging.getLogger(__name__).warning('Deprecated config key (ignoring): {}'.format(full_key))
        return True
    return False


def _raise_key_rename_error(full_key):
    new_key, msg = _RENAMED_KEYS[full_key], ''
    if isinstance(new_key, tuple):
        new_key, msg = new_key[0], ' Note: ' + new_key[1]
    raise KeyError('Key {} was renamed to {}; please update your config.{}'.format(
        full_key, new_key, msg))


def merge_cfg_from_list(cfg_list):
    """['TEST.NMS', 0.5, ...] — values are python literals or plain strings
    (ref: config.py:1252-1273, behaviour pinned by tests/golden/reference_cfg_behaviour.json)."""
    assert len(cfg_list) % 2 == 0
    for full_key, v in zip(cfg_list[0::2], cfg_list[1::2]):
        if _key_is_deprecated(full_key):
            continue
        if full_key in _RENAMED_KEYS:
            _raise_key_rename_error(full_key)
        node = cfg
        parts = full_key.split('.')
        for sub in parts[:-1]:
            assert sub in node, 'Non-existent key: {}'.format(full_key)
            node = node[sub]
        assert parts[-1] in node, 'Non-existent key: {}'.format(full_key)
        node[parts[-1]] = _coerce(_decode(v), node[parts[-1]], full_key)


def _decode(v):
    if isinstance(v, dict):
        return CfgNode(v)
    if not isinstance(v, str):
        return v
    try:
        return ast.literal_eval(v)
    except (ValueError, SyntaxError):
        return v


def _coerce(new, old, full_key):
    """Accept `new` if it has the default's type, or one of the reference's conversions
    (config.py:1393-1420): an ndarray default takes anything np.array accepts, a string default
    takes str(new) of ANYTHING, tuple <-> list.  Nothing else - in particular an int for a float
    default is a type mismatch there ('SOLVER.BASE_LR 1' fails) and therefore here."""
    t_old, t_new = type(old), type(new)
    if t_old is t_new:
        return new
    if isinstance(old, np.ndarray):
        return np.array(new, dtype=old.dtype)
    if isinstance(old, str):
        return str(new)
    if isinstance(new, tuple) and isinstance(old, list):
        return list(new)
    if isinstance(new, list) and isinstance(old, tuple):
        return tuple(new)
    raise ValueError('Type mismatch ({} vs. {}) with values ({} vs. {}) for config key: {}'.format(
        t_old, t_new, old, new, full_key))


def _merge(a, b, stack):
    for k, v_ in a.items():
        full_key = '.'.join(stack + [k])
        if k not in b:
            if _key_is_deprecated(full_key):
                continue
            if full_key in _RENAMED_KEYS:
                _raise_key_rename_error(full_key)
            raise KeyError('Non-existent config key: {}'.format(full_key))
        v = _decode(copy.deepcopy(v_))
        if isinstance(b[k], CfgNode):
            if not isinstance(v, dict):
                raise ValueError('Type mismatch ({} vs. {}) with values ({} vs. {}) for config '
                                 'key: {}'.format(type(b[k]), type(v), b[k], v, full_key))
            _merge(v, b[k], stack + [k])
        else:
            b[k] = _coerce(v, b[k], full_key)
